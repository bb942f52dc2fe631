#!/bin/bash
# effective clock + MFMA busy of the QKV-shape GEMM under the ablation knobs (each in its own rocprofv3 --pmc run)
export TMPDIR=/tmp
mkdir -p gpurun_out/clk
for a in 0 1 3; do
  GRIT_GEMM_ABLATE=$a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/clk/a$a -o g --output-format csv -- python tools/gemm_ablate.py > gpurun_out/clk/a$a.log 2>&1
done
python - <<'PY'
import csv, collections
for a in (0,1,3):
    d=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f"gpurun_out/clk/a{a}/g_counter_collection.csv")):
        if "gemm_bf16" not in r["Kernel_Name"]: continue
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9)
    g=sum(d["GRBM_GUI_ACTIVE"])/len(d["GRBM_GUI_ACTIVE"]); t=sum(dur)/len(dur); m=sum(d["SQ_VALU_MFMA_BUSY_CYCLES"])/max(len(d["SQ_VALU_MFMA_BUSY_CYCLES"]),1)
    print(f"ABLATE={a}: dur {t*1e3:.2f} ms  clock {g/8/t/1e9:.2f} GHz  mfma_busy {m/(g/8*1024):.3f}")
PY
