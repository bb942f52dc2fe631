#!/bin/bash
# rocprofv3 --pmc passes (own runs, --kernel-trace only) on tools/attn_bwd_pmc_probe.py: the attention backward kernels at the contrastive
# step's chunk shape (32 x 512, packed and padded) and at 8 x 2048 -> gpurun_out/attn_bwd_pmc_<tag>/summary.json
# (copy to profiles/<tag>_attn_bwd_pmc.json).   bash tools/attn_bwd_pmc.sh <tag>
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/attn_bwd_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "b:GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  RAW=/tmp/attn_bwd_pmc_raw_$TAG/$name; rm -rf $RAW; mkdir -p $RAW
  timeout 280 rocprofv3 --kernel-trace --pmc $ctrs -d $RAW -o a --output-format csv -- python tools/attn_bwd_pmc_probe.py > $OUT/$name.log 2>&1
  cp $(find $RAW -name "*counter_collection.csv" | head -1) $OUT/$name.csv 2>/dev/null
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/?.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_bwd" not in n and "attn_delta" not in n:
            continue
        key = n.split("(")[0].replace("void grit::", "") + f" grid {r['Grid_Size']}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            agg[key]["_dur_ns_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, c in res.items():
    wc = c.get("SQ_WAVE_CYCLES") or 1.0
    mf = c.get("SQ_INSTS_MFMA") or 1.0
    c["derived"] = {
        "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
        "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "lds_wait_frac": c.get("SQ_WAIT_INST_LDS", 0) / wc,
        "mfma_pipe_busy_frac_of_simd_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else None,
        "instructions_per_mfma": {"valu_incl_mfma": c.get("SQ_INSTS_VALU", 0) / mf, "salu": c.get("SQ_INSTS_SALU", 0) / mf, "lds": c.get("SQ_INSTS_LDS", 0) / mf,
                                  "smem": c.get("SQ_INSTS_SMEM", 0) / mf, "vmem_rd": c.get("SQ_INSTS_VMEM_RD", 0) / mf} if c.get("SQ_INSTS_MFMA") else None,
        "effective_clock_ghz": (c.get("GRBM_GUI_ACTIVE", 0) / 8) / c["_dur_ns_GRBM_GUI_ACTIVE"] if c.get("_dur_ns_GRBM_GUI_ACTIVE") else None,
        "avg_duration_us": c.get("_dur_ns_SQ_WAVE_CYCLES", 0) / 1e3}
    print(k, json.dumps(c["derived"]))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
