#!/bin/bash
# rocprofv3 --pmc passes (own runs, --kernel-trace only) on tools/attn_pmc_probe.py: the bf16 and the fp16 instantiation of the attention
# forward at B 256 x S 512 and B 32 x S 2048 -> gpurun_out/attn_pmc_<tag>/summary.json (copy to profiles/<tag>_attn_fwd_pmc.json)
TAG=${1:-r05}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/attn_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for pass in "a:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "b:GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  RAW=/tmp/attn_pmc_raw_$TAG/$name; rm -rf $RAW; mkdir -p $RAW
  timeout 280 rocprofv3 --kernel-trace --pmc $ctrs -d $RAW -o a --output-format csv -- python tools/attn_pmc_probe.py > $OUT/$name.log 2>&1
  cp $(find $RAW -name "*counter_collection.csv" | head -1) $OUT/$name.csv 2>/dev/null
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/?.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn_bidir_fwd_k<false, false, true>" in n: k = "fp16 operands (attn_bidir_fwd_k<false,false,true>)"
        elif "attn_bidir_fwd_k<false, false, false>" in n: k = "bf16 (attn_bidir_fwd_k<false,false,false>)"
        else: continue
        key = f"{k} {'B256xS512' if int(r['Grid_Size']) == 2097152 else 'B32xS2048'}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            agg[key]["_dur_ns_" + r["Counter_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, c in res.items():
    wc = c.get("SQ_WAVE_CYCLES") or 1.0
    mf = c.get("SQ_INSTS_MFMA") or 1.0
    c["derived"] = {
        "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
        "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        "mfma_pipe_busy_frac_of_simd_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024) if c.get("GRBM_GUI_ACTIVE") else None,
        "instructions_per_mfma": {"valu_incl_mfma": c.get("SQ_INSTS_VALU", 0) / mf, "salu": c.get("SQ_INSTS_SALU", 0) / mf, "lds": c.get("SQ_INSTS_LDS", 0) / mf,
                                  "smem": c.get("SQ_INSTS_SMEM", 0) / mf, "vmem_rd": c.get("SQ_INSTS_VMEM_RD", 0) / mf},
        "effective_clock_ghz": (c.get("GRBM_GUI_ACTIVE", 0) / 8) / c["_dur_ns_GRBM_GUI_ACTIVE"] if c.get("_dur_ns_GRBM_GUI_ACTIVE") else None,
        "avg_duration_us": c.get("_dur_ns_SQ_WAVE_CYCLES", 0) / 1e3}
    print(k, json.dumps(c["derived"]))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
