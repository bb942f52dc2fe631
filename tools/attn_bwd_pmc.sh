#!/bin/bash
# rocprofv3 --pmc pass (own run, --kernel-trace only) on the attention BACKWARD harness: bash tools/attn_bwd_pmc.sh <tag> [lib path]
cd "$(dirname "$0")/.."
TAG=$1; L=${2:-$PWD/gritlm_amd/libgritlm_hip.so}
export TMPDIR=/tmp
OUT=gpurun_out/attn_bwd_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
RAW=/tmp/attn_bwd_pmc_raw; rm -rf $RAW; mkdir -p $RAW
ATTN_OLD=$L ATTN_NEW=$L rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $RAW -o a --output-format csv -- tools/ubench/attn_bwd_ab.bin time > $OUT/run.log 2>&1
cp $(find $RAW -name "*counter_collection.csv" | head -1) $OUT/a.csv
python3 - "$OUT" <<'PY'
import collections, csv, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{out}/a.csv")):
    n = r["Kernel_Name"]
    if "attn_bwd" not in n and "attn_delta" not in n:
        continue
    key = n.split("(")[0].replace("void grit::", "") + f" grid {r['Grid_Size']}"
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        agg[key]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
for k, c in res.items():
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(k, "dur_us %.0f parked %.3f issue-stall %.3f active %.3f wait_lds %.3f mfma/wavecyc %.3f" % (c["_dur_ns"] / 1e3, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)))
PY
grep -E "old .* new|RESULT" $OUT/run.log | cut -c1-200
