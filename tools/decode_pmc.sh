#!/bin/bash
# HBM-side traffic of the decode step's kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, each in its own run with --kernel-trace only):
#   bash tools/decode_pmc.sh <tag> [decode_bench.py flags, e.g. --precision f16_stream]  ->  gpurun_out/<tag>_decode_pmc.json   (FETCH_SIZE x 2 on gfx950, KB units: the guide's HBM section)
cd "$(dirname "$0")/.."
TAG=${1:-r05}
shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/dpmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/dpmc_$c -o dec --output-format csv -- python tools/decode_bench.py --new 8 --no-graph "$@" > /tmp/dpmc_$c.log 2>&1
done
python - "$TAG" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/dpmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "grit::" in n and ("gemv" in n or "attn_decode" in n or "argmax" in n):
            agg[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, d in agg.items():
    fe, wr = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    # the two instantiation-sharing launches (o_proj / down: same template arguments) are told apart by their traffic: report the mean and both modes
    out[n] = {"launches": len(fe), "fetch_mb_mean": sum(fe) / len(fe) * 2 * 1024 / 1e6, "write_mb_mean": sum(wr) / max(len(wr), 1) * 1024 / 1e6,
              "fetch_mb_min": min(fe) * 2 * 1024 / 1e6, "fetch_mb_max": max(fe) * 2 * 1024 / 1e6}
res = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own runs) on tools/decode_bench.py --new 8 --no-graph: HBM-side MB per launch (FETCH_SIZE x 2 x 1 KB on gfx950)",
       "algorithmic_weight_mb": {"q|k|v 6144x4096": 50.3, "o_proj 4096x4096": 33.6, "gate|up 28672x4096": 234.9, "down 4096x14336": 117.4, "lm_head 32000x4096": 262.1},
       "kernels": out}
json.dump(res, open(f"gpurun_out/{tag}_decode_pmc.json", "w"), indent=1)
for n, v in out.items():
    print(f"{n[:60]:60s} n={v['launches']:6d} fetch {v['fetch_mb_mean']:8.1f} MB (min {v['fetch_mb_min']:.1f} max {v['fetch_mb_max']:.1f})  write {v['write_mb_mean']:6.2f} MB")
PY
