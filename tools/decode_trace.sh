#!/bin/bash
# Per-kernel durations of the native decode step (rocprofv3 --kernel-trace --stats): bash tools/decode_trace.sh <tag> [decode_bench.py flags, e.g. --precision f16_stream]  ->  gpurun_out/<tag>_decode_kernel_stats.csv
cd "$(dirname "$0")/.."
TAG=${1:-r05}
shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/dprof
rocprofv3 --kernel-trace --stats -d gpurun_out/dprof -o dec --output-format csv -- python tools/decode_bench.py --new 64 "$@" > gpurun_out/${TAG}_decode_bench_under_rocprof.json 2> gpurun_out/dprof.err
f=$(find gpurun_out/dprof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_decode_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:110]:110s} {r["Calls"]:>7s} {float(r["AverageNs"]) / 1e3:8.2f} us {r["Percentage"]:>6s} %')
PY
tail -1 gpurun_out/${TAG}_decode_bench_under_rocprof.json
