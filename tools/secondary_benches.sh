#!/bin/bash
# The stand-alone benches behind README's second table, one after the other on the GPU box: bash tools/secondary_benches.sh <tag>
# -> gpurun_out/<tag>_*.json|log  (copy what is to be judged into profiles/).
TAG=${1:-r05}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { name=$1; shift; timeout 300 "$@" > gpurun_out/${TAG}_$name 2> gpurun_out/${TAG}_$name.err; echo "$name rc=$? $(tail -c 300 gpurun_out/${TAG}_$name | tr '\n' ' ')"; }
run decode_bench.json python tools/decode_bench.py
run generative_bench.json python tools/generative_bench.py
run mixtral_bench.json python tools/mixtral_bench.py --steps 2
run infonce_bench.log python tools/infonce_bench.py
run knn_bench.log python tools/knn_bench.py
run encode_e2e.log python tools/encode_e2e.py
run microbench.log python tools/gpu_microbench.py
