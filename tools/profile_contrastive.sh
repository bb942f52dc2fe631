#!/bin/bash
# rocprofv3 kernel statistics of bench.py with ONE warm-up + ONE timed contrastive (GradCache) step next to a short encode leg:
#   bash tools/profile_contrastive.sh <tag>   ->  gpurun_out/prof_<tag>/bench_contrastive/bench_kernel_stats.csv (+ the bench line)
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/bench_contrastive
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
RAW=/tmp/prof_raw_${TAG}_c; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats -d $RAW -o bench --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --no-mixtral --no-rag --contrastive-parity-pairs 0 --contrastive-steps 1 --pairs ${PAIRS:-256} --chunk 32 > $OUT/bench_contrastive.json 2> $OUT/bench_contrastive.err
cp $(find $RAW -name "*kernel_stats.csv" | head -1) $OUT/bench_contrastive/bench_kernel_stats.csv
ls -la $OUT/bench_contrastive
