#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box: bash tools/profile_round.sh <tag>
# (kernel-trace stats of bench.py; separate --pmc passes on the GEMM shapes, never combined with other traces)
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o bench --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-contrastive --no-ragged --no-torch-baseline > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA -d $OUT/pmc_mfma -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/bench_contrastive -o bench --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --pairs ${PAIRS:-256} --chunk 32 > $OUT/bench_contrastive.json 2> $OUT/bench_contrastive.err
find $OUT -name "*.csv" | xargs ls -la
