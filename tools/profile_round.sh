#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box: bash tools/profile_round.sh <tag>
# (kernel-trace stats of bench.py; separate --pmc passes on the GEMM shapes, never combined with other traces).
# Only the small summaries stay under gpurun_out/ (the raw kernel traces of a training step are tens of MB; gpurun merges <= 64 MiB back).
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
RAW=/tmp/prof_raw_$TAG; rm -rf $RAW; mkdir -p $RAW
if [ -n "$PMC_ONLY" ]; then     # PMC_ONLY=1 [PROBE_F16=1]: only the four counter passes on tools/gemm_probe.py (e.g. the fp16-operand instantiations)
for pass in "pmc_sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "pmc_fetch:FETCH_SIZE" "pmc_write:WRITE_SIZE" "pmc_mfma:GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs -d $RAW/$name -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/$name.log 2>&1
  mkdir -p $OUT/$name; cp $(find $RAW/$name -name "*counter_collection.csv" | head -1) $OUT/$name/gemm_counter_collection.csv
done
du -sh $OUT; exit 0
fi
rocprofv3 --kernel-trace --stats -d $RAW/bench_stats -o bench --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-contrastive --no-ragged --no-torch-baseline --no-mixtral --no-rag > $OUT/bench_stats.json 2> $OUT/bench_stats.err
mkdir -p $OUT/bench_stats; cp $(find $RAW/bench_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_stats/bench_kernel_stats.csv
for pass in "pmc_sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "pmc_fetch:FETCH_SIZE" "pmc_write:WRITE_SIZE" "pmc_mfma:GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs -d $RAW/$name -o gemm --output-format csv -- python tools/gemm_probe.py > $OUT/$name.log 2>&1
  mkdir -p $OUT/$name; cp $(find $RAW/$name -name "*counter_collection.csv" | head -1) $OUT/$name/gemm_counter_collection.csv
done
[ -n "$SKIP_CONTRASTIVE" ] && { du -sh $OUT; exit 0; }     # SKIP_CONTRASTIVE=1: encode stats + GEMM counters only (~3 min instead of ~8)
rocprofv3 --kernel-trace --stats -d $RAW/bench_contrastive -o bench --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --no-mixtral --no-rag --contrastive-parity-pairs 0 --contrastive-steps 1 --pairs ${PAIRS:-256} --chunk 32 > $OUT/bench_contrastive.json 2> $OUT/bench_contrastive.err
mkdir -p $OUT/bench_contrastive; cp $(find $RAW/bench_contrastive -name "*kernel_stats.csv" | head -1) $OUT/bench_contrastive/bench_kernel_stats.csv
du -sh $OUT; find $OUT -type f | xargs ls -la
