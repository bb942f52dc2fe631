#!/bin/bash
# GPU session: full checks on the final attention forward (asm reads, spread DMA, early V reads), persistent form at K = 14336 / 16384, encode bench
cd "$(dirname "$0")/../.."
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
V=tools/ubench/_var
for c in "131072 4096 14336 1" "16384 4096 14336 1" "28672 4096 16384 1" "14336 4096 16384 1" "131072 4096 14336 0"; do
  ( GEMM_SELF_AB=1 GRIT_GEMM_PERSIST_MAXKT=256 GEMM_OLD=$V/libgemm_cur_copy.so GEMM_NEW=$V/libgemm_cur.so timeout 200 tools/ubench/gemm_ab.bin case $c 6 ) >> $O/gemm_persist_k.log 2>&1
done
( ATTN_OLD=$V/libattn_base.so ATTN_NEW=gritlm_amd/libgritlm_hip.so timeout 200 tools/ubench/attn_ab.bin all ) > $O/attn_final.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-contrastive --no-ragged ) > $O/bench_encode.json 2> $O/bench_encode.err
tail -3 $O/pytest.log; grep -E "time|RESULT" $O/gemm_persist_k.log; grep -E "old .* new|RESULT" $O/attn_final.log | cut -c1-170; python -c "
import json;d=json.load(open('$O/bench_encode.json'));print(d['value'],d['roofline']['frac'],{k:round(v['tflops']) for k,v in d['roofline']['by_shape'].items()}, d['kernels']['attn_bidir_fwd'], d['roofline'].get('vendor_gemm_tflops_same_shapes_no_epilogue'), d.get('rocm_torch_baseline',{}).get('value'))"
