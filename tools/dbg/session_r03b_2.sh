#!/bin/bash
# GPU session 2: full GPU checks (new RMSNorm backward, shared silu / swiglu-bwd arithmetic, opaque-zero persistent GEMM), GEMM A/B
# (folded vs opaque accumulator zeros; silu division vs v_rcp), attention forward variants (asm transposing reads / asm K reads).
cd "$(dirname "$0")/../.."
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
( timeout 200 python tools/dbg/run_checks.py rmsnorm_bwd rmsnorm_bwd_4096_res swiglu_train_epilogues swiglu_train_epilogues_big ) > $O/checks.log 2>&1
V=tools/ubench/_var
( GEMM_OLD=$V/libgemm_folded.so GEMM_NEW=$V/libgemm_opaque.so timeout 300 tools/ubench/gemm_ab.bin check 2 ) > $O/gemm_ab_check.log 2>&1
( GEMM_OLD=$V/libgemm_folded.so GEMM_NEW=$V/libgemm_opaque.so timeout 300 tools/ubench/gemm_ab.bin time 6 ) > $O/gemm_ab_time.log 2>&1
( GEMM_OLD=$V/libgemm_folded.so GEMM_NEW=$V/libgemm_opaque.so AB_M=16384 timeout 300 tools/ubench/gemm_ab.bin time 6 ) >> $O/gemm_ab_time.log 2>&1
( GEMM_OLD=$V/libgemm_silu_div.so GEMM_NEW=$V/libgemm_opaque.so timeout 200 tools/ubench/gemm_ab.bin case 131072 28672 4096 2 6 ) > $O/gemm_ab_silu.log 2>&1
for v in asmtr asmk2 asmk3 asmk2tr asmk3tr; do
  ( echo "== $v"; ATTN_OLD=$V/libattn_base.so ATTN_NEW=$V/libattn_$v.so timeout 200 tools/ubench/attn_ab.bin all ) >> $O/attn_ab.log 2>&1
done
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-contrastive --no-ragged --no-torch-baseline ) > $O/bench_encode.json 2> $O/bench_encode.err
tail -4 $O/pytest.log; cat $O/checks.log | cut -c1-300; tail -2 $O/gemm_ab_check.log; cat $O/gemm_ab_time.log $O/gemm_ab_silu.log; grep -E "==|B=256|B=64 S=2048|RESULT|B=8 S=512" $O/attn_ab.log; python -c "
import json;d=json.load(open('$O/bench_encode.json'));print(d['value'],d['roofline']['frac'],{k:round(v['tflops']) for k,v in d['roofline']['by_shape'].items()})"
