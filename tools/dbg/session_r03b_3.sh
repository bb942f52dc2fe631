#!/bin/bash
# GPU session 3: attention forward variants on top of the asm reads (LDS-DMA pieces spread between the QK products), new dw reduce
cd "$(dirname "$0")/../.."
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
V=tools/ubench/_var
for v in k2tr k2tr_spread k3tr_spread k3tr; do
  ( echo "== $v"; ATTN_OLD=$V/libattn_base.so ATTN_NEW=$V/libattn_$v.so timeout 200 tools/ubench/attn_ab.bin all ) >> $O/attn_ab.log 2>&1
done
( echo "== k2tr_spread vs k2tr"; ATTN_OLD=$V/libattn_k2tr.so ATTN_NEW=$V/libattn_k2tr_spread.so timeout 200 tools/ubench/attn_ab.bin check ) >> $O/attn_ab.log 2>&1
( timeout 300 python tools/dbg/run_checks.py rmsnorm_bwd rmsnorm_bwd_4096_res rmsnorm_bwd_2048_res rmsnorm_bwd_4096 attn_production_shape attn_seam_qpw2 attn_seam_qpw4 attn_ragged encoder_7b_layer train_7b_layer swiglu_train_epilogues gemm_swiglu ) > $O/checks.log 2>&1
grep -E "==|old .* new|RESULT" $O/attn_ab.log; cut -c1-250 $O/checks.log
