#!/bin/bash
# quick attention-backward A/B session: bash tools/dbg/session_attn_bwd.sh <out dir> <old lib name> <variant> [variant ...]
cd "$(dirname "$0")/../.."
O=gpurun_out/$1; OLD=$2; shift 2; mkdir -p $O
V=tools/ubench/_var
for v in "$@"; do
  ( echo "== $v vs $OLD"; ATTN_OLD=$V/libattn_bwd_$OLD.so ATTN_NEW=$V/libattn_bwd_$v.so timeout 300 tools/ubench/attn_bwd_ab.bin ${MODE:-all} ) >> $O/attn_bwd_ab.log 2>&1
done
grep -E "==|old .* new|RESULT|MISMATCH" $O/attn_bwd_ab.log | cut -c1-190
