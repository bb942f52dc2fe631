#!/bin/bash
# quick attention A/B session: bash tools/dbg/session_attn.sh <out dir> <old lib name> <variant> [variant ...]   (libraries in tools/ubench/_var/libattn_<name>.so)
cd "$(dirname "$0")/../.."
[ -n "$ATTN_TOL" ] && export ATTN_TOL
O=gpurun_out/$1; OLD=$2; shift 2; mkdir -p $O
V=tools/ubench/_var
for v in "$@"; do
  ( echo "== $v vs $OLD"; ATTN_OLD=$V/libattn_$OLD.so ATTN_NEW=$V/libattn_$v.so timeout 200 tools/ubench/attn_ab.bin ${MODE:-all} ) >> $O/attn_ab.log 2>&1
done
grep -E "==|old .* new|RESULT" $O/attn_ab.log | cut -c1-175
