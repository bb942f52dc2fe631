#!/bin/bash
# A/B of the decode step's RMSNorm placement on one box (round 5): exact fused / own launch / deferred form (grit_rmsnorm_gemv_bf16_deferred).
#   bash tools/decode_norm_ab.sh > gpurun_out/decode_norm_ab.log
cd "$(dirname "$0")/.."
run() { echo "== $1"; shift; env "$@" python tools/decode_bench.py --new 64 2>&1 | tail -1; }
run "qkv (q|k|v norm fused exactly, MLP / final norm own launches: the round-5 default so far)" GRIT_DECODE_FUSE_NORM=qkv
run "deferred_mlp (q|k|v exact fused, MLP / final norm deferred)" GRIT_DECODE_FUSE_NORM=deferred_mlp
run "deferred (all three norms deferred)" GRIT_DECODE_FUSE_NORM=deferred
run "all (all three exact fused)" GRIT_DECODE_FUSE_NORM=all
run "qkv again" GRIT_DECODE_FUSE_NORM=qkv
run "deferred again" GRIT_DECODE_FUSE_NORM=deferred
