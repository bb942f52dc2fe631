#!/bin/bash
# A/B of the weight rows per workgroup of the RESIDUAL GEMVs (o_proj, down) in the decode step, one box (round 5).
cd "$(dirname "$0")/.."
run() { echo "== $1"; shift; env "$@" python tools/decode_bench.py --new 64 2>&1 | tail -1; }
run "4 rows per workgroup (1024 workgroups for N = 4096)" GRIT_GV_ROWS_RES=4
run "2 rows" GRIT_GV_ROWS_RES=2
run "1 row" GRIT_GV_ROWS_RES=1
run "4 rows again" GRIT_GV_ROWS_RES=4
