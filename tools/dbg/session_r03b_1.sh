#!/bin/bash
# GPU session 1 of the second half of round 3: full GPU checks, SWIGLU_BWD epilogue A/B (kernel level), contrastive A/B (model level).
cd "$(dirname "$0")/../.."
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log 2>&1
V=tools/ubench/_var
( GEMM_OLD=$V/libgemm_swb_direct.so GEMM_NEW=$V/libgemm_swb_lds.so timeout 300 tools/ubench/gemm_ab.bin check 2 ) > $O/ab_check.log 2>&1
for shape in "16384 14336 4096" "65536 14336 4096"; do
  ( GEMM_OLD=$V/libgemm_swb_direct.so GEMM_NEW=$V/libgemm_swb_lds.so timeout 120 tools/ubench/gemm_ab.bin case $shape 6 6 ) >> $O/ab_time.log 2>&1
  ( GEMM_OLD=$V/libgemm_swb_lds.so GEMM_NEW=$V/libgemm_swb_lds_rcp.so timeout 120 tools/ubench/gemm_ab.bin case $shape 6 6 ) >> $O/ab_time.log 2>&1
  ( GEMM_OLD=$V/libgemm_swb_direct.so GEMM_NEW=$V/libgemm_swb_direct_rcp.so timeout 120 tools/ubench/gemm_ab.bin case $shape 6 6 ) >> $O/ab_time.log 2>&1
done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --pairs 128"
( GRIT_HIP_LIB=$PWD/$V/full_swb_direct/libgritlm_hip.so GRIT_GRADCACHE_PASS1_MULT=1 timeout 400 $B ) > $O/bench_direct_mult1.json 2> $O/bench_direct_mult1.err
( GRIT_GRADCACHE_PASS1_MULT=4 timeout 400 $B ) > $O/bench_lds_mult4.json 2> $O/bench_lds_mult4.err
( GRIT_HIP_LIB=$PWD/$V/full_swb_lds_rcp/libgritlm_hip.so GRIT_GRADCACHE_PASS1_MULT=4 timeout 400 $B ) > $O/bench_ldsrcp_mult4.json 2> $O/bench_ldsrcp_mult4.err
( GRIT_GRADCACHE_PASS1_MULT=1 timeout 400 $B ) > $O/bench_lds_mult1.json 2> $O/bench_lds_mult1.err
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["contrastive"]
    print(sys.argv[1].split("/")[-1], "docs/s %.1f frac %.4f | pairs/s %.4f frac %.4f ms %.0f pass1rows %s" % (d["value"], d["roofline"]["frac"], c["value"], c["mfma_roofline_frac"], c["ms_per_step"], c.get("gradcache_pass1_rows_per_call")))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt 2>&1
cat $O/pytest.log | tail -5; tail -3 $O/ab_check.log; cat $O/ab_time.log; cat $O/summary.txt
