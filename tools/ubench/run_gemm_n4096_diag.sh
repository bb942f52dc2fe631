#!/bin/bash
# Is the o_proj / down gap to the vendor GEMM the RESIDUAL epilogue or the N = 4096 shape?  STORE vs RESIDUAL at N = 4096 (K = 4096 and
# 14336), and the tile-group height GRIT_GEMM_GM on those shapes (base build in both slots of the harness: the columns are two samples).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
log=gpurun_out/gemm_n4096_diag.log; : > $log
export GEMM_OLD=tools/ubench/_var/libgemm_base.so GEMM_NEW=tools/ubench/_var/libgemm_base.so
for gm in 4 2 8; do
  echo "== GRIT_GEMM_GM=$gm" >> $log
  for c in "131072 4096 4096 0" "131072 4096 4096 1" "131072 4096 14336 0" "131072 4096 14336 1" "131072 6144 4096 0"; do
    GRIT_GEMM_GM=$gm timeout 120 tools/ubench/gemm_ab.bin case $c 4 2>&1 | grep "^time" >> $log
  done
done
cat $log
