#!/bin/bash
# Same-box A/B of GEMM barrier variants (tools/ubench/_var/libgemm_<name>.so from build_gemm_flags.sh) against the base build:
# a tiny case first under a short timeout (a barrier-count mistake would hang the kernel), then the bit-equality cases, then the
# four 7B shapes timed interleaved.     usage: run_gemm_bar_ab.sh name [name ...]   (log: gpurun_out/gemm_bar_ab_<name>.log)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in "$@"; do
  log=gpurun_out/gemm_bar_ab_$v.log
  export GEMM_OLD=tools/ubench/_var/libgemm_base.so GEMM_NEW=tools/ubench/_var/libgemm_$v.so
  echo "== $v" > $log
  timeout 60 tools/ubench/gemm_ab.bin case 512 512 512 0 2 >> $log 2>&1 || { echo "tiny case failed/hung rc=$?" >> $log; continue; }
  timeout 60 tools/ubench/gemm_ab.bin case 9000 4352 512 1 2 >> $log 2>&1 || { echo "persistent case failed/hung rc=$?" >> $log; continue; }
  timeout 300 tools/ubench/gemm_ab.bin check 2 >> $log 2>&1
  timeout 300 tools/ubench/gemm_ab.bin time 6 >> $log 2>&1
  tail -8 $log
done
