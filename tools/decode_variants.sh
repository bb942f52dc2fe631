#!/bin/bash
# A/B of the decode step's GEMV launch forms on one box (round 5): RMSNorm fused into the following GEMV or launched on its own, and the
# number of weight rows per GEMV workgroup (GRIT_GV_ROWS builds of csrc/decode.hip, loaded through GRIT_HIP_LIB).
#   build (CPU, here):  bash tools/decode_variants.sh build        run (GPU box):  bash tools/decode_variants.sh run > gpurun_out/decode_variants.log
cd "$(dirname "$0")/.."
B=tools/ubench/_build/decode_variants
if [ "$1" = build ]; then
  mkdir -p $B
  for r in 2 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGRIT_GV_ROWS=$r -c gritlm_amd/csrc/decode.hip -o $B/decode_gv$r.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgritlm_hip_gv$r.so $B/decode_gv$r.o $(ls gritlm_amd/csrc/*.o | grep -v '/decode.o') || exit 1
  done
  ls -la $B; exit 0
fi
run() { echo "== $1"; shift; env "$@" python tools/decode_bench.py --new 64 2>&1 | tail -1; }
run "default (norms fused, 4 rows per workgroup)" A=1
run "MLP norm as its own launch" GRIT_DECODE_FUSE_NORM=qkv
run "no norm fused" GRIT_DECODE_FUSE_NORM=none
run "8 rows per workgroup" GRIT_HIP_LIB=$PWD/$B/libgritlm_hip_gv8.so
run "2 rows per workgroup" GRIT_HIP_LIB=$PWD/$B/libgritlm_hip_gv2.so
run "8 rows per workgroup, MLP norm as its own launch" GRIT_HIP_LIB=$PWD/$B/libgritlm_hip_gv8.so GRIT_DECODE_FUSE_NORM=qkv
run "default again" A=1
