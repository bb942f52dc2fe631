#!/bin/bash
# Debug build of the GEMM with shader-clock stamps at the seams of the persistent tile loop + its harness (tools/ubench/gemm_stamp.cpp).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGRIT_GEMM_STAMP -Wno-unused-variable -Igritlm_amd/csrc -o tools/ubench/_var/libgemm_stamp.so \
    gritlm_amd/csrc/gemm_bf16.hip tools/ubench/err_stub.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/gemm_stamp.bin tools/ubench/gemm_stamp.cpp -ldl
echo built
