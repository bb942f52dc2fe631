#!/bin/bash
# Builds the GEMM A/B harness: the previous round's kernel (from git history, commit $1, default a64d1f4) as a private shared
# library + tools/ubench/gemm_ab.bin.  Needs the .git directory (run in the build container; the outputs travel to the GPU box).
set -e
cd "$(dirname "$0")/../.."
REF=${1:-a64d1f4}
mkdir -p tools/ubench/_r01/csrc tools/ubench/_r01/include
git show $REF:gritlm_amd/csrc/gemm_bf16.hip > tools/ubench/_r01/csrc/gemm_bf16.hip
git show $REF:gritlm_amd/csrc/common.h | sed 's#../../include/gritlm_hip.h#../include/gritlm_hip.h#' > tools/ubench/_r01/csrc/common.h
git show $REF:include/gritlm_hip.h > tools/ubench/_r01/include/gritlm_hip.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable -o tools/ubench/_r01/libgemm_r01.so \
    -Itools/ubench/_r01/csrc tools/ubench/_r01/csrc/gemm_bf16.hip tools/ubench/err_stub.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/gemm_ab.bin tools/ubench/gemm_ab.cpp -ldl
echo built
