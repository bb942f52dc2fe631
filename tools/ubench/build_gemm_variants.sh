#!/bin/bash
# Builds one private library per GRIT_GEMM_VAR value (gemm_bf16.hip + error stub) for tools/ubench/gemm_ab.bin (GEMM_NEW=...).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable -DGRIT_GEMM_VAR=$v -Igritlm_amd/csrc \
      -o tools/ubench/_var/libgemm_v$v.so gritlm_amd/csrc/gemm_bf16.hip tools/ubench/err_stub.hip &
done
wait
echo built "$@"
