#!/bin/bash
# Builds the attention A/B harness: the kernel of commit $1 (default: HEAD) as a private library + tools/ubench/attn_ab.bin.
set -e
cd "$(dirname "$0")/../.."
REF=${1:-HEAD}
mkdir -p tools/ubench/_r01/csrc_attn tools/ubench/_r01/include
git show $REF:gritlm_amd/csrc/attention.hip > tools/ubench/_r01/csrc_attn/attention.hip
git show $REF:gritlm_amd/csrc/common.h | sed 's#../../include/gritlm_hip.h#../include/gritlm_hip.h#' > tools/ubench/_r01/csrc_attn/common.h
git show $REF:include/gritlm_hip.h > tools/ubench/_r01/include/gritlm_hip.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable -Itools/ubench/_r01/csrc_attn \
    -o tools/ubench/_r01/libattn_old.so tools/ubench/_r01/csrc_attn/attention.hip tools/ubench/err_stub.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/attn_ab.bin tools/ubench/attn_ab.cpp -ldl
echo built
