#!/bin/bash
# Builds the attention-backward A/B harness: attention_bwd.hip of commit $1 (default: HEAD) as a private library + attn_bwd_ab.bin.
set -e
cd "$(dirname "$0")/../.."
REF=${1:-HEAD}
mkdir -p tools/ubench/_r01/csrc_attn_bwd tools/ubench/_r01/include
git show $REF:gritlm_amd/csrc/attention_bwd.hip > tools/ubench/_r01/csrc_attn_bwd/attention_bwd.hip
git show $REF:gritlm_amd/csrc/common.h | sed 's#../../include/gritlm_hip.h#../include/gritlm_hip.h#' > tools/ubench/_r01/csrc_attn_bwd/common.h
git show $REF:include/gritlm_hip.h > tools/ubench/_r01/include/gritlm_hip.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable -Itools/ubench/_r01/csrc_attn_bwd \
    -o tools/ubench/_r01/libattn_bwd_old.so tools/ubench/_r01/csrc_attn_bwd/attention_bwd.hip tools/ubench/err_stub.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/attn_bwd_ab.bin tools/ubench/attn_bwd_ab.cpp -ldl
echo built
