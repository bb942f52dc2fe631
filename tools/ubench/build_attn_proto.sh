#!/bin/bash
# PROTOTYPE attention forward kernels (tools/ubench/attn_fwd_*.hip: copies of csrc/attention.hip with one structural change each) as
# private libraries for tools/ubench/attn_ab.bin:   build_attn_proto.sh pipe [extra flags]  ->  tools/ubench/_var/libattn_proto_pipe.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable $2 -Igritlm_amd/csrc \
    -o tools/ubench/_var/libattn_proto_$1.so tools/ubench/attn_fwd_$1.hip tools/ubench/err_stub.hip
echo built tools/ubench/_var/libattn_proto_$1.so
