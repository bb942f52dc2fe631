#!/bin/bash
# PROTOTYPE / ablation attention forward kernels (tools/ubench/attn_fwd_<src>.hip: copies of csrc/attention.hip with one structural change
# each) as private libraries for tools/ubench/attn_ab.bin:
#   build_attn_proto.sh <src> [name "flags"] ...   ->  tools/ubench/_var/libattn_proto_<name>.so   (no pairs: name = src, no flags)
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
SRC=$1; shift
[ $# -eq 0 ] && set -- "$SRC" ""
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable $2 -Igritlm_amd/csrc \
      -o tools/ubench/_var/libattn_proto_$1.so tools/ubench/attn_fwd_$SRC.hip tools/ubench/err_stub.hip &
  shift 2
done
wait
echo built
