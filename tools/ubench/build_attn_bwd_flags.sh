#!/bin/bash
# Private attention-backward libraries for tools/ubench/attn_bwd_ab.bin from the CURRENT attention_bwd.hip with extra compiler flags
# (the reference kernel of a comparison comes from git history: build_attn_bwd_ab.sh <commit>):
#   build_attn_bwd_flags.sh name1 "-DFLAG_A" name2 "-DFLAG_B" ...   ->  tools/ubench/_var/libattn_bwd_<name>.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable $2 -Igritlm_amd/csrc \
      -o tools/ubench/_var/libattn_bwd_$1.so gritlm_amd/csrc/attention_bwd.hip tools/ubench/err_stub.hip &
  shift 2
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/attn_bwd_ab.bin tools/ubench/attn_bwd_ab.cpp -ldl
echo built
