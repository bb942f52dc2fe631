#!/bin/bash
# Private attention-forward libraries for tools/ubench/attn_ab.bin from the CURRENT attention.hip with extra compiler flags (how the
# round-3 levers were A/B'd one by one; the knobs they used are resolved in the source now -- the reference kernel of a comparison comes
# from git history: build_attn_ab.sh <commit>):
#   build_attn_flags.sh name1 "-DFLAG_A" name2 "-DFLAG_B -DFLAG_C" ...   ->  tools/ubench/_var/libattn_<name>.so
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/_var
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable $2 -Igritlm_amd/csrc \
      -o tools/ubench/_var/libattn_$1.so gritlm_amd/csrc/attention.hip tools/ubench/err_stub.hip &
  shift 2
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/ubench/attn_ab.bin tools/ubench/attn_ab.cpp -ldl
echo built
