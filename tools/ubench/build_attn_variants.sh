#!/bin/bash
# Builds the attention A/B set: the kernel of commit $1 as _r01/libattn_old.so, the working-tree kernel with the deferred softmax rescale
# off (must be bit-identical to the old one) as _r01/libattn_nodefer.so and on (the shipped default) as _r01/libattn_new.so.
set -e
cd "$(dirname "$0")/../.."
bash tools/ubench/build_attn_ab.sh ${1:-HEAD}
for v in "nodefer:-DATT_DEFER_MAX=0" "new:-DATT_DEFER_MAX=1"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable ${v#*:} -Igritlm_amd/csrc \
      -o tools/ubench/_r01/libattn_${v%%:*}.so gritlm_amd/csrc/attention.hip tools/ubench/err_stub.hip
done
echo built variants
