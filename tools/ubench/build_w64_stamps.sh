#!/bin/bash
# Diagnostic build of the library with shader-clock stamps in the W64 attention loop (-DW64_STAMPS): tools/ubench/_build/libgritlm_hip_w64stamps.so,
# used through GRIT_HIP_LIB by tools/attn_w64_stamps.py.  Never shipped.
set -e
cd "$(dirname "$0")/../../gritlm_amd/csrc"
mkdir -p ../../tools/ubench/_build/stamps_obj
for f in *.hip; do
  extra=""; [ "$f" = attention.hip ] && extra="${W64_EXTRA:--DW64_STAMPS}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$f" -o ../../tools/ubench/_build/stamps_obj/"${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ubench/_build/libgritlm_hip_w64stamps.so ../../tools/ubench/_build/stamps_obj/*.o
echo built tools/ubench/_build/libgritlm_hip_w64stamps.so
