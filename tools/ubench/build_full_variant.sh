#!/bin/bash
# A complete libgritlm_hip.so built from the CURRENT sources with extra compiler flags, for model-level A/B runs through the
# GRIT_HIP_LIB hook of gritlm_amd/_lib.py:   build_full_variant.sh <name> "<flags>"  ->  tools/ubench/_var/full_<name>/libgritlm_hip.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; FLAGS=$2
OUT=tools/ubench/_var/full_$NAME
mkdir -p $OUT
for f in gritlm_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $FLAGS -Igritlm_amd/csrc -c $f -o $OUT/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libgritlm_hip.so $OUT/*.o
rm -f $OUT/*.o
echo built $OUT/libgritlm_hip.so
