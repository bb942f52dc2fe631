#!/bin/bash
# Same-box A/B against an earlier revision of ONE kernel file: builds tools/ubench/_build/libgritlm_hip_prev.so from the current tree with
# <file> taken from git revision <rev> (default HEAD).   usage: build_prev_lib.sh attention.hip [rev]
set -e
F=${1:?file under gritlm_amd/csrc}; REV=${2:-HEAD}
cd "$(dirname "$0")/../.."
B=tools/ubench/_build/prev_obj; mkdir -p $B
git show $REV:gritlm_amd/csrc/$F > $B/$F
for f in gritlm_amd/csrc/*.hip; do
  n=$(basename $f); src=$f; [ "$n" = "$F" ] && src=$B/$F
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Igritlm_amd/csrc -Iinclude -c $src -o $B/${n%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/_build/libgritlm_hip_prev.so $B/*.o
echo built tools/ubench/_build/libgritlm_hip_prev.so "($F @ $REV)"
