#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the C-ABI shim (the argument validation, workspace arithmetic and launch set-up of every
# grit_* entry point; device code is NOT instrumented: -fno-gpu-sanitize) and the ABI tests run against it.  No GPU needed: the ABI tests
# only make calls that must be rejected before anything is launched.
#   bash tools/asan_host_shim.sh [pytest args]      -> tools/_asan/libgritlm_hip_asan.so, exit code of pytest
#   bash tools/asan_host_shim.sh --fuzz [seed] [n]   -> tools/fuzz_abi.py (random arguments for every entry point) on the same build
set -e
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/tools/_asan; mkdir -p $OUT
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so 2>/dev/null | head -1)
[ -n "$RT" ] || { echo "no shared ASAN runtime under /opt/rocm/lib/llvm"; exit 3; }
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -Wno-unused-variable -Wno-unused-function"
pids=()
for f in $R/gritlm_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ ! -f $OUT/$b.o ] || [ $f -nt $OUT/$b.o ] || [ $R/gritlm_amd/csrc/common.h -nt $OUT/$b.o ] || [ $R/include/gritlm_hip.h -nt $OUT/$b.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -I$R/gritlm_amd/csrc -c $f -o $OUT/$b.o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o $OUT/libgritlm_hip_asan.so $OUT/*.o
ldd $OUT/libgritlm_hip_asan.so | grep -q asan || { echo "the library is not linked against the ASAN runtime"; exit 4; }
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 GRIT_HIP_LIB=$OUT/libgritlm_hip_asan.so
if [ "$1" = "--fuzz" ]; then shift; exec python tools/fuzz_abi.py "$@"; fi
python -m pytest tests/test_abi.py -q -p no:cacheprovider "$@"
