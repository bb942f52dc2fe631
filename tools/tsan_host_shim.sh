#!/bin/bash
# Host-side ThreadSanitizer build of the C-ABI shim + tools/tsan_driver.cpp (8 threads calling into the library at once; no GPU needed).
#   bash tools/tsan_host_shim.sh     -> tools/_tsan/libgritlm_hip_tsan.so, tools/_tsan/tsan_driver; exit code 0 = no race, expected results
set -e
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/tools/_tsan; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan -Wno-unused-variable -Wno-unused-function"
pids=()
for f in $R/gritlm_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ ! -f $OUT/$b.o ] || [ $f -nt $OUT/$b.o ] || [ $R/gritlm_amd/csrc/common.h -nt $OUT/$b.o ] || [ $R/include/gritlm_hip.h -nt $OUT/$b.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -I$R/gritlm_amd/csrc -c $f -o $OUT/$b.o & pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=thread -fno-gpu-sanitize -shared-libsan -o $OUT/libgritlm_hip_tsan.so $OUT/*.o
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -shared-libsan -o $OUT/tsan_driver tools/tsan_driver.cpp -ldl -lpthread \
  -Wl,-rpath,$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1))
TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 $OUT/tsan_driver $OUT/libgritlm_hip_tsan.so
