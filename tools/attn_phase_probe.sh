#!/bin/bash
# build (CPU): bash tools/attn_phase_probe.sh build      run (GPU): bash tools/attn_phase_probe.sh run > gpurun_out/attn_phase.log
cd "$(dirname "$0")/.."
B=tools/ubench/_build/attn_timing
if [ "$1" = build ]; then
  mkdir -p $B
  make -C gritlm_amd/csrc -j8 >/dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -DATT_TIMING $EXTRA -c gritlm_amd/csrc/attention.hip -o $B/attention_t.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgritlm_hip_timing.so $B/attention_t.o $(ls gritlm_amd/csrc/*.o | grep -v '/attention.o') || exit 1
  exit 0
fi
for shape in "256 512" "64 2048"; do GRIT_HIP_LIB=$PWD/$B/libgritlm_hip_timing.so python tools/attn_phase_probe.py $shape; done
