#!/bin/bash
# A/B of the attention forward's wave-priority scheme on one box (round 5): ATT_PRIO_MODE builds of csrc/attention.hip, loaded through
# GRIT_HIP_LIB and timed by tools/attn_time.py (B 256 x S 512, B 64 x S 2048, B 16 x S 8192, packed ragged).
#   build (CPU, here):  bash tools/attn_prio_variants.sh build      run (GPU box):  bash tools/attn_prio_variants.sh run > gpurun_out/attn_prio.log
cd "$(dirname "$0")/.."
B=tools/ubench/_build/attn_prio
MODES="${MODES:-1 2 3 4 5}"
if [ "$1" = build ]; then
  mkdir -p $B
  make -C gritlm_amd/csrc -j8 >/dev/null || exit 1
  for m in $MODES; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -DATT_PRIO_MODE=$m $EXTRA -c gritlm_amd/csrc/attention.hip -o $B/attention_p$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libgritlm_hip_p$m.so $B/attention_p$m.o $(ls gritlm_amd/csrc/*.o | grep -v '/attention.o') || exit 1
  done
  ls -la $B; exit 0
fi
run() { echo "== $1"; shift; env "$@" python tools/attn_time.py 2>&1 | tail -1; }
run "shipped (mode 0: s_setprio 1 over QK in both workgroups)" A=1
for m in $MODES; do run "ATT_PRIO_MODE=$m" GRIT_HIP_LIB=$PWD/$B/libgritlm_hip_p$m.so; done
run "shipped again" A=1
