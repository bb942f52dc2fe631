#!/bin/bash
# final evidence of the round: rocprofv3 kernel stats + PMC passes (tools/profile_round.sh), then the driver-style bench line
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -3 gpurun_out/profile_round.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_final.json'))
c=d.get('contrastive',{})
print('docs/s',d['value'],'frac',d['roofline']['frac'],'vendor',d['roofline'].get('vendor_gemm_tflops_same_shapes_no_epilogue',{}).get('flop_weighted'),'torch',d.get('rocm_torch_baseline',{}).get('value'))
print('attn',d['kernels']['attn_bidir_fwd'])
print('pairs/s',c.get('value'),'frac',c.get('mfma_roofline_frac'),'ms',c.get('ms_per_step'),'peak',c.get('peak_hbm_gib'))
print('parity',d.get('parity_full_depth'))
PY
