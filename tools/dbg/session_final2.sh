#!/bin/bash
# final verification of the shipped tree: GPU checks, smoke(), the driver-style bench line, the contrastive step under rocprofv3
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/final2; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_final.json 2> $O/bench_final.err
RAW=/tmp/prof_raw_c; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats -d $RAW -o bench --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --pairs 256 --chunk 32 > $O/bench_contrastive.json 2> $O/bench_contrastive.err
cp $(find $RAW -name "*kernel_stats.csv" | head -1) $O/bench_contrastive_kernel_stats.csv
tail -2 $O/pytest.log; tail -1 $O/smoke.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/final2/bench_final.json'))
c=d.get('contrastive',{})
print('docs/s',d['value'],'frac',d['roofline']['frac'],'stale',d['roofline'].get('traffic_stale'),'vendor',d['roofline'].get('vendor_gemm_tflops_same_shapes_no_epilogue',{}).get('flop_weighted'),'torch',d.get('rocm_torch_baseline',{}).get('value'))
print('attn',d['kernels']['attn_bidir_fwd']['tflops'],'pairs/s',c.get('value'),'frac',c.get('mfma_roofline_frac'),'ms',c.get('ms_per_step'))
PY
