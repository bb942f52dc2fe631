#!/bin/bash
# last verification of the shipped tree: attention forward / backward bit-identity against the kernels of commit a71e37f, GPU checks, encode bench
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/final4; mkdir -p $O
( ATTN_OLD=tools/ubench/_r01/libattn_old.so ATTN_NEW=gritlm_amd/libgritlm_hip.so timeout 300 tools/ubench/attn_ab.bin all ) > $O/attn_vs_round2.log 2>&1
( ATTN_OLD=tools/ubench/_var/libattn_bwd_base.so ATTN_NEW=gritlm_amd/libgritlm_hip.so timeout 300 tools/ubench/attn_bwd_ab.bin all ) > $O/attn_bwd_vs_round2.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-contrastive --no-torch-baseline ) > $O/bench_encode.json 2> $O/bench_encode.err
grep -E "old .* new|RESULT" $O/attn_vs_round2.log | cut -c1-170; grep -E "old .* new|RESULT" $O/attn_bwd_vs_round2.log | cut -c1-170; tail -2 $O/pytest.log
python -c "
import json;d=json.load(open('$O/bench_encode.json'));print(d['value'],d['roofline']['frac'],d['kernels']['attn_bidir_fwd'],d.get('ragged_batch'))"
