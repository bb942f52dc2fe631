#!/bin/bash
# same-box A/B of the library at the start of this half of round 3 (commit a71e37f, GRIT_HIP_LIB) against the shipped one
cd "$(dirname "$0")/../.."
O=gpurun_out/s12; mkdir -p $O
OLD=$PWD/tools/ubench/_var/full_r03a/libgritlm_hip.so
E="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-contrastive --no-ragged --no-torch-baseline"
C="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged --no-torch-baseline --pairs 128"
( timeout 300 $E ) > $O/enc_new_1.json 2> $O/err.log
( GRIT_HIP_LIB=$OLD timeout 300 $E ) > $O/enc_old_1.json 2>> $O/err.log
( timeout 400 $C ) > $O/con_new.json 2>> $O/err.log
( GRIT_HIP_LIB=$OLD GRIT_GRADCACHE_PASS1_MULT=1 timeout 400 $C ) > $O/con_old.json 2>> $O/err.log
( GRIT_HIP_LIB=$OLD timeout 300 $E ) > $O/enc_old_2.json 2>> $O/err.log
( timeout 300 $E ) > $O/enc_new_2.json 2>> $O/err.log
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d.get("contrastive")
    s="%s docs/s %.2f frac %.4f attn %.0f TF" % (sys.argv[1].split("/")[-1], d["value"], d["roofline"]["frac"], d["kernels"]["attn_bidir_fwd"]["tflops"])
    if c: s+=" | pairs/s %.4f frac %.4f ms %.0f" % (c["value"], c["mfma_roofline_frac"], c["ms_per_step"])
    print(s)
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $O/summary.txt
