#!/usr/bin/env python3
"""Which GEMM of the contrastive (GradCache) step runs at what rate: one timed step of bench.py's contrastive leg at the 7B shape with the
live HIP-event timer on and the tags carrying M (forward, dgrad and wgrad launches of one weight have different (M, N, K)).
    GRIT_TIMER_TAG_M=1 python tools/contrastive_shapes.py [--pairs 64]"""
import argparse, json, os, sys
os.environ["GRIT_TIMER_TAG_M"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gritlm_amd import ops
from gritlm_amd.encoder import EncoderConfig
ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int, default=64); a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8, vocab_size=32000)
timer = ops.KernelTimer()
orig = torch.cuda.synchronize
state = {"n": 0}
res = None
# warm-up runs untimed inside contrastive_leg; the timer is switched on for the timed step only: the leg synchronises right after its warm-up
def sync_hook(*x, **k):
    orig(*x, **k)
    state["n"] += 1
    if state["n"] == 1:
        ops.set_timer(timer)
torch.cuda.synchronize = sync_hook
line = bench.contrastive_leg(cfg, dev, 1, 0, None, pairs=a.pairs, steps=1, warmup=1, ragged_pairs=0)
torch.cuda.synchronize = orig
ops.set_timer(None)
ks = timer.summary()
out = {"pairs": a.pairs, "pairs_per_s": line["value"], "mfma_roofline_frac": line["mfma_roofline_frac"], "kernels": {}}
tot = 0.0
for name, v in ks.items():
    tot += v["total_ms"]
    out["kernels"][name] = {"launches": v["launches"], "total_ms": round(v["total_ms"], 1), "tflops": v["work"] / (v["total_ms"] * 1e-3) / 1e12 if v["work"] else None}
    if "by_tag" in v:
        out["kernels"][name]["by_shape"] = {t: {"launches": d["launches"], "total_ms": round(d["total_ms"], 1), "tflops": round(d["work"] / (d["total_ms"] * 1e-3) / 1e12, 1)}
                                            for t, d in sorted(v["by_tag"].items(), key=lambda kv: -kv[1]["total_ms"])}
out["timed_kernels_total_ms"] = tot; out["step_ms"] = line["ms_per_step"]
print(json.dumps(out, indent=1))
