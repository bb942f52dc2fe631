#!/usr/bin/env python3
"""Decode-step microbenchmark at the GritLM-7B shape: ms per generated token on top of a cached prefix (native decoder, HIP graph).
python tools/decode_bench.py [--prefix 2048 --new 128 --batch 1 --precision bf16|f16_operands|f16_stream] [--experts 8: the Mixtral-8x7B
shape -- sparse-MoE decode, two of the eight experts' weights streamed per row and layer]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd.decoder import MistralDecoder  # noqa: E402
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prefix", type=int, default=2048)
ap.add_argument("--new", type=int, default=128)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--precision", default="bf16", help="engine policy; the decoder follows it (fp16 operands under f16_operands / f16_stream)")
ap.add_argument("--experts", type=int, default=0, help="sparse-MoE layers with this many experts (top-2): the Mixtral-8x7B shape at 8")
ap.add_argument("--no-graph", action="store_true", help="eager launches (rocprofv3 --pmc cannot follow HIP-graph launches)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=a.layers, num_attention_heads=32, num_key_value_heads=8, vocab_size=32000,
                    **(dict(num_local_experts=a.experts, num_experts_per_tok=2, rope_theta=1e6) if a.experts else {}))
eng = MistralEncoderEngine.random_init(cfg, dev, seed=0)
lm_head = (torch.randn((32000, 4096), device=dev) * 0.02).to(torch.bfloat16)
dec = MistralDecoder(eng, lm_head)
if a.no_graph:
    dec.use_graph = False
g = torch.Generator(device=dev).manual_seed(3)
doc = torch.randint(3, 32000, (a.batch, a.prefix), generator=g, device=dev)
eng.set_precision(a.precision)
_, kv = eng.forward(doc, torch.ones_like(doc), return_kv=True, kv_dtype=None)
q = torch.randint(3, 32000, (a.batch, 4), generator=g, device=dev)
dec.generate(q, 8, past_key_values=kv)
torch.cuda.synchronize()
# ms per token = the slope between a run of n and a run of 3n new tokens (prefill of the query and launch set-up cancel); the median of three
# such pairs: one slow first run (a box hiccup) used to show up as an impossibly FAST token rate
slopes, raw = [], []
for _ in range(3):
    res = {}
    for n in (a.new, 3 * a.new):
        t0 = time.perf_counter()
        dec.generate(q, n, past_key_values=kv)
        torch.cuda.synchronize()
        res[n] = time.perf_counter() - t0
    slopes.append((res[3 * a.new] - res[a.new]) / (2 * a.new) * 1e3)
    raw.append([res[a.new], res[3 * a.new]])
ms = sorted(slopes)[1]
if a.experts:      # per token and row: attention weights + the router + TWO experts' w13 / w2 per layer
    wbytes = (sum(L.wqkv.numel() + L.wo.numel() + L.wgate.numel() + 2 * (L.w13[0].numel() + L.w2[0].numel()) for L in eng.layers) + lm_head.numel()) * 2
else:
    wbytes = (sum(sum(getattr(L, k).numel() for k in ("wqkv", "wo", "wgu", "wdown")) for L in eng.layers) + lm_head.numel()) * 2
print(json.dumps({"metric": "native decode ms per token (" + (f"8x7B shape, {a.experts} experts top-2" if a.experts else "7B shape") + ")", "precision": a.precision, "decode_arithmetic": dec.last_precision, "ms_per_token": ms, "tokens_per_s": a.batch * 1e3 / ms, "batch": a.batch,
                  "prefix": a.prefix, "ms_per_token_runs": slopes, "raw_s_n_3n": raw, "weight_gb_per_token": wbytes / 1e9, "hbm_roofline_ms": wbytes / 8e12 * 1e3, "frac_of_hbm_roofline": wbytes / 8e12 * 1e3 / ms}))
