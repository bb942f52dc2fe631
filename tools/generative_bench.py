#!/usr/bin/env python3
"""Generative branch at the GritLM-7B shape on one MI355X: tokens/s of one training step =
causal forward (saved activations) + lm_head + fused CE + backward + AdamW.   python tools/generative_bench.py [--bs 8 --seq 2048]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd.encoder import EncoderConfig  # noqa: E402
from gritlm_amd.training.engine import MistralTrainEngine, SyntheticBackbone  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=8)
ap.add_argument("--seq", type=int, default=2048)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=a.layers, num_attention_heads=32, num_key_value_heads=8,
                    vocab_size=32000)
bb = SyntheticBackbone(cfg, dev, seed=1)
lm_head = torch.nn.Linear(4096, 32000, bias=False, device=dev, dtype=torch.bfloat16)
eng = MistralTrainEngine(bb, cfg, dev, lm_head=lm_head)
eng.cache_transposed_weights = True
opt = torch.optim.AdamW(list(bb.parameters()) + list(lm_head.parameters()), lr=1e-5, fused=True)
gen = torch.Generator(device=dev).manual_seed(7)
ids = torch.randint(3, cfg.vocab_size, (a.bs, a.seq), generator=gen, device=dev)
lens = torch.randint(a.seq // 2, a.seq + 1, (a.bs,), generator=gen, device=dev); lens[0] = a.seq
mask = (torch.arange(a.seq, device=dev).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
labels = ids.clone(); labels[mask == 0] = -100; labels[:, :64] = -100


def step():
    loss, st = eng.forward_lm(ids, mask, labels, "mixed", 1.0)
    eng.backward_lm(st, 1.0)
    opt.step(); opt.zero_grad(set_to_none=True); eng.weights_updated()
    return loss


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
tokens = int(mask.sum().item())
gemm = 2.0 * a.layers * (4096 * 6144 + 4096 * 4096 + 3 * 4096 * 14336) + 2.0 * 4096 * 32000
attn = 4.0 * a.layers * 4096 * float((lens.float() ** 2).sum().item()) / 2 / tokens          # causal: half of the S^2 products
print(json.dumps({"metric": "generative training tokens/sec (causal fwd + lm_head + CE + bwd + AdamW)", "value": tokens / dt, "unit": "tokens/s",
                  "ms_per_step": dt * 1e3, "batch": a.bs, "seq": a.seq, "real_tokens": tokens, "layers": a.layers, "loss": float(loss),
                  "mfma_roofline_frac": 3.0 * (gemm + attn) * tokens / dt / 2.5e15, "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9}))
