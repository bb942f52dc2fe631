#!/usr/bin/env python3
"""Mixtral-8x7B-shape (GritLM-8x7B) bidirectional encode on one MI355X: docs/s at 256 docs x 512 tokens, random-init weights
generated on the device (90 GB bf16).  SURVEY §8 f1.  python tools/mixtral_bench.py [--layers 32] [--docs 256] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--docs", type=int, default=256)
ap.add_argument("--seq", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=a.layers, num_attention_heads=32, num_key_value_heads=8,
                    vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_local_experts=8, num_experts_per_tok=2)
t0 = time.perf_counter()
eng = MistralEncoderEngine.random_init(cfg, dev, seed=0)
torch.cuda.synchronize()
t_init = time.perf_counter() - t0
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(3, cfg.vocab_size, (a.docs, a.seq), generator=gen, device=dev)
mask = torch.ones((a.docs, a.seq), dtype=torch.int64, device=dev)
for _ in range(a.warmup):
    eng.encode_pooled(ids, mask, "mean", True)
torch.cuda.synchronize()
timer = ops.KernelTimer()
ops.set_timer(timer)
eng.record_routing = []
t0 = time.perf_counter()
for _ in range(a.steps):
    e = eng.encode_pooled(ids, mask, "mean", True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ops.set_timer(None)
ks = timer.summary()
counts = torch.stack([torch.bincount(r.reshape(-1).long(), minlength=8) for r in eng.record_routing[:a.layers]]).float()
docs_per_s = a.docs * a.steps / dt
flops_doc = eng.flops_per_token(a.seq) * a.seq
print(json.dumps({
    "metric": f"encoded docs/sec @ seq{a.seq} (Mixtral-8x7B shape, sparse MoE top-2)", "value": docs_per_s, "unit": "docs/s", "n_gpus": 1,
    "steps": a.steps, "ms_per_step": dt / a.steps * 1e3, "dtype": "bf16", "data": "synthetic, random-init weights",
    "config": {"workload": f"Mixtral-8x7B shape, {a.layers}L, 8 experts top-2, batch {a.docs} x seq{a.seq}, mean pool + normalise"},
    "model_flops_utilisation": docs_per_s * flops_doc / 2.5e15,
    "roofline": {"bound": "mfma", "kernel": "gemm_bf16_nt_k (grouped launches: expert GEMMs)", "peak": 2500.0, "unit": "TFLOP/s",
                 "achieved": ks["gemm_bf16_nt_grouped"]["work"] / (ks["gemm_bf16_nt_grouped"]["total_ms"] * 1e-3) / 1e12,
                 "frac": ks["gemm_bf16_nt_grouped"]["work"] / (ks["gemm_bf16_nt_grouped"]["total_ms"] * 1e-3) / 1e12 / 2500.0,
                 "whole_step_frac": docs_per_s * flops_doc / 2.5e15},
    "tokens_per_s": docs_per_s * a.seq, "weights_init_s": t_init, "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9,
    "expert_load_max_over_mean": float((counts.max(dim=1)[0] / counts.mean(dim=1)).mean()), "finite": bool(torch.isfinite(e).all()),
    "kernels": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"], 2), "tflops": v["work"] / (v["total_ms"] * 1e-3) / 1e12}
                for k, v in ks.items()}}))
