#!/usr/bin/env python3
"""GEMM A/B under the conditions of the model, not of a microbenchmark: the four GEMM shapes of a GritLM-7B layer at M = 131072, walked
layer by layer over DISTINCT weight sets (L layers x 436 MB: nothing stays in the 256 MB Infinity Cache from one use to the next),
sustained for seconds (the chip settles at its power-limited clock), passes alternating between this repository's kernel (with its fused
RoPE / residual / SwiGLU epilogues) and the vendor GEMM behind ``torch.matmul`` (hipBLASLt, plain store) on the SAME operands.
bench.py's ``vendor_gemm_tflops_same_shapes_no_epilogue`` times 5 back-to-back launches of one shape (weights warm in the Infinity
Cache, chip cool): this is the like-for-like number.   python tools/gemm_sustained_ab.py [--layers 16] [--passes 3] [--zeros]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402
from gritlm_amd._lib import EPI_RESIDUAL, EPI_SWIGLU  # noqa: E402
from gritlm_amd.encoder import rope_tables  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=16)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--m", type=int, default=131072)
ap.add_argument("--zeros", action="store_true", help="zero-filled operands (no data toggling: the structure's ceiling, not a result)")
a = ap.parse_args()
dev, BF = "cuda", torch.bfloat16
M, H, I, NQKV = a.m, 4096, 14336, 6144
g = torch.Generator(device=dev).manual_seed(0)
mk = (lambda *s: torch.zeros(s, device=dev, dtype=BF)) if a.zeros else \
    (lambda *s: (torch.randn(s, generator=g, device=dev, dtype=torch.float32) * (0.02 if s[0] != M else 1.0)).to(BF))
layers = [dict(qkv=mk(NQKV, H), o=mk(H, H), gu=mk(2 * I, H), down=mk(H, I)) for _ in range(a.layers)]
x, ctx, act, res = mk(M, H), mk(M, H), mk(M, I), mk(M, H)
o_qkv, o_h, o_act, o_gu = torch.empty((M, NQKV), device=dev, dtype=BF), torch.empty((M, H), device=dev, dtype=BF), \
    torch.empty((M, I), device=dev, dtype=BF), torch.empty((M, 2 * I), device=dev, dtype=BF)
cos, sin = rope_tables(512, 128, 10000.0, True, dev)
shapes = {"qkv": 2.0 * M * NQKV * H, "o": 2.0 * M * H * H, "gate_up": 2.0 * M * 2 * I * H, "down": 2.0 * M * H * I}
ev = {k: {n: [] for n in shapes} for k in ("ours", "vendor")}


def timed(kind, name, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    ev[kind][name].append((e0, e1))


def one_pass(kind):
    for L in layers:
        if kind == "ours":
            timed(kind, "qkv", lambda: ops.gemm_nt_rope(x, L["qkv"], cos, sin, 40 * 128, S=512, out=o_qkv))
            timed(kind, "o", lambda: ops.gemm_nt(ctx, L["o"], out=o_h, epilogue=EPI_RESIDUAL, residual=res))
            timed(kind, "gate_up", lambda: ops.gemm_nt(x, L["gu"], out=o_act, epilogue=EPI_SWIGLU))
            timed(kind, "down", lambda: ops.gemm_nt(act, L["down"], out=o_h, epilogue=EPI_RESIDUAL, residual=res))
        else:
            timed(kind, "qkv", lambda: torch.matmul(x, L["qkv"].t(), out=o_qkv))
            timed(kind, "o", lambda: torch.matmul(ctx, L["o"].t(), out=o_h))
            timed(kind, "gate_up", lambda: torch.matmul(x, L["gu"].t(), out=o_gu))
            timed(kind, "down", lambda: torch.matmul(act, L["down"].t(), out=o_h))


one_pass("ours"); one_pass("vendor")                # warm-up (also brings the chip to its sustained state)
for k in ev:
    for n in ev[k]:
        ev[k][n].clear()
for _ in range(a.passes):
    one_pass("ours"); one_pass("vendor")
torch.cuda.synchronize()
out = {"operands": "zeros" if a.zeros else "random", "layers_of_distinct_weights": a.layers, "passes": a.passes, "M": M}
tot = {}
for k in ev:
    tf, tt = 0.0, 0.0
    for n, fl in shapes.items():
        ms = [p.elapsed_time(q) for p, q in ev[k][n]]
        out[f"{k}_{n}_tflops"] = fl * len(ms) / (sum(ms) * 1e-3) / 1e12
        tf += fl * len(ms); tt += sum(ms) * 1e-3
    out[f"{k}_flop_weighted_tflops"] = tf / tt / 1e12
    tot[k] = tf / tt / 1e12
out["ours_over_vendor"] = tot["ours"] / tot["vendor"]
print(json.dumps(out))
