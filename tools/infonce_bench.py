#!/usr/bin/env python3
"""The loss step every rank executes at BASELINE configs[2] on 8 GPUs, on ONE GPU: q [2048,4096] x p [16384,4096] fp32 (the gathered
global batch), similarity + cross entropy + gradients of the rank's own rows (256 q rows, 2048 p rows) -- gritlm/training/model.py:36-64.
Also the 1-rank shape (256 x 2048).  Prints one JSON line per shape: ms, TF/s on the exact-f32 matrix pipe (157 TF peak), and torch
(matmul + cross_entropy + autograd, fp32) on the same GPU.   python tools/infonce_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402

F32_MFMA_PEAK_TF = 157.3


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def case(world, bq=256, group=8, H=4096, tau=0.02, rank=0):
    dev = "cuda"
    Nq, Np = world * bq, world * bq * group
    g = torch.Generator(device=dev).manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn((Nq, H), generator=g, device=dev), dim=-1)
    p = torch.nn.functional.normalize(torch.randn((Np, H), generator=g, device=dev), dim=-1)
    q_off, p_off = rank * bq, rank * bq * group
    med, best = timeit(lambda: ops.infonce(q, p, tau, q_off, bq, p_off, bq * group))
    flops = 2.0 * H * (Nq * Np + bq * Np + bq * group * Nq)
    loss, dq, dp = ops.infonce(q, p, tau, q_off, bq, p_off, bq * group)

    def ref():
        ql = q[q_off:q_off + bq].clone().requires_grad_(); pl = p[p_off:p_off + bq * group].clone().requires_grad_()
        qa = torch.cat([q[:q_off], ql, q[q_off + bq:]]); pa = torch.cat([p[:p_off], pl, p[p_off + bq * group:]])
        l = torch.nn.functional.cross_entropy(qa @ pa.t() / tau, torch.arange(Nq, device=dev) * group)
        l.backward()
        return l, ql.grad, pl.grad
    rmed, _ = timeit(ref, iters=5, warm=1)
    l, gq, gp = ref()
    return {"shape": f"world {world}: q [{Nq},{H}] x p [{Np},{H}] fp32, local rows q {bq} / p {bq * group}", "ms": med, "ms_best": best,
            "tflops": flops / med / 1e9, "frac_of_f32_mfma_peak": flops / med / 1e9 / F32_MFMA_PEAK_TF,
            "torch_fp32_ms": rmed, "loss": float(loss.item()), "torch_loss": float(l.item()),
            "dq_max_abs_diff_vs_torch": float((dq - gq).abs().max().item()), "dp_max_abs_diff_vs_torch": float((dp - gp).abs().max().item())}


if __name__ == "__main__":
    for w in (8, 1):
        print(json.dumps(case(w)), flush=True)
