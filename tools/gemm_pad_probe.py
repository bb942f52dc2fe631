#!/usr/bin/env python3
"""(Result: no effect beyond run-order noise -- +2-3 % in this probe, where the unpadded case runs first, 0 % in bench.py with padded
weights; not adopted.)  Does the power-of-two row stride (K*2 bytes = 8 KiB) of the GEMM operands cost bandwidth (L2 / MALL channel camping)?
Times the QKV-shaped GEMM with lda = ldw = K and with padded leading dimensions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gritlm_amd import ops  # noqa: E402

BF = torch.bfloat16
M = int(os.environ.get("MB_M", 131072))
for (N, K) in ((6144, 4096), (4096, 14336)):
    for pad_a, pad_w in ((0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (192, 192), (32, 32)):
        A = torch.randn((M, K + pad_a), device="cuda", dtype=torch.float32).to(BF)
        W = (torch.randn((N, K + pad_w), device="cuda", dtype=torch.float32) * 0.02).to(BF)
        a, w = A[:, :K], W[:, :K]
        out = torch.empty((M, N), device="cuda", dtype=BF)
        for _ in range(2):
            ops.gemm_nt(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_nt(a, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"N={N} K={K} lda=K+{pad_a} ldw=K+{pad_w}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
        del A, W, a, w, out
