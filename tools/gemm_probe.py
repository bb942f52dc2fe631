#!/usr/bin/env python3
"""Launch each GritLM-7B GEMM shape a few times (target of the rocprofv3 --pmc passes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gritlm_amd import ops  # noqa: E402
from gritlm_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU  # noqa: E402

M = int(os.environ.get("MB_M", 131072))
reps = int(os.environ.get("REPS", 3))
BF = torch.bfloat16
for (N, K, epi) in ((6144, 4096, EPI_STORE), (4096, 4096, EPI_RESIDUAL), (28672, 4096, EPI_SWIGLU), (4096, 14336, EPI_RESIDUAL)):
    a = torch.randn((M, K), device="cuda", dtype=torch.float32).to(BF)
    w = (torch.randn((N, K), device="cuda", dtype=torch.float32) * 0.02).to(BF)
    out = torch.empty((M, N // 2 if epi == EPI_SWIGLU else N), device="cuda", dtype=BF)
    res = torch.randn((M, N), device="cuda", dtype=torch.float32).to(BF) if epi == EPI_RESIDUAL else None
    if N == 6144:      # the model's QKV projection runs with the RoPE epilogue (7B: 32 q + 8 k heads rotated, 8 v heads stored)
        from gritlm_amd.encoder import rope_tables
        cos, sin = rope_tables(512, 128, 1e4, True, "cuda")
    for _ in range(reps):
        if N == 6144:
            ops.gemm_nt_rope(a, w, cos, sin, 5120, S=512, out=out)
        else:
            ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res)
    torch.cuda.synchronize()
    del a, w, out, res
print("done")
