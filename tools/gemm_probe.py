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
    for _ in range(reps):
        ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res)
    torch.cuda.synchronize()
    del a, w, out, res
print("done")
