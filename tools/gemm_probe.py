#!/usr/bin/env python3
"""Launch each GritLM-7B GEMM shape a few times (target of the rocprofv3 --pmc passes).  PROBE_F16=1: the fp16-operand instantiations of
the f16_operands policy (RoPE / RESIDUAL_F32 / SWIGLU epilogues) instead of the bf16 ones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gritlm_amd import ops  # noqa: E402
from gritlm_amd._lib import EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_STORE, EPI_SWIGLU  # noqa: E402

M = int(os.environ.get("MB_M", 131072))
reps = int(os.environ.get("REPS", 3))
F16 = os.environ.get("PROBE_F16") == "1"
BF = torch.float16 if F16 else torch.bfloat16
RES = EPI_RESIDUAL_F32 if F16 else EPI_RESIDUAL
for (N, K, epi) in ((6144, 4096, EPI_STORE), (4096, 4096, RES), (28672, 4096, EPI_SWIGLU), (4096, 14336, RES)):
    a = torch.randn((M, K), device="cuda", dtype=torch.float32).to(BF)
    w = (torch.randn((N, K), device="cuda", dtype=torch.float32) * 0.02).to(BF)
    out = torch.empty((M, N // 2 if epi == EPI_SWIGLU else N), device="cuda", dtype=torch.float32 if epi == EPI_RESIDUAL_F32 else BF)
    res = None
    if epi == EPI_RESIDUAL:
        res = torch.randn((M, N), device="cuda", dtype=torch.float32).to(BF)
    elif epi == EPI_RESIDUAL_F32:
        res = out.normal_()                                  # in place: C = C + A W^T, as the engine's residual stream
    if N == 6144:      # the model's QKV projection runs with the RoPE epilogue (7B: 32 q + 8 k heads rotated, 8 v heads stored)
        from gritlm_amd.encoder import rope_tables
        cos, sin = rope_tables(512, 128, 1e4, not F16, "cuda")
    for _ in range(reps):
        if N == 6144:
            ops.gemm_nt_rope(a, w, cos, sin, 5120, S=512, out=out)
        else:
            ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res)
    torch.cuda.synchronize()
    del a, w, out, res
print("done")
