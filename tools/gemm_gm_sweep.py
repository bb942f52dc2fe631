import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gritlm_amd import ops
from gritlm_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU
M = 131072
for (N, K, epi, tag) in ((28672, 4096, EPI_SWIGLU, "gate_up"), (4096, 14336, EPI_RESIDUAL, "down"), (4096, 4096, EPI_RESIDUAL, "o_proj"), (6144, 4096, EPI_STORE, "qkv")):
    a = torch.randn((M, K), device="cuda").to(torch.bfloat16); w = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
    out = torch.empty((M, N // 2 if epi == EPI_SWIGLU else N), device="cuda", dtype=torch.bfloat16)
    res = torch.randn((M, N), device="cuda").to(torch.bfloat16) if epi == EPI_RESIDUAL else None
    for _ in range(2): ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res)
    torch.cuda.synchronize(); ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"GM={os.environ.get('GRIT_GEMM_GM','4')} {tag}: {ts[3]:.3f} ms {2.0*M*N*K/ts[3]/1e9:.0f} TF/s", flush=True)
    del a, w, out, res
