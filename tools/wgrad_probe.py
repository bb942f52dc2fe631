"""Weight-gradient GEMM  g[N,K] += dy[T,N]^T x[T,K]  two ways at the four shapes of a 7B layer (T = 16384 = one GradCache chunk):
ours (two transposes + NT GEMM with the accumulate epilogue, engine._wgrad) vs torch addmm_ (hipBLASLt reads both operands as they lie)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gritlm_amd import ops
from gritlm_amd._lib import EPI_RESIDUAL
dev = torch.device("cuda:0")
T = 16384
def tm(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, K in (("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)):
    dy = torch.randn((T, N), device=dev).to(torch.bfloat16)
    x = torch.randn((T, K), device=dev).to(torch.bfloat16)
    g1 = torch.zeros((N, K), device=dev, dtype=torch.bfloat16); g2 = torch.zeros_like(g1)
    dyT = torch.zeros((N, T), device=dev, dtype=torch.bfloat16); xT = torch.zeros((K, T), device=dev, dtype=torch.bfloat16)
    def ours():
        ops.transpose(dy, out=dyT); ops.transpose(x, out=xT)
        ops.gemm_nt(dyT, xT, out=g1, epilogue=EPI_RESIDUAL, residual=g1)
    def vendor():
        g2.addmm_(dy.t(), x)
    t1, t2 = tm(ours), tm(vendor)
    fl = 2.0 * T * N * K
    g1.zero_(); g2.zero_(); ours(); vendor(); torch.cuda.synchronize()
    rel = float((g1.float() - g2.float()).norm() / g2.float().norm())
    print(f"{name:8s} ours {t1*1e3:8.1f} us ({fl/t1/1e9:6.0f} TF incl. transposes)   addmm_ {t2*1e3:8.1f} us ({fl/t2/1e9:6.0f} TF)   rel diff {rel:.2e}", flush=True)
