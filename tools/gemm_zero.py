import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gritlm_amd import ops
M, N, K = 131072, 6144, 4096
for fill in ("randn", "zeros", "ones_small"):
    if fill == "randn":
        a = torch.randn((M, K), device="cuda").to(torch.bfloat16); w = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
    elif fill == "zeros":
        a = torch.zeros((M, K), device="cuda", dtype=torch.bfloat16); w = torch.zeros((N, K), device="cuda", dtype=torch.bfloat16)
    else:
        a = torch.full((M, K), 0.5, device="cuda", dtype=torch.bfloat16); w = torch.full((N, K), 0.25, device="cuda", dtype=torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for name, fn in (("ours", lambda: ops.gemm_nt(a, w, out=out)), ("torch", lambda: torch.matmul(a, w.t()))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{fill:10s} {name:6s} med {ts[5]:.3f} ms -> {2.0*M*N*K/ts[5]/1e9:.0f} TF/s", flush=True)
