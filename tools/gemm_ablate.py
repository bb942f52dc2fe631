import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gritlm_amd import ops
M, N, K = 131072, 6144, 4096
a = torch.randn((M, K), device="cuda").to(torch.bfloat16); w = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(a, w, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm_nt(a, w, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"ABLATE={os.environ.get('GRIT_GEMM_ABLATE','0')} VARIANT={os.environ.get('GRIT_GEMM_VARIANT','1')}: med {ts[5]:.3f} ms  -> {2.0*M*N*K/ts[5]/1e9:.0f} TF/s-equivalent")
