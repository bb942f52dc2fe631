"""Do the decode GEMV forms give the same bits for a row whatever the number of rows in the launch (the NB instantiations)?  python tools/gemv_rows_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gritlm_amd import ops
from gritlm_amd._lib import EPI_RESIDUAL, EPI_SWIGLU, EPI_STORE
from gritlm_amd.encoder import swiglu_interleave
dev = "cuda"
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float16):
    K, N = 4096, 1024
    w = (torch.randn((N, K), device=dev) * 0.05).to(dt)
    wi = swiglu_interleave(w[:N // 2].contiguous(), w[N // 2:].contiguous())
    x = torch.randn((8, K), device=dev).to(dt)
    lnw = (1 + 0.1 * torch.randn((K,), device=dev)).to(torch.bfloat16)
    res = torch.randn((8, N), device=dev).to(torch.float32 if dt == torch.float16 else dt)
    for name, fn in (("store", lambda xx, rr: ops.gemv(xx, w)),
                     ("residual", lambda xx, rr: ops.gemv(xx, w, epilogue=EPI_RESIDUAL, residual=rr)),
                     ("swiglu", lambda xx, rr: ops.gemv(xx, wi, epilogue=EPI_SWIGLU)),
                     ("norm_store", lambda xx, rr: ops.rmsnorm_gemv(xx, lnw, 1e-5, w, deferred=True)),
                     ("norm_swiglu", lambda xx, rr: ops.rmsnorm_gemv(xx, lnw, 1e-5, wi, epilogue=EPI_SWIGLU, deferred=True))):
        full = fn(x, res)
        same = {}
        for nb in (1, 2, 3, 5):
            parts = torch.cat([fn(x[a:a + nb].contiguous(), res[a:a + nb].contiguous()) for a in range(0, 8, nb)])[:8]
            same[nb] = bool(torch.equal(parts, full))
        print(dt, name, same)
