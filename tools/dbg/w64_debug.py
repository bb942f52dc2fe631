#!/usr/bin/env python3
"""Debug aid for the W64 attention forward: where do its outputs differ from the default kernel?  Prints, per case, the maximum |diff| by
(32-row query group) x (32-column d-block) and by head, and whether LSE agrees."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gritlm_amd import ops
D = 128
g = torch.Generator(device="cuda").manual_seed(7)
for (B, S, nq, nkv) in [(1, 64, 4, 4), (1, 128, 4, 4), (1, 256, 4, 4), (1, 512, 8, 2)]:
    qkv = torch.randn((B * S, (nq + 2 * nkv) * D), generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    res = {}
    for which in ("v3", "w64"):
        if which == "w64": os.environ["GRIT_ATTN_FWD"] = "w64"
        else: os.environ.pop("GRIT_ATTN_FWD", None)
        out = torch.zeros((B * S, nq * D), dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros((B, nq, S), dtype=torch.float32, device="cuda")
        ops.attn_bidir(qkv, bits, B, S, nq, nkv, D, out=out, lse=lse)
        torch.cuda.synchronize()
        res[which] = (out.float(), lse)
    d = (res["v3"][0] - res["w64"][0]).abs().view(B * S, nq, D)
    print(f"--- B{B} S{S} nq{nq} nkv{nkv}: max |dO| {float(d.max()):.4f}  max |dLSE| {float((res['v3'][1]-res['w64'][1]).abs().max()):.4g}")
    print(" by head:", [round(float(d[:, h].max()), 3) for h in range(nq)])
    print(" by 32-row group:", [round(float(d[r:r + 32].max()), 3) for r in range(0, B * S, 32)])
    print(" by 32-col d-block:", [round(float(d[:, :, c:c + 32].max()), 3) for c in range(0, D, 32)])
    print(" by 8-col block (head 0):", [round(float(d[:, 0, c:c + 8].max()), 2) for c in range(0, D, 8)])
    dl = (res["v3"][1] - res["w64"][1]).abs()[0]
    print(" dLSE by head x 32-row group:", [[round(float(dl[h, r:r + 32].max()), 3) for r in range(0, S, 32)] for h in range(nq)])
os.environ.pop("GRIT_ATTN_FWD", None)
