#!/usr/bin/env python3
"""Time the DEFAULT bidirectional attention forward of whatever library GRIT_HIP_LIB points at (same-box A/B of two builds: run it
twice).  TFLOP/s on B 256 x S 512, B 64 x S 2048, B 16 x S 8192, packed ragged; median of 7 rounds of 3 launches."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gritlm_amd import ops
NQ, NKV, D = 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(5)
out = {"lib": os.environ.get("GRIT_HIP_LIB", "shipped")}
def t(fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ms = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); fn(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1) / 3)
    m = sorted(ms)[3]
    return {"ms": m, "tflops": flops / (m * 1e-3) / 1e12}
for name, B, S in (("B256_S512", 256, 512), ("B64_S2048", 64, 2048), ("B16_S8192", 16, 8192)):
    qkv = torch.randn((B * S, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    o = torch.empty((B * S, NQ * D), dtype=torch.bfloat16, device="cuda")
    out[name] = t(lambda: ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=o), 4.0 * B * NQ * S * S * D)
    del qkv, o
lens = torch.randint(64, 513, (256,), generator=torch.Generator().manual_seed(3), dtype=torch.int32).cuda()
cu = torch.zeros((257,), dtype=torch.int32, device="cuda"); cu[1:] = torch.cumsum(lens, 0)
T = int(cu[-1]); qkv = torch.randn((T, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
o = torch.empty((T, NQ * D), dtype=torch.bfloat16, device="cuda"); mx = int(lens.max())
out["packed_ragged"] = t(lambda: ops.attn_bidir_varlen(qkv, cu, mx, NQ, NKV, D, out=o), 4.0 * NQ * D * float((lens.double() ** 2).sum()))
print(json.dumps(out))
