#!/usr/bin/env python3
"""Time the attention backward (delta + dK/dV + dQ launches) of whatever library GRIT_HIP_LIB points at, at the contrastive step's chunk
shape (32 x 512, packed) and at 8 x 2048; median of 7 rounds of 3 calls; a checksum of dq | dk | dv for same-bits A/B runs (same-box A/B of
two builds: run it twice, tools/ubench/build_prev_lib.sh attention_bwd.hip <rev>)."""
import hashlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gritlm_amd import ops
NQ, NKV, D = 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(5)
out = {"lib": os.environ.get("GRIT_HIP_LIB", "shipped")}
for name, B, S in (("B32_S512", 32, 512), ("B8_S2048", 8, 2048)):
    T = B * S
    qkv = torch.randn((T, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
    dout = torch.randn((T, NQ * D), generator=g, device="cuda").to(torch.bfloat16)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device="cuda")
    o = torch.empty((T, NQ * D), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((T, NQ), dtype=torch.float32, device="cuda")
    ops.attn_bidir_varlen(qkv, cu, S, NQ, NKV, D, out=o, lse=lse)
    dq = torch.empty_like(qkv)
    for _ in range(2):
        ops.attn_bidir_varlen_bwd(qkv, cu, S, o, dout, lse, NQ, NKV, D, dqkv=dq)
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.attn_bidir_varlen_bwd(qkv, cu, S, o, dout, lse, NQ, NKV, D, dqkv=dq)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 3)
    m = sorted(ms)[3]
    fl = 10.0 * B * NQ * S * S * D
    out[name] = {"ms": m, "tflops": fl / m / 1e9, "frac_of_2500": fl / m / 1e9 / 2500.0,
                 "sha16": hashlib.sha256(dq.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]}
print(json.dumps(out))
