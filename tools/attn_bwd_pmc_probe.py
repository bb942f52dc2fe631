#!/usr/bin/env python3
"""Workload of tools/attn_bwd_pmc.sh (round 6): the attention backward (attn_delta_k, attn_bwd_dkdv_k, attn_bwd_dq_k) at the shapes the
contrastive step launches it with -- one GradCache chunk, 32 sequences x 512 tokens, 32 / 8 heads, packed (varlen) and padded -- and at
B 8 x S 2048; 4 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402

NQ, NKV, D = 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(3)
for B, S in ((32, 512), (8, 2048)):
    T = B * S
    qkv = torch.randn((T, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
    dout = torch.randn((T, NQ * D), generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device="cuda")
    out = torch.empty((T, NQ * D), dtype=torch.bfloat16, device="cuda")
    lse_p = torch.empty((B, NQ, S), dtype=torch.float32, device="cuda")
    lse_v = torch.empty((T, NQ), dtype=torch.float32, device="cuda")
    ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=out, lse=lse_p)
    ops.attn_bidir_varlen(qkv, cu, S, NQ, NKV, D, out=out, lse=lse_v)
    for _ in range(4):
        ops.attn_bidir_varlen_bwd(qkv, cu, S, out, dout, lse_v, NQ, NKV, D)
    for _ in range(4):
        ops.attn_bidir_bwd(qkv, bits, out, dout, lse_p, B, S, NQ, NKV, D)
    torch.cuda.synchronize()
    del qkv, dout, out
