#!/usr/bin/env python3
"""Where a W64 attention iteration spends its cycles: runs the -DW64_STAMPS build (tools/ubench/build_w64_stamps.sh, loaded through
GRIT_HIP_LIB) on B 256 x S 512 and B 16 x S 8192 and prints shader cycles per tile iteration and section for wave 0 of workgroup 0.
    GRIT_HIP_LIB=tools/ubench/_build/libgritlm_hip_w64stamps.so python tools/attn_w64_stamps.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gritlm_amd import ops  # noqa: E402

os.environ["GRIT_ATTN_FWD"] = "w64"

NQ, NKV, D = 32, 8, 128
names = ["dma_wait", "barrier", "top", "qk_and_exp", "pv_and_stats", "block_epilogue_and_loop"]
out = {}
for B, S in ((256, 512), (16, 8192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn((B * S, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    lse = torch.zeros((B, NQ, S), dtype=torch.float32, device="cuda")
    for _ in range(3):
        ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, lse=lse)
    torch.cuda.synchronize()
    v = lse.view(-1)[:7].cpu().tolist()
    n = v[6]
    out[f"B{B}_S{S}"] = {"iterations": n, **{k: x / n for k, x in zip(names, v[:6])}, "total_per_iteration": sum(v[:6]) / n}
print(json.dumps(out, indent=1))
