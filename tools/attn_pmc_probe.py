#!/usr/bin/env python3
"""Workload of tools/attn_pmc_r04.sh: the default bidirectional attention forward and the opt-in W64 kernel, 4 launches each, at B 256 x
S 512 and at B 16 x S 8192 (the two kernels have different names in the trace; the two shapes differ in Grid_Size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402

NQ, NKV, D = 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(2)
for B, S in ((256, 512), (16, 8192)):
    qkv = torch.randn((B * S, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    out = torch.empty((B * S, NQ * D), dtype=torch.bfloat16, device="cuda")
    for which in (None, "w64"):
        if which:
            os.environ["GRIT_ATTN_FWD"] = which
        else:
            os.environ.pop("GRIT_ATTN_FWD", None)
        for _ in range(4):
            ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=out)
        torch.cuda.synchronize()
    del qkv, out
os.environ.pop("GRIT_ATTN_FWD", None)
