#!/usr/bin/env python3
"""Workload of tools/attn_pmc_round.sh: the bidirectional attention forward, 4 launches each of the bf16 and the fp16
instantiation (different kernel names in the trace), at B 256 x S 512 and at B 32 x S 2048 (the two shapes differ in Grid_Size: 2097152 / 1048576 threads).
(Round 4 also launched the W64 prototype here; it left the library in round 5: tools/ubench/w64_variants/.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402

NQ, NKV, D = 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(2)
for B, S in ((256, 512), (32, 2048)):
    base = torch.randn((B * S, (NQ + 2 * NKV) * D), generator=g, device="cuda")
    bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
    for dt in (torch.bfloat16, torch.float16):
        qkv = base.to(dt)
        out = torch.empty((B * S, NQ * D), dtype=dt, device="cuda")
        for _ in range(4):
            ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=out)
        torch.cuda.synchronize()
        del qkv, out
    del base
