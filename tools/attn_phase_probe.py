#!/usr/bin/env python3
"""Where a workgroup of the attention forward spends its cycles (ATT_TIMING build of csrc/attention.hip, loaded through GRIT_HIP_LIB):
thread 0 of every workgroup stamps s_memtime at kernel entry, after the prologue wait, past the barrier of every KV tile, at the end of a
query block's tile loop, after the seam wait and after the block's output stores were issued -- written into the LSE rows of the launch.
    build: hipcc -DATT_TIMING ... (tools/attn_phase_probe.sh build)     run: GRIT_HIP_LIB=... python tools/attn_phase_probe.py [B S]"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gritlm_amd import ops
NQ, NKV, D = 32, 8, 128
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
g = torch.Generator(device="cuda").manual_seed(5)
qkv = torch.randn((B * S, (NQ + 2 * NKV) * D), generator=g, device="cuda").to(torch.bfloat16)
bits = ops.mask_pack(torch.ones((B, S), dtype=torch.int64, device="cuda"))
o = torch.empty((B * S, NQ * D), dtype=torch.bfloat16, device="cuda")
lse = torch.zeros((B, NQ, S), dtype=torch.float32, device="cuda")
for _ in range(3):
    lse.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=o, lse=lse); e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
st = lse.view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff            # [B, NQ, S] raw stamps
nqb = S // 128
qpw = min(4, nqb)
ngx = nqb // qpw
st = st.reshape(B * NQ * ngx, qpw * 128)
nt = S // 64
per_blk = nt + 3
n_expected = 5 + qpw * per_blk + 2
assert (st[:, n_expected - 1] == 0xffffffff).all(), "stamp layout mismatch"
rt0, hwid, xcc = st[:, 1], st[:, 2], st[:, 3]
rt1 = st[:, n_expected - 2]
cyc_total = st[:, 5 + qpw * per_blk - 1]
wall_ns = ((rt1 - rt0) & 0xffffffff) * 10.0
res = {"B": B, "S": S, "launch_ms": ms, "workgroups": int(st.shape[0]), "qpw": qpw, "tiles_per_block": nt,
       "wg_cycles_mean": float(cyc_total.mean()), "wg_wall_us_mean": float(wall_ns.mean() / 1e3),
       "effective_clock_ghz": float((cyc_total / wall_ns).mean()),
       "prologue_cycles": float(st[:, 4].mean())}
blk = st[:, 5:5 + qpw * per_blk].reshape(-1, qpw, per_blk)
tile_t = blk[:, :, :nt]
d_tiles = np.diff(tile_t, axis=2)                                                  # tile i -> i+1 (past-barrier to past-barrier)
res["tile_cycles_by_index_in_block_mean"] = [float(x) for x in d_tiles.mean(axis=(0, 1))]
res["tile_cycles_first_block_by_index"] = [float(x) for x in d_tiles[:, 0].mean(axis=0)]
res["last_tile_to_loop_end"] = float((blk[:, :, nt] - blk[:, :, nt - 1]).mean())
res["seam_wait"] = float((blk[:, :, nt + 1] - blk[:, :, nt]).mean())
res["output_stores_issue"] = float((blk[:, :, nt + 2] - blk[:, :, nt + 1]).mean())
if qpw > 1:
    res["stores_issued_to_next_block_first_tile"] = float((blk[:, 1:, 0] - blk[:, :-1, nt + 2]).mean())
res["first_tile_stamp_of_block0_minus_prologue"] = float((blk[:, 0, 0] - st[:, 4]).mean())
res["block_cycles_mean"] = [float((blk[:, i, nt + 2] - (blk[:, i, 0] if i == 0 else blk[:, i - 1, nt + 2])).mean()) for i in range(qpw)]
steady = float(np.median(d_tiles))
res["tile_cycles_median"] = steady
res["overhead_cycles_per_wg_vs_median_tiles"] = float(cyc_total.mean() - steady * nt * qpw)
# dispatch: per (xcc, se, sh, cu) the workgroups in start order; the gap between a slot's end and the next start on that CU
cu_key = (xcc & 0xf) * 4096 + ((hwid >> 8) & 0xff)                                 # CU_ID[11:8], SH_ID[12], SE_ID[15:13]
t_first = rt0.min()
res["start_spread_us"] = {"p50": float(np.percentile((rt0 - t_first) * 0.01, 50)), "max": float(((rt0 - t_first) * 0.01).max())}
res["kernel_span_us_from_stamps"] = float((rt1.max() - t_first) * 0.01)
res["distinct_cus"] = int(len(np.unique(cu_key)))
busy = 0.0
for k in np.unique(cu_key):
    m = cu_key == k
    busy += float(((rt1[m] - rt0[m]) * 0.01).sum())
res["mean_resident_workgroups_per_cu"] = busy / (len(np.unique(cu_key)) * res["kernel_span_us_from_stamps"])
res["wave_slot_ids_seen"] = sorted(int(x) for x in np.unique(hwid & 0xf))
res["tg_ids_seen"] = sorted(int(x) for x in np.unique((hwid >> 16) & 0xf))
print(json.dumps(res, indent=1))
