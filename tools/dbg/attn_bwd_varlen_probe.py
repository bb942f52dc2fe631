#!/usr/bin/env python3
"""Where does the packed causal attention backward sit against the fp64 oracle, per sequence and per gradient (dq / dk / dv)?
python tools/dbg/attn_bwd_varlen_probe.py   (GPU)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import gritlm_oracle as O
from gritlm_amd import ops
from gpu_checks import bf, f32, rnd, DEV

def run(lens, nq, nkv, causal, seed=47, scale=0.7):
    d = 128; width = (nq + 2 * nkv) * d
    T, S = sum(lens), max(lens)
    qkv = rnd((T, width), seed, scale); dout = rnd((T, nq * d), seed + 3)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); tcu = torch.from_numpy(cu).to(DEV)
    tq, tdo = bf(qkv), bf(dout)
    lse = torch.empty((T, nq), dtype=torch.float32, device=DEV)
    out = ops.attn_bidir_varlen(tq, tcu, S, nq, nkv, d, lse=lse, causal=causal)
    got = f32(ops.attn_bidir_varlen_bwd(tq, tcu, S, out, tdo, lse, nq, nkv, d, causal=causal))
    for b, L in enumerate(lens):
        x = f32(tq)[cu[b]:cu[b + 1]].reshape(1, L, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
        dq, dk, dv = O.attention_bidirectional_backward(x[:, :nq], x[:, nq:nq + nkv], x[:, nq + nkv:], np.ones((1, L), dtype=np.int64),
                                                        f32(tdo)[cu[b]:cu[b + 1]].reshape(1, L, nq * d), causal=causal)
        ref = np.concatenate([dq, dk, dv], axis=1).transpose(0, 2, 1, 3).reshape(L, width)
        g = got[cu[b]:cu[b + 1]]
        rms = float(np.sqrt(np.mean(ref ** 2)))
        parts = {"dq": slice(0, nq * d), "dk": slice(nq * d, (nq + nkv) * d), "dv": slice((nq + nkv) * d, width)}
        msg = []
        for nm, sl in parts.items():
            r, x_ = ref[:, sl], g[:, sl]
            e = np.abs(x_ - r) / (2.0 ** -7 * np.abs(r) + 6e-2 * rms + 1e-12)
            i = np.unravel_index(np.argmax(e), e.shape)
            msg.append(f"{nm}: worst {e.max():.3f} at row {i[0]} (ref {r[i]:+.4f} got {x_[i]:+.4f}) rms_part {np.sqrt(np.mean(r**2)):.4f} rel_l2 {np.linalg.norm(x_-r)/np.linalg.norm(r):.2e}")
        print(f"lens={lens} nq={nq} nkv={nkv} causal={causal} seq {b} (L={L}) rms {rms:.4f} | " + " | ".join(msg))

run((385, 33, 512, 7), 8, 2, True)
run((385, 33, 512, 7), 8, 2, False)
run((385, 33, 512, 7), 4, 2, True)
run((129, 64, 257), 8, 2, True)
