#!/usr/bin/env python3
"""Per-operand error budget of the encode path at depth 32 -- EMULATED on the host CPU, before a kernel is touched (VERDICT r04 #1a).

The fp32 restatement below is the engine's data flow (gritlm_amd/encoder.py: RMSNorm -> q|k|v + RoPE -> softmax(QK^T)V -> o_proj +
residual -> RMSNorm -> SwiGLU(gate|up) -> down + residual; final RMSNorm, mean pool, L2 normalise) at the 7B LAYER shape (H 4096,
I 14336, 32/8 heads, d 128) with 32 DISTINCT layers of N(0, 0.02) bf16-representable weights (bench.py's model family), run once per
POLICY.  A policy says, for every class of value that is an MFMA operand or a stored activation in the engine, which 16-bit format it is
rounded to ("bf16", "f16") or that it is left in fp32 ("-"):

    x     RMSNorm output               (A operand of q|k|v and gate|up)
    qkv   q|k|v after RoPE             (operands of QK^T and PV)
    p     exp(s - max), un-normalised  (B operand of PV)
    ctx   attention output             (A operand of o_proj)
    act   silu(gate) * up              (A operand of down)
    lin   Linear output before the residual add  (the reference's bf16 arithmetic rounds it; an fp32 stream does not)
    h     the residual stream
    out   final RMSNorm output handed to the pooling kernel

Every policy is compared with the all-fp32 run of the same weights: 1 - cos of the pooled, normalised embeddings (float64).
Scenarios: the two shipped policies (reference bf16 arithmetic; bf16 operands + fp32 stream: these calibrate the emulation against the
GPU measurements of profiles/r04_depth_parity.json), ONE operand class at a time in bf16 and in f16 with everything else in fp32 (the
budget), all operands in f16 with an fp32 stream (the candidate policy), and that policy with f16 subnormal weights flushed to zero
(what an MFMA that flushes denormal inputs would compute).

    python tools/precision_budget.py [--docs 2] [--seq 512] [--layers 32] [--out profiles/r05_precision_budget.json]
"""
import argparse
import json
import math
import os
import time

import torch

H, I, NQ, NKV, D, EPS, THETA = 4096, 14336, 32, 8, 128, 1e-5, 10000.0
CLASSES = ("x", "qkv", "p", "ctx", "act", "lin", "h", "out")


def rnd(t: torch.Tensor, fmt: str) -> torch.Tensor:
    if fmt == "-":
        return t
    return t.to(torch.bfloat16 if fmt == "bf16" else torch.float16).float()


def policy(default="-", **kw):
    p = {c: default for c in CLASSES}
    p.update(kw)
    return p


def scenarios():
    sc = {
        "fp32": policy(),
        "engine_bf16_residual(reference arithmetic)": policy("bf16", double_round=True),
        "engine_bf16_operands_fp32_residual": policy("bf16", lin="-", h="-", double_round=True),
        "f16_operands_fp32_residual": policy("f16", lin="-", h="-"),
        "f16_operands_fp32_residual_flush_subnormal_weights": policy("f16", lin="-", h="-", flush_w=True),
        "f16_operands_bf16_out_fp32_residual": policy("f16", lin="-", h="-", out="bf16"),
        # what a MIXED policy would cost: o_proj and down back on bf16 MFMA operands (ctx, act in bf16: 37 % of the GEMM FLOPs at the bf16
        # clock), x / q|k|v / P in fp16 -- priced here, not built
        "mixed_f16_x_qkv_p__bf16_ctx_act_out_fp32_residual": policy("f16", lin="-", h="-", out="bf16", ctx="bf16", act="bf16"),
        # an fp16 residual stream instead of the fp32 one (16-bit epilogue / norm traffic), everything else as f16_operands
        "f16_operands_f16_residual": policy("f16", lin="-", out="bf16"),
    }
    for fmt in ("bf16", "f16"):
        for c in ("x", "qkv", "p", "ctx", "act", "out"):
            sc[f"only_{c}_{fmt}"] = policy(**{c: fmt})
    return sc


def rmsnorm(h, w, pol, cls):
    v = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + EPS)
    if pol.get("double_round"):            # the reference: hidden.to(input_dtype), then weight * hidden in bf16 (:84-89)
        return rnd(w * rnd(v, pol[cls]), pol[cls])
    return rnd(w * v, pol[cls])


def rope(x, cos, sin):                      # x [S, heads, D]
    x1, x2 = x[..., :D // 2], x[..., D // 2:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)


def layer(h, W, pol, cos, sin, B, S):
    """one decoder layer on h [B*S, H] under `pol`; W = dict of fp32 tensors holding bf16-representable values"""
    fl = (lambda w: torch.where(w.abs() < 6.103515625e-05, torch.zeros_like(w), w)) if pol.get("flush_w") else (lambda w: w)
    x = rmsnorm(h, W["ln1"], pol, "x")
    qkv = x @ fl(W["wqkv"]).t()
    if pol.get("double_round"):
        qkv = rnd(qkv, pol["qkv"])          # q/k/v = bf16(Linear) first (:655-657), then the rotation, rounded once more
    qkv = qkv.view(B, S, NQ + 2 * NKV, D)
    q, k, v = qkv[:, :, :NQ], qkv[:, :, NQ:NQ + NKV], qkv[:, :, NQ + NKV:]
    q = rnd(rope(q, cos, sin), pol["qkv"]); k = rnd(rope(k, cos, sin), pol["qkv"]); v = rnd(v, pol["qkv"])
    ctx = torch.empty((B, S, NQ, D))
    scale = 1.0 / math.sqrt(D)
    for b in range(B):
        for g in range(NKV):
            qg = q[b, :, g * (NQ // NKV):(g + 1) * (NQ // NKV)].permute(1, 0, 2)          # [4, S, D]
            s = (qg @ k[b, :, g].t()) * scale                                               # [4, S, S]
            p = torch.exp(s - s.max(-1, keepdim=True).values)
            l = p.sum(-1, keepdim=True)                                                      # fp32 row sum of the UN-rounded p (as the kernel)
            o = (rnd(p, pol["p"]) @ v[b, :, g]) / l
            ctx[b, :, g * (NQ // NKV):(g + 1) * (NQ // NKV)] = o.permute(1, 0, 2)
    ctx = rnd(ctx.reshape(B * S, NQ * D), pol["ctx"])
    h = rnd(h + rnd(ctx @ fl(W["wo"]).t(), pol["lin"]), pol["h"])
    x = rmsnorm(h, W["ln2"], pol, "x")
    g_, u_ = x @ fl(W["wg"]).t(), x @ fl(W["wu"]).t()
    if pol.get("double_round"):             # bf16(gate), bf16(up), bf16(silu) -- the reference's elementwise bf16 chain (:177-178)
        g_, u_ = rnd(g_, pol["act"]), rnd(u_, pol["act"])
        act = rnd(rnd(torch.nn.functional.silu(g_), pol["act"]) * u_, pol["act"])
    else:
        act = rnd(torch.nn.functional.silu(g_) * u_, pol["act"])
    stats = {"act_absmax": float(act.abs().max()), "x_absmax": float(x.abs().max()), "h_absmax": float(h.abs().max())}
    h = rnd(h + rnd(act @ fl(W["wd"]).t(), pol["lin"]), pol["h"])
    return h, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=2)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_precision_budget.json"))
    ap.add_argument("--only", default="", help="comma-separated scenario-name substrings")
    args = ap.parse_args()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    B, S = args.docs, args.seq
    sc = scenarios()
    if args.only:
        keep = [s for s in args.only.split(",") if s]
        sc = {k: v for k, v in sc.items() if k == "fp32" or any(s in k for s in keep)}
    gen = torch.Generator().manual_seed(0)
    bf = lambda t: t.to(torch.bfloat16).float()
    lin = lambda o, i: bf(torch.randn((o, i), generator=gen) * 0.02)
    nrm = lambda: bf(1.0 + 0.1 * torch.randn((H,), generator=gen))
    emb0 = lin(B * S, H)                     # the embedding rows of B*S random tokens
    inv = 1.0 / (THETA ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    fr = torch.outer(torch.arange(S, dtype=torch.float32), inv)
    cos32, sin32 = fr.cos()[None, :, None, :], fr.sin()[None, :, None, :]
    cosb, sinb = bf(cos32), bf(sin32)        # the reference casts its tables to the model dtype (:124-125)
    hs = {k: rnd(emb0.clone(), p["h"]) for k, p in sc.items()}
    t0 = time.time()
    curve = {k: {} for k in sc}
    absmax = {"act": 0.0, "x": 0.0, "h": 0.0}
    final_w = None
    for li in range(args.layers):
        W = {"wqkv": lin((NQ + 2 * NKV) * D, H), "wo": lin(H, NQ * D), "wg": lin(I, H), "wu": lin(I, H), "wd": lin(H, I), "ln1": nrm(), "ln2": nrm()}
        for k, p in sc.items():
            tabs = (cosb, sinb) if p.get("double_round") else (cos32, sin32)
            hs[k], st = layer(hs[k], W, p, tabs[0], tabs[1], B, S)
            if k == "fp32":
                for c in absmax:
                    absmax[c] = max(absmax[c], st[c + "_absmax"])
        if (li + 1) in (1, 2, 4, 8, 16, 32, args.layers):
            if final_w is None:
                final_w = nrm()
            emb = {}
            for k, p in sc.items():
                o = rmsnorm(hs[k], final_w, p, "out").view(B, S, H)
                emb[k] = torch.nn.functional.normalize(o.double().mean(1), dim=-1)
            for k in sc:
                d = 1.0 - (emb[k] * emb["fp32"]).sum(-1)
                curve[k][str(li + 1)] = {"max": float(d.max()), "mean": float(d.mean())}
            print(f"layer {li + 1:2d}  {time.time() - t0:6.0f} s  " + "  ".join(f"{k[:28]}={curve[k][str(li + 1)]['max']:.2e}" for k in list(sc)[1:5]), flush=True)
    res = {"what": __doc__.split("\n\n")[0], "docs": B, "seq": S, "layers": args.layers, "shape": "7B layer (H 4096, I 14336, 32/8 heads, d 128), distinct N(0,0.02) bf16 weights per layer",
           "host": "torch CPU fp32, " + str(torch.get_num_threads()) + " threads", "one_minus_cos_vs_fp32": curve,
           "fp32_run_absmax": absmax, "f16_max": 65504.0, "seconds": time.time() - t0}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps({k: v.get(str(args.layers)) for k, v in curve.items()}, indent=1))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
