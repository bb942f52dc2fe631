"""GPU parity checks: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Each check returns a dict(name, ok, detail).  Used by tests/test_gpu_parity.py (pytest -m gpu) and by
tools/gpu_diag.py (one table for a whole gpurun call)."""
from __future__ import annotations

import os

import numpy as np
import torch

import gritlm_oracle as O
import synth
from gritlm_amd import ops
from gritlm_amd._lib import EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_STORE, EPI_SWIGLU
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine, swiglu_interleave

DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DUMP = os.environ.get("GRIT_DUMP_DIR")


def _yard() -> dict:
    """The yardsticks taken from the reference's OWN bf16 runs, FROZEN as numbers (tests/golden/numeric_bounds.json, written by
    make_bounds.py from the committed fixtures): the `*_bf16` arrays depend on the host that generated the fixture, the bounds built
    on them must not."""
    import json
    if not hasattr(_yard, "v"):
        _yard.v = json.load(open(os.path.join(GOLDEN, "numeric_bounds.json")))["values"]
    return _yard.v


def bf(x: np.ndarray) -> torch.Tensor:
    """bf16-representable fp32 numpy -> bf16 cuda tensor (exact)."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV).to(torch.bfloat16)


def f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def _res(name, ok, **kw):
    return dict(name=name, ok=bool(ok), detail=" ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kw.items()))


def _dump(name, **arrs):
    if DUMP:
        os.makedirs(DUMP, exist_ok=True)
        np.savez_compressed(os.path.join(DUMP, name + ".npz"), **arrs)


def rnd(shape, seed, scale=1.0):
    return O.bf16_round(np.random.default_rng(seed).standard_normal(shape, dtype=np.float32) * scale)


# ---------------------------------------------------------------------------------------------
def check_embed():
    tab = rnd((97, 64), 1)
    ids = np.random.default_rng(2).integers(0, 97, size=(3, 11))
    out = f32(ops.embed_gather(bf(tab), torch.from_numpy(ids).to(DEV).view(-1)))
    return _res("embed_gather", np.array_equal(out, tab[ids.reshape(-1)]))


def check_rmsnorm(T=37, H=4096):
    x, w = rnd((T, H), 3, 2.0), O.bf16_round(1 + 0.1 * rnd((H,), 4))
    out = f32(ops.rmsnorm(bf(x), bf(w), 1e-5))
    ref = O.rmsnorm(x, w, 1e-5, emulate_bf16=True)
    err = float(np.max(np.abs(out - ref) / (np.abs(ref) + 1e-3)))
    exact = float(np.mean(out == ref))
    return _res(f"rmsnorm[T={T},H={H}]", err < 1e-2 and exact > 0.98, max_rel=err, exact_frac=exact)


def check_rope(B=2, S=50, nq=4, nkv=2, d=128, inverse=False):
    width = (nq + 2 * nkv) * d
    qkv = rnd((B * S, width), 5)
    cos, sin = O.rope_tables(S, d, 10000.0)
    t = bf(qkv)
    ops.rope_qk_(t, torch.from_numpy(cos[:, : d // 2].copy()).to(DEV), torch.from_numpy(sin[:, : d // 2].copy()).to(DEV), S, nq, nkv, d,
                 inverse=inverse)
    out = f32(t)
    x = qkv.reshape(B, S, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
    sgn = -1.0 if inverse else 1.0
    ref = x.copy()
    ref[:, : nq + nkv] = O.apply_rope(x[:, : nq + nkv], cos, sgn * sin)
    ref = O.bf16_round(ref.transpose(0, 2, 1, 3).reshape(B * S, width))
    err = float(np.max(np.abs(out - ref)))
    v_same = np.array_equal(out[:, (nq + nkv) * d:], qkv[:, (nq + nkv) * d:])
    return _res(f"rope[inv={int(inverse)}]", err < 4e-2 and v_same and float(np.mean(out == ref)) > 0.97, max_abs=err, v_untouched=v_same,
                exact_frac=float(np.mean(out == ref)))


def check_gemm(M, N, K, epi=EPI_STORE, seed=7):
    a, w = rnd((M, K), seed), rnd((N, K), seed + 1, 0.05)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    if epi == EPI_RESIDUAL:
        r = rnd((M, N), seed + 2)
        out = f32(ops.gemm_nt(bf(a), bf(w), epilogue=epi, residual=bf(r)))
        ref = O.bf16_round(ref.astype(np.float32)).astype(np.float64) + r
    elif epi == EPI_SWIGLU:
        I = N // 2
        wg, wu = w[:I], w[I:]
        wi = swiglu_interleave(bf(wg), bf(wu))
        out = f32(ops.gemm_nt(bf(a), wi, epilogue=epi))
        g = O.bf16_round((a.astype(np.float64) @ wg.astype(np.float64).T).astype(np.float32))
        u = O.bf16_round((a.astype(np.float64) @ wu.astype(np.float64).T).astype(np.float32))
        ref = (O.bf16_round(O.silu(g.astype(np.float64)).astype(np.float32)) * u).astype(np.float64)
    else:
        out = f32(ops.gemm_nt(bf(a), bf(w)))
    scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-9
    # outputs are bf16: 1 ulp = 2^-8 relative; allow ~2 ulp of the value + a small absolute floor
    err = float(np.max(np.abs(out - ref) / (1.2e-2 * np.abs(ref) + 1e-2 * scale)))
    ok = err < 1.0
    if not ok:
        _dump(f"gemm_{M}_{N}_{K}_{epi}", a=a, w=w, out=out, ref=ref.astype(np.float32))
    return _res(f"gemm[M={M},N={N},K={K},epi={epi}]", ok, max_err_over_tol=err)


def check_gemm_residual_f32(M=520, N=512, K=192, seed=27, in_place=True):
    """GRIT_EPI_RESIDUAL_F32 (the fp32 residual stream): C = R + A W^T with fp32 R / C and NOTHING rounded -- against fp64 of the same bf16
    operands, to fp32 accumulation accuracy (1e-5 relative to the row scale), in place and out of place, ragged M / N edges."""
    a, w = rnd((M, K), seed), rnd((N, K), seed + 1, 0.05)
    r = np.random.default_rng(seed + 2).standard_normal((M, N)).astype(np.float32) * 3.0        # NOT bf16-representable on purpose
    ref = a.astype(np.float64) @ w.astype(np.float64).T + r.astype(np.float64)
    rt = torch.from_numpy(r).to(DEV)
    if in_place:
        out = rt
        ops.gemm_nt(bf(a), bf(w), out=out, epilogue=EPI_RESIDUAL_F32, residual=rt)
    else:
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm_nt(bf(a), bf(w), out=out, epilogue=EPI_RESIDUAL_F32, residual=rt)
    got = out.cpu().numpy().astype(np.float64)
    scale = float(np.sqrt(np.mean(ref ** 2)))
    err = float(np.max(np.abs(got - ref)) / scale)
    return _res(f"gemm_residual_f32[M={M},N={N},K={K},in_place={in_place}]", np.isfinite(got).all() and err < 2e-5, max_err_over_rms=err)


def check_f32_stream_ops(T=37, H=4096):
    """grit_embed_gather_f32 (exact widening) and grit_rmsnorm_fwd_f32in (fp32 rows, ONE rounding) vs numpy."""
    tab = rnd((97, H), 1)
    ids = np.random.default_rng(2).integers(0, 97, size=(T,))
    out = torch.empty((T, H), dtype=torch.float32, device=DEV)
    ops.embed_gather(bf(tab), torch.from_numpy(ids).to(DEV), out=out)
    same = np.array_equal(out.cpu().numpy(), tab[ids])
    x = np.random.default_rng(3).standard_normal((T, H)).astype(np.float32) * 2.0
    w = O.bf16_round(1 + 0.1 * rnd((H,), 4))
    y = f32(ops.rmsnorm(torch.from_numpy(x).to(DEV), bf(w), 1e-5))
    x64 = x.astype(np.float64)
    ref = w * (x64 * (1.0 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5)))
    refb = O.bf16_round(ref.astype(np.float32))
    exact = float(np.mean(y == refb))
    err = float(np.max(np.abs(y - ref) / (np.abs(ref) + 1e-3)))
    return _res(f"f32_stream_ops[T={T},H={H}]", same and err < 5e-3 and exact > 0.99, gather_exact=same, rms_max_rel=err, rms_exact_frac=exact)


def check_gemm_counter_rings_are_reclaimed(n_streams=40):
    """The persistent GEMM's tile-queue counter rings (csrc/gemm_bf16.hip: 16 rings keyed by (device, stream)): more streams than rings over
    the life of a process must keep working AND keep the persistent form -- an idle stream's ring is reclaimed -- with results
    bit-identical to the default stream's; streams that are all busy at once fall back to the per-tile form, same bits."""
    M, N, K = 8192, 4096, 512                                      # 512 tiles >= 2 per CU: the persistent form
    g = torch.Generator(device=DEV).manual_seed(77)
    a = torch.randn((M, K), generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    ref = ops.gemm_nt(a, w)
    torch.cuda.synchronize()
    same = True
    for i in range(n_streams):                                     # one after the other: every earlier stream is idle when the next one starts
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            out = ops.gemm_nt(a, w)
        st.synchronize()
        same &= bool(torch.equal(out, ref))
        del st
    streams = [torch.cuda.Stream() for _ in range(24)]             # 24 streams busy at the same time: some take the per-tile form
    outs = []
    for st in streams:
        with torch.cuda.stream(st):
            for _ in range(3):
                o = ops.gemm_nt(a, w)
            outs.append(o)
    torch.cuda.synchronize()
    same_busy = all(bool(torch.equal(o, ref)) for o in outs)
    return _res(f"gemm counter rings reclaimed [{n_streams} sequential + 24 concurrent streams]", same and same_busy, sequential_identical=same,
                concurrent_identical=same_busy)


def check_gemm_pair(M1=300, N1=272, M2=520, N2=720, K=192, accumulate=True, seed=17):
    """grit_gemm_bf16_nt_pair: two problems in one launch must give the bits of two grit_gemm_bf16_nt launches (strided operands too)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    a1, w1, a2, w2 = mk(M1, K + 64)[:, :K], mk(N1, K), mk(M2, K), mk(N2, K + 128)[:, :K]
    o1, o2 = mk(M1, N1), mk(M2, N2 + 8)[:, :N2]
    r1, r2 = o1.clone(), o2.clone()
    if accumulate:
        ops.gemm_nt(a1, w1, out=r1, epilogue=EPI_RESIDUAL, residual=r1)
        ops.gemm_nt(a2, w2, out=r2, epilogue=EPI_RESIDUAL, residual=r2)
    else:
        ops.gemm_nt(a1, w1, out=r1)
        ops.gemm_nt(a2, w2, out=r2)
    ops.gemm_nt_pair(a1, w1, o1, a2, w2, o2, accumulate=accumulate)
    same = bool(torch.equal(o1, r1)) and bool(torch.equal(o2, r2))
    return _res(f"gemm_pair[{M1}x{N1}+{M2}x{N2},K={K},acc={accumulate}]", same,
                diff1=float((o1.float() - r1.float()).abs().max()), diff2=float((o2.float() - r2.float()).abs().max()))


def check_gemm_rope(M=300, nq=4, nkv=2, K=256, S=77, packed=False):
    """Fused QKV GEMM + RoPE epilogue == GEMM followed by the stand-alone RoPE kernel, bit for bit (v heads untouched)."""
    from gritlm_amd.encoder import rope_tables
    d = 128
    N = (nq + 2 * nkv) * d
    a, w = bf(rnd((M, K), 71)), bf(rnd((N, K), 72, 0.05))
    cos, sin = rope_tables(max(S, 128), d, 10000.0, True, DEV)
    if packed:
        pos = torch.from_numpy((np.arange(M) * 7 % max(S, 128)).astype(np.int32)).to(DEV)
        ref = ops.rope_qk_pos_(ops.gemm_nt(a, w), cos, sin, pos, nq, nkv, d)
        got = ops.gemm_nt_rope(a, w, cos, sin, (nq + nkv) * d, positions=pos)
    else:
        ref = ops.rope_qk_(ops.gemm_nt(a, w), cos[:S].contiguous(), sin[:S].contiguous(), S, nq, nkv, d)
        got = ops.gemm_nt_rope(a, w, cos, sin, (nq + nkv) * d, S=S)
    same = bool(torch.equal(got, ref))
    plain = ops.gemm_nt(a, w)
    ok = same and bool(torch.equal(got[:, (nq + nkv) * d:], plain[:, (nq + nkv) * d:])) and not bool(torch.equal(got[:, :d], plain[:, :d]))
    return _res(f"gemm+rope epilogue [M={M},nq={nq},nkv={nkv},K={K},packed={int(packed)}]", ok,
                max_abs_diff=float((got.float() - ref.float()).abs().max()))


def check_mask_pack():
    rng = np.random.default_rng(9)
    m = (rng.random((5, 200)) < 0.6).astype(np.int64)
    m[1] = 1; m[2] = 0; m[3, 130:] = 0
    bits = ops.mask_pack(torch.from_numpy(m).to(DEV)).cpu().numpy().view(np.uint64)
    ok = True
    for b in range(5):
        for s in range(256):
            want = int(m[b, s]) if s < 200 else 0
            ok &= ((int(bits[b, s // 64]) >> (s % 64)) & 1) == (1 if want else 0)
    return _res("mask_pack", ok)


def check_attention(B=2, S=200, nq=4, nkv=2, mask_kind="ragged", seed=11, causal=False, window=0):
    d = 128
    width = (nq + 2 * nkv) * d
    qkv = rnd((B * S, width), seed)
    rng = np.random.default_rng(seed + 1)
    mask = np.ones((B, S), dtype=np.int64)
    if mask_kind == "ragged":
        for b in range(1, B):
            mask[b, rng.integers(S // 3, S):] = 0
    elif mask_kind == "holes":
        mask = (rng.random((B, S)) < 0.7).astype(np.int64); mask[:, 0] = 1
    elif mask_kind == "ragged_long":
        mask[0, S - 700:] = 0
    elif mask_kind == "left":          # instruction-style: leading zeros, first 64-key tile fully masked
        mask[:, :70] = 0
    elif mask_kind == "short_rows":    # documents whose keys end inside the first 64-key tile (one K/V tile per query block) next to a full one
        mask[0, 1:] = 0
        if B > 1:
            mask[1, 37:] = 0
    x = qkv.reshape(B, S, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
    q, k, v = x[:, :nq], x[:, nq:nq + nkv], x[:, nq + nkv:]
    ref = O.attention_bidirectional(q, k, v, mask, causal=causal, window=window)
    lse_t = torch.empty((B, nq, S), dtype=torch.float32, device=DEV)
    bits = ops.mask_pack(torch.from_numpy(mask).to(DEV))
    out = f32(ops.attn_bidir(bf(qkv), bits, B, S, nq, nkv, d, lse=lse_t, causal=causal, window=window)).reshape(B, S, nq * d)
    # reference lse
    kk = np.repeat(k, nq // nkv, axis=1)
    sc = np.einsum("bhqd,bhkd->bhqk", q.astype(np.float64), kk.astype(np.float64)) / np.sqrt(d)
    allowed = np.broadcast_to(mask.astype(bool)[:, None, None, :], sc.shape)
    if causal:
        allowed = allowed & O.causal_window_mask(S, window)[None, None]
    sc = np.where(allowed, sc, -np.inf)
    sees = allowed.any(-1)                      # a padding query behind a sliding window sees no key at all: output 0, lse -inf
    mx = np.where(sees, sc.max(-1), 0.0)[..., None]
    with np.errstate(divide="ignore"):
        lse_ref = (mx[..., 0] + np.log(np.exp(sc - mx).sum(-1)))
    err = float(np.max(np.abs(out - ref)))
    lerr = float(np.max(np.abs(f32(lse_t) - lse_ref)[sees]))
    ok = err < 2e-2 and lerr < 2e-3 and not np.isnan(out).any() and bool(np.all(np.isneginf(f32(lse_t)[~sees])))
    if not ok:
        _dump(f"attn_{mask_kind}_{S}", qkv=qkv, mask=mask, out=out, ref=ref, lse=f32(lse_t), lse_ref=lse_ref.astype(np.float32))
    return _res(f"attention[B={B},S={S},nq={nq},nkv={nkv},{mask_kind},causal={int(causal)},window={window}]", ok, max_abs=err, lse_abs=lerr)


# ---------------------------------------------------------------------------------------------  the forward's multi-block ("seam") path
_SEAM_CASES = [      # (function, kwargs): shapes with several 128-row query blocks, S not a multiple of 128, rows whose keys end inside the
    #                  first 64-key tile (ntiles == 1 for their workgroups), holes, leading masked tiles, causal, packed rows
    ("check_attention", dict(B=2, S=200, nq=4, nkv=2, mask_kind="ragged")),
    ("check_attention", dict(B=3, S=513, nq=4, nkv=2, mask_kind="ragged", seed=21)),
    ("check_attention", dict(B=2, S=330, nq=2, nkv=1, mask_kind="holes", seed=22)),
    ("check_attention", dict(B=2, S=448, nq=4, nkv=1, mask_kind="left", seed=23)),
    ("check_attention", dict(B=2, S=512, nq=8, nkv=2, mask_kind="short_rows", seed=24)),
    ("check_attention", dict(B=1, S=1024, nq=2, nkv=1, mask_kind="none", seed=25)),
    ("check_attention", dict(B=2, S=513, nq=4, nkv=2, mask_kind="ragged", seed=26, causal=True)),
    ("check_attention", dict(B=2, S=512, nq=8, nkv=2, mask_kind="short_rows", seed=27, causal=True)),
    ("check_attention_bwd_varlen", dict(lens=(200, 71, 128, 1, 300, 513, 64, 40), nq=4, nkv=2)),
    ("check_attention_bwd_varlen", dict(lens=(385, 33, 512, 7), nq=4, nkv=2, causal=True)),
]


def check_attention_seam(qpw=4):
    """The forward keeps one workgroup on ``qpw`` consecutive query blocks when B * heads * blocks is large (the bench and production
    shapes: 64 x 512 x 32 heads -> qpw 4) and streams K / V through the block seams; the small parity shapes all run qpw = 1.  The
    choice is a function static read from GRIT_ATTN_QPW at first use, so this check re-runs the oracle comparisons in a child process
    with the knob forced (attention.hip:443)."""
    import json
    import subprocess
    import sys
    prog = ("import json, sys; sys.path[:0] = %r; import gpu_checks as G; "
            "print('SEAM ' + json.dumps([getattr(G, f)(**kw) for f, kw in G._SEAM_CASES]))") % [p for p in sys.path if p]
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=900, env=dict(os.environ, GRIT_ATTN_QPW=str(qpw)))
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("SEAM ")]
    if r.returncode != 0 or not line:
        return _res(f"attention seam path [qpw={qpw}]", False, rc=r.returncode, stderr=r.stderr[-400:].replace("\n", " | "))
    res = json.loads(line[0][5:])
    bad = [x["name"] + ": " + x["detail"] for x in res if not x["ok"]]
    return _res(f"attention seam path [qpw={qpw}]", not bad, cases=len(res), failed="; ".join(bad) if bad else "none")


def check_attention_production_shape(B=64, S=512, nq=32, nkv=8, causal=False, seed=31):
    """The shape class the bench launches (B * heads * query blocks >= 2048 -> the multi-block path WITHOUT any knob), ragged keys incl.
    rows that end inside the first tile, vs fp32 math attention (torch, explicit additive mask) on the same bf16 inputs; packed rows
    (varlen launch) must reproduce the padded result bit for bit."""
    d = 128
    width = (nq + 2 * nkv) * d
    g = torch.Generator(device=DEV).manual_seed(seed)
    qkv = (torch.randn((B * S, width), generator=g, device=DEV) * 0.8).to(torch.bfloat16)
    lens = torch.randint(1, S + 1, (B,), generator=g, device=DEV)
    lens[0], lens[1], lens[2], lens[3] = S, 1, 37, 129
    mask = (torch.arange(S, device=DEV).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
    bits = ops.mask_pack(mask)
    lse = torch.empty((B, nq, S), dtype=torch.float32, device=DEV)
    out = ops.attn_bidir(qkv, bits, B, S, nq, nkv, d, lse=lse, causal=causal).view(B, S, nq, d)
    x = qkv.view(B, S, nq + 2 * nkv, d).float()
    q, k, v = x[:, :, :nq].permute(0, 2, 1, 3), x[:, :, nq:nq + nkv].permute(0, 2, 1, 3), x[:, :, nq + nkv:].permute(0, 2, 1, 3)
    k, v = k.repeat_interleave(nq // nkv, dim=1), v.repeat_interleave(nq // nkv, dim=1)
    worst = worst_lse = 0.0
    for b0 in range(0, B, 8):                      # fp32 scores of 8 rows at a time: 8 x 32 x 512 x 512 x 4 B = 268 MB
        sl = slice(b0, b0 + 8)
        sc = torch.matmul(q[sl], k[sl].transpose(-1, -2)) * d ** -0.5
        sc = sc.masked_fill(mask[sl, None, None, :] == 0, float("-inf"))
        if causal:
            sc = sc.masked_fill(~torch.tril(torch.ones((S, S), dtype=torch.bool, device=DEV)), float("-inf"))
        ref = torch.matmul(torch.softmax(sc, dim=-1), v[sl]).permute(0, 2, 1, 3)              # [b,S,nq,d]
        valid = (mask[sl] != 0)
        worst = max(worst, float(((out[sl].float() - ref).abs() * valid[:, :, None, None]).max().item()))
        dl = (lse[sl] - torch.logsumexp(sc, dim=-1)).abs() * valid[:, None, :]
        worst_lse = max(worst_lse, float(torch.nan_to_num(dl, nan=0.0).max().item()))
    cu = torch.zeros((B + 1,), dtype=torch.int32, device=DEV); cu[1:] = torch.cumsum(lens, 0)
    keep = mask.bool().view(-1)
    pout = ops.attn_bidir_varlen(qkv[keep].contiguous(), cu, S, nq, nkv, d, causal=causal)
    same = bool(torch.equal(pout, out.reshape(B * S, nq * d)[keep]))
    ok = worst < 2e-2 and worst_lse < 2e-3 and same and bool(torch.isfinite(out[mask.bool()]).all())
    return _res(f"attention production shape [B={B},S={S},nq={nq},nkv={nkv},causal={int(causal)}]", ok, max_abs=worst, lse_abs=worst_lse,
                packed_identical=same)


def check_pool(method, normalize=True, B=5, S=70, H=256):
    hid = rnd((B, S, H), 13)
    rng = np.random.default_rng(14)
    mask = np.ones((B, S), dtype=np.int64)
    mask[1, S // 2:] = 0
    if B > 2:
        mask[2] = (rng.random(S) < 0.5); mask[2, 5] = 1
    if B > 3:
        mask[3, 1:] = 0
    instr = np.array([0, 3, 0, 0, 17][:B], dtype=np.int32) if method in ("mean", "weightedmean") else None
    pm = O.instruction_mask(mask, instr)
    ref = O.pooling(hid, pm, method)
    if normalize:
        ref = O.l2_normalize(ref)
    inv = torch.empty((B,), dtype=torch.float32, device=DEV)
    out = f32(ops.pool_norm(bf(hid), torch.from_numpy(mask).to(DEV), method, normalize,
                            None if instr is None else torch.from_numpy(instr).to(DEV), inv_norm=inv))
    err = float(np.max(np.abs(out - ref)))
    ok = err < 2e-5 * max(1.0, float(np.abs(ref).max()))
    return _res(f"pool[{method},norm={int(normalize)}]", ok, max_abs=err)


def check_pool_bwd(method, normalize=True, B=3, S=40, H=256):
    hid = rnd((B, S, H), 15)
    mask = np.ones((B, S), dtype=np.int64); mask[1, 25:] = 0; mask[2, :7] = 0
    instr = np.array([2, 0, 9], dtype=np.int32)
    pm = O.instruction_mask(mask, instr)
    go = np.random.default_rng(16).standard_normal((B, H)).astype(np.float32)
    ref = O.pool_normalize_backward(hid, pm, method, normalize, go)
    tm, ti = torch.from_numpy(mask).to(DEV), torch.from_numpy(instr).to(DEV)
    inv = torch.empty((B,), dtype=torch.float32, device=DEV)
    y = ops.pool_norm(bf(hid), tm, method, normalize, ti, inv_norm=inv)
    dh = f32(ops.pool_norm_bwd(y, torch.from_numpy(go).to(DEV), inv, tm, method, normalize, S, ti))
    err = float(np.max(np.abs(dh - ref))) / (float(np.abs(ref).max()) + 1e-12)
    return _res(f"pool_bwd[{method},norm={int(normalize)}]", err < 1e-2, max_rel_to_peak=err)


def check_infonce(tag):
    g = np.load(os.path.join(GOLDEN, "infonce.npz"))
    q, p, tau = g[f"{tag}_q"], g[f"{tag}_p"], float(g[f"{tag}_tau"])
    loss, dq, dp = ops.infonce(torch.from_numpy(q).to(DEV), torch.from_numpy(p).to(DEV), tau)
    lerr = abs(float(loss.item()) - float(g[f"{tag}_loss"]))
    e1 = float(np.max(np.abs(f32(dq) - g[f"{tag}_dq"]))) / (float(np.abs(g[f"{tag}_dq"]).max()) + 1e-12)
    e2 = float(np.max(np.abs(f32(dp) - g[f"{tag}_dp"]))) / (float(np.abs(g[f"{tag}_dp"]).max()) + 1e-12)
    # the fp32 torch golden itself loses digits in (softmax - onehot) when the positive dominates (tag c);
    # the fp64 oracle is the tight comparison, the golden the loose one
    _, dq64, dp64, _ = O.infonce(q, p, tau)
    o1 = float(np.max(np.abs(f32(dq) - dq64))) / (float(np.abs(dq64).max()) + 1e-12)
    o2 = float(np.max(np.abs(f32(dp) - dp64))) / (float(np.abs(dp64).max()) + 1e-12)
    ok = lerr < 1e-3 and e1 < 1e-2 and e2 < 1e-2 and o1 < 1e-3 and o2 < 1e-3
    return _res(f"infonce[{tag}] vs reference golden", ok, loss_abs=lerr, dq_rel=e1, dp_rel=e2, dq_rel_fp64=o1, dp_rel_fp64=o2)


def check_infonce_local_rows():
    g = np.load(os.path.join(GOLDEN, "infonce_dist2.npz"))
    q, p, tau, world = g["q"], g["p"], float(g["tau"]), int(g["world"])
    bq, bp = q.shape[0] // world, p.shape[0] // world
    ok, worst = True, 0.0
    for r in range(world):
        loss, dq, dp = ops.infonce(torch.from_numpy(q).to(DEV), torch.from_numpy(p).to(DEV), tau, r * bq, bq, r * bp, bp)
        e = max(abs(float(loss.item()) - float(g[f"loss_rank{r}"])),
                float(np.max(np.abs(f32(dq) - g[f"dq_rank{r}"]))), float(np.max(np.abs(f32(dp) - g[f"dp_rank{r}"]))))
        worst = max(worst, e); ok &= e < 1e-3
    return _res("infonce local rows vs 2-rank gloo reference", ok, worst_abs=worst)


def check_infonce_reproducible(nq=333, group=4, H=256, tau=0.02, reps=8):
    """The loss is reduced in a fixed order (per-row terms + one fixed summation tree, no float atomics): the same inputs must give
    the same BITS every time (check_train_packed_vs_padded compares two steps' losses for equality)."""
    rng = np.random.default_rng(22)
    q = torch.from_numpy(O.l2_normalize(rng.standard_normal((nq, H), dtype=np.float32))).to(DEV)
    p = torch.from_numpy(O.l2_normalize(rng.standard_normal((nq * group, H), dtype=np.float32))).to(DEV)
    first = None
    same = True
    for _ in range(reps):
        loss, dq, dp = ops.infonce(q, p, tau)
        cur = (loss.clone(), dq.clone(), dp.clone())
        if first is None:
            first = cur
        else:
            same &= all(bool(torch.equal(a, b)) for a, b in zip(first, cur))
    return _res("infonce loss and gradients bit-reproducible", same, loss=float(first[0].item()))


def check_infonce_big(nq=256, group=8, H=512, tau=0.02):
    rng = np.random.default_rng(21)
    q = O.l2_normalize(rng.standard_normal((nq, H), dtype=np.float32))
    p = O.l2_normalize(rng.standard_normal((nq * group, H), dtype=np.float32))
    loss_ref, dq_ref, dp_ref, _ = O.infonce(q, p, tau)
    loss, dq, dp = ops.infonce(torch.from_numpy(q).to(DEV), torch.from_numpy(p).to(DEV), tau)
    lerr = abs(float(loss.item()) - loss_ref)
    e1 = float(np.max(np.abs(f32(dq) - dq_ref))) / float(np.abs(dq_ref).max())
    e2 = float(np.max(np.abs(f32(dp) - dp_ref))) / float(np.abs(dp_ref).max())
    return _res(f"infonce[Nq={nq},G={group},H={H}] vs oracle", lerr < 1e-3 and e1 < 2e-3 and e2 < 2e-3, loss_abs=lerr, dq_rel=e1, dp_rel=e2)


def check_infonce_shapes(nq, group, H, q_off=0, nq_loc=None, p_off=0, np_loc=None, tau=0.05, seed=61):
    """The three products of the step (scores, dq, dp) through every path of the round-3 f32 GEMM (csrc/gemm_f32.hip): 128x128 / 64x64
    / 32x128 tiles, k-contiguous and row-contiguous LDS images, 16-byte and guarded scalar operand loads (H or offsets not a multiple
    of 4), ragged M / N / K edges, local-row ranges -- vs the fp64 oracle."""
    rng = np.random.default_rng(seed)
    q = O.l2_normalize(rng.standard_normal((nq, H), dtype=np.float32))
    p = O.l2_normalize(rng.standard_normal((nq * group, H), dtype=np.float32))
    nq_loc = nq - q_off if nq_loc is None else nq_loc
    np_loc = nq * group - p_off if np_loc is None else np_loc
    loss_ref, dq_ref, dp_ref, _ = O.infonce(q, p, tau)
    dq_ref, dp_ref = dq_ref[q_off:q_off + nq_loc], dp_ref[p_off:p_off + np_loc]
    loss, dq, dp = ops.infonce(torch.from_numpy(q).to(DEV), torch.from_numpy(p).to(DEV), tau, q_off, nq_loc, p_off, np_loc)
    lerr = abs(float(loss.item()) - loss_ref)
    e1 = float(np.max(np.abs(f32(dq) - dq_ref))) / float(np.abs(dq_ref).max())
    e2 = float(np.max(np.abs(f32(dp) - dp_ref))) / float(np.abs(dp_ref).max())
    ok = lerr < 1e-3 and e1 < 2e-3 and e2 < 2e-3 and dq.shape == dq_ref.shape and dp.shape == dp_ref.shape
    return _res(f"infonce shapes [Nq={nq},G={group},H={H},q[{q_off}:+{nq_loc}],p[{p_off}:+{np_loc}]] vs oracle", ok, loss_abs=lerr, dq_rel=e1,
                dp_rel=e2)


def check_transpose(R=136, Cc=200):
    x = rnd((R, Cc), 23)
    ok = np.array_equal(f32(ops.transpose(bf(x))), x.T)
    wide = torch.zeros((Cc, 192), dtype=torch.bfloat16, device=DEV)          # padded-K wgrad operand
    ops.transpose(bf(x), out=wide)
    ok &= np.array_equal(f32(wide)[:, :R], x.T) and float(wide[:, R:].abs().max()) == 0.0
    y = rnd((20, 64), 24)                                                     # rows not a multiple of 8
    ok &= np.array_equal(f32(ops.transpose(bf(y))), y.T)
    for (r, c) in ((1, 8), (64, 256), (65, 264), (300, 520), (1000, 1032)):     # tile edges of the 64 x 256 workgroup tile
        z = rnd((r, c), 25 + r)
        ok &= np.array_equal(f32(ops.transpose(bf(z))), z.T)
    big = bf(rnd((200, 400), 26))
    view = big[:, 16:336]                                                      # row stride 400, 320 columns
    ok &= bool(torch.equal(ops.transpose(view), view.t()))
    # rate at a weight-gradient operand of the training step (16384 tokens x 4096): read + write
    a = torch.randn((16384, 4096), device=DEV).to(torch.bfloat16)
    o = torch.empty((4096, 16384), dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        ops.transpose(a, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.transpose(a, out=o)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    ok &= bool(torch.equal(o, a.t()))
    return _res("transpose", ok, us_16384x4096=us, TB_per_s=2 * a.numel() * 2 / (us * 1e-6) / 1e12)


def check_rmsnorm_bwd(T=50, H=256, with_res=True):
    x, dy = rnd((T, H), 31, 1.5), rnd((T, H), 32)
    w = O.bf16_round(1 + 0.1 * rnd((H,), 33))
    dres = rnd((T, H), 34) if with_res else None
    dx_ref, dw_ref = O.rmsnorm_backward(dy, x, w, 1e-5)
    if with_res:
        dx_ref = dx_ref + dres
    dw = torch.full((H,), 0.5, dtype=torch.float32, device=DEV)              # accumulate semantics
    dx = f32(ops.rmsnorm_bwd(bf(dy), bf(x), bf(w), 1e-5, dw, None if dres is None else bf(dres)))
    e1 = float(np.max(np.abs(dx - dx_ref) / (1.2e-2 * np.abs(dx_ref) + 1e-2 * np.sqrt(np.mean(dx_ref ** 2)))))
    e2 = float(np.max(np.abs(f32(dw) - 0.5 - dw_ref))) / float(np.abs(dw_ref).max())
    extra = {}
    if with_res and H == 256:     # rate at a training chunk (16384 tokens x 4096): dy, x, dres read + dx written
        Tb, Hb = 16384, 4096
        a, b_, c_ = (torch.randn((Tb, Hb), device=DEV).to(torch.bfloat16) for _ in range(3))
        wb = torch.ones((Hb,), device=DEV, dtype=torch.bfloat16)
        dwb = torch.zeros((Hb,), dtype=torch.float32, device=DEV)
        for _ in range(3):
            ops.rmsnorm_bwd(a, b_, wb, 1e-5, dwb, c_)
        e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.rmsnorm_bwd(a, b_, wb, 1e-5, dwb, c_)
        e1_.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1_) * 100.0
        extra = dict(us_16384x4096=us, TB_per_s=4 * Tb * Hb * 2 / (us * 1e-6) / 1e12)
    return _res(f"rmsnorm_bwd[T={T},H={H},res={int(with_res)}]", e1 < 1.0 and e2 < 1e-3, dx_err_over_tol=e1, dw_rel=e2, **extra)


def check_swiglu(T=37, I=512):
    gu = rnd((T, 2 * I), 35, 1.5)
    dact = rnd((T, I), 36)
    g, u = gu[:, :I], gu[:, I:]
    act_ref = O.bf16_round(O.silu(g.astype(np.float64)).astype(np.float32)) * u
    dg, du = O.swiglu_backward(g, u, dact)
    act = f32(ops.swiglu(bf(gu)))
    dgu = f32(ops.swiglu_bwd(bf(gu), bf(dact)))
    tol = lambda a, r: float(np.max(np.abs(a - r) / (1.2e-2 * np.abs(r) + 1e-2 * np.sqrt(np.mean(r ** 2)))))
    e = max(tol(act, act_ref), tol(dgu[:, :I], dg), tol(dgu[:, I:], du))
    return _res("swiglu fwd/bwd (concat layout)", e < 1.0, err_over_tol=e)


def check_attention_bwd(B=2, S=200, nq=4, nkv=2, mask_kind="ragged", seed=41, causal=False, window=0):
    d = 128
    width = (nq + 2 * nkv) * d
    qkv = rnd((B * S, width), seed, 0.7)
    dout = rnd((B * S, nq * d), seed + 3)
    rng = np.random.default_rng(seed + 1)
    mask = np.ones((B, S), dtype=np.int64)
    if mask_kind == "ragged":
        for b in range(1, B):
            mask[b, rng.integers(S // 3, S):] = 0
    elif mask_kind == "holes":
        mask = (rng.random((B, S)) < 0.7).astype(np.int64); mask[:, 0] = 1
    x = qkv.reshape(B, S, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
    q, k, v = x[:, :nq], x[:, nq:nq + nkv], x[:, nq + nkv:]
    dq, dk, dv = O.attention_bidirectional_backward(q, k, v, mask, dout.reshape(B, S, nq * d), causal=causal, window=window)
    ref = np.concatenate([dq, dk, dv], axis=1).transpose(0, 2, 1, 3).reshape(B * S, width)
    tq, bits = bf(qkv), ops.mask_pack(torch.from_numpy(mask).to(DEV))
    lse = torch.empty((B, nq, S), dtype=torch.float32, device=DEV)
    out = ops.attn_bidir(tq, bits, B, S, nq, nkv, d, lse=lse, causal=causal, window=window)
    got = f32(ops.attn_bidir_bwd(tq, bits, out, bf(dout), lse, B, S, nq, nkv, d, causal=causal, window=window))
    errs = {}
    ok = not np.isnan(got).any()
    for name, sl in (("dq", slice(0, nq * d)), ("dk", slice(nq * d, (nq + nkv) * d)), ("dv", slice((nq + nkv) * d, width))):
        r, g_ = ref[:, sl], got[:, sl]
        # outputs are bf16 (1 ulp = 2^-8 relative): elementwise bound = one ulp of the value + a floor relative to the RMS
        # (causal rows near the diagonal hold values 30x the RMS), and a per-row relative L2 bound
        e = float(np.max(np.abs(g_ - r) / (2.0 ** -7 * np.abs(r) + 6e-2 * float(np.sqrt(np.mean(r ** 2))) + 1e-12)))
        rowrel = float(np.max(np.linalg.norm(g_ - r, axis=1) / (np.linalg.norm(r, axis=1) + 1e-3 * float(np.sqrt(np.mean(r ** 2))) * np.sqrt(r.shape[1]))))
        errs[name + "_maxerr_over_tol"] = e; errs[name + "_worst_row_rel_l2"] = rowrel
        ok &= e < 1.0 and rowrel < 1e-2
    if not ok:
        _dump(f"attn_bwd_{mask_kind}_{S}", qkv=qkv, mask=mask, dout=dout, got=got, ref=ref)
    return _res(f"attention_bwd[B={B},S={S},nq={nq},nkv={nkv},{mask_kind},causal={int(causal)},window={window}]", ok, **errs)


def check_attention_bwd_varlen(lens=(200, 71, 128, 1, 300), nq=4, nkv=2, seed=47, causal=False, window=0):
    """Packed attention backward vs (1) the oracle per sequence and (2) the padded kernel on the same rows (bit-identical:
    padded query rows / masked keys only ever add exact zeros)."""
    d = 128
    width = (nq + 2 * nkv) * d
    B, S, T = len(lens), max(lens), sum(lens)
    qkv = rnd((T, width), seed, 0.7)
    dout = rnd((T, nq * d), seed + 3)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tcu = torch.from_numpy(cu).to(DEV)
    tq, tdo = bf(qkv), bf(dout)
    lse = torch.empty((T, nq), dtype=torch.float32, device=DEV)
    out = ops.attn_bidir_varlen(tq, tcu, S, nq, nkv, d, lse=lse, causal=causal, window=window)
    got = f32(ops.attn_bidir_varlen_bwd(tq, tcu, S, out, tdo, lse, nq, nkv, d, causal=causal, window=window))
    # padded twin
    mask = np.zeros((B, S), dtype=np.int64)
    pq = np.zeros((B, S, width), dtype=np.float32); pdo = np.zeros((B, S, nq * d), dtype=np.float32)
    for b, L in enumerate(lens):
        mask[b, :L] = 1
        pq[b, :L] = f32(tq)[cu[b]:cu[b + 1]]; pdo[b, :L] = f32(tdo)[cu[b]:cu[b + 1]]
    tpq, bits = bf(pq.reshape(B * S, width)), ops.mask_pack(torch.from_numpy(mask).to(DEV))
    plse = torch.empty((B, nq, S), dtype=torch.float32, device=DEV)
    pout = ops.attn_bidir(tpq, bits, B, S, nq, nkv, d, lse=plse, causal=causal, window=window)
    pgot = f32(ops.attn_bidir_bwd(tpq, bits, pout, bf(pdo.reshape(B * S, nq * d)), plse, B, S, nq, nkv, d, causal=causal,
                                  window=window)).reshape(B, S, width)
    ok = not np.isnan(got).any()
    same = all(np.array_equal(got[cu[b]:cu[b + 1]], pgot[b, :L]) for b, L in enumerate(lens))
    same_fwd = all(np.array_equal(f32(out)[cu[b]:cu[b + 1]], f32(pout).reshape(B, S, -1)[b, :L]) for b, L in enumerate(lens))
    worst = 0.0
    for b, L in enumerate(lens):
        x = f32(tq)[cu[b]:cu[b + 1]].reshape(1, L, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
        dq, dk, dv = O.attention_bidirectional_backward(x[:, :nq], x[:, nq:nq + nkv], x[:, nq + nkv:], np.ones((1, L), dtype=np.int64),
                                                        f32(tdo)[cu[b]:cu[b + 1]].reshape(1, L, nq * d), causal=causal, window=window)
        ref = np.concatenate([dq, dk, dv], axis=1).transpose(0, 2, 1, 3).reshape(L, width)
        rms = float(np.sqrt(np.mean(ref ** 2)))
        worst = max(worst, float(np.max(np.abs(got[cu[b]:cu[b + 1]] - ref) / (2.0 ** -7 * np.abs(ref) + 6e-2 * rms + 1e-12))))
    ok &= same and same_fwd and worst < 1.0
    return _res(f"attention_bwd varlen [lens={list(lens)},causal={int(causal)},window={window}]", ok, identical_to_padded=bool(same), fwd_identical=bool(same_fwd),
                maxerr_over_tol_vs_oracle=worst)


def check_pool_bwd_varlen(method="mean", normalize=True, lens=(40, 25, 33, 1), H=256):
    """Packed pool+normalise forward/backward == padded kernels on the same documents (bit for bit)."""
    B, S, T = len(lens), max(lens), sum(lens)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    hid = rnd((T, H), 18)
    go = np.random.default_rng(19).standard_normal((B, H)).astype(np.float32)
    instr = np.array([2, 0, 9, 0][:B], dtype=np.int32) if "mean" in method else None
    ti = None if instr is None else torch.from_numpy(instr).to(DEV)
    tcu, th, tgo = torch.from_numpy(cu).to(DEV), bf(hid), torch.from_numpy(go).to(DEV)
    inv = torch.empty((B,), dtype=torch.float32, device=DEV)
    y = ops.pool_norm_varlen(th, tcu, method, normalize, ti, inv_norm=inv)
    dh = f32(ops.pool_norm_varlen_bwd(y, tgo, inv, tcu, T, method, normalize, ti))
    mask = np.zeros((B, S), dtype=np.int64); ph = np.zeros((B, S, H), dtype=np.float32)
    for b, L in enumerate(lens):
        mask[b, :L] = 1; ph[b, :L] = f32(th)[cu[b]:cu[b + 1]]
    tm = torch.from_numpy(mask).to(DEV)
    pinv = torch.empty((B,), dtype=torch.float32, device=DEV)
    py = ops.pool_norm(bf(ph), tm, method, normalize, ti, inv_norm=pinv)
    pdh = f32(ops.pool_norm_bwd(py, tgo, pinv, tm, method, normalize, S, ti))
    ok = np.array_equal(f32(y), f32(py)) and np.array_equal(f32(inv), f32(pinv))
    ok &= all(np.array_equal(dh[cu[b]:cu[b + 1]], pdh[b, :L]) for b, L in enumerate(lens))
    pm = O.instruction_mask(mask, instr) if instr is not None else mask
    ref = O.pool_normalize_backward(ph, pm, method, normalize, go)
    err = max(float(np.max(np.abs(dh[cu[b]:cu[b + 1]] - ref[b, :L]))) for b, L in enumerate(lens)) / (float(np.abs(ref).max()) + 1e-12)
    ok &= err < 1e-2
    return _res(f"pool_bwd varlen[{method},norm={int(normalize)}]", ok, max_rel_to_peak_vs_oracle=err)


def check_embed_scatter():
    """Embedding weight gradient: bf16 table += fp32 sums of the token rows, summed in token order per id (deterministic): equal to
    the sequential reference BIT FOR BIT, repeatable, untouched rows untouched; plus the fp32 -> bf16 fold used for the norm weights."""
    rng = np.random.default_rng(45)
    V, H, T = 50, 264, 700                                              # ~14 tokens per id: long runs, 33 column groups (not 256 | HC)
    ids = rng.integers(0, V - 3, size=T).astype(np.int64)               # ids V-3 .. V-1 never occur
    dh = rnd((T, H), 43)
    base = rnd((V, H), 44)
    want = base.copy()
    sums = np.zeros((V, H), dtype=np.float32)
    for t in range(T):                                                  # fp32 running sums in token order, like the kernel
        sums[ids[t]] = sums[ids[t]] + dh[t]
    touched = np.unique(ids)
    want[touched] = O.bf16_round(base[touched] + sums[touched])
    tid = torch.from_numpy(ids).to(DEV)
    outs = []
    for _ in range(3):
        tab = bf(base)
        ops.embed_scatter_add(bf(dh), tid, tab)
        outs.append(f32(tab))
    ok = all(np.array_equal(o, want) for o in outs)
    acc = bf(rnd((9, 64), 44)); x = torch.from_numpy(rnd((9, 64), 46) * 3).to(DEV)
    want2 = O.bf16_round(f32(acc) + f32(x))
    ops.accum_bf16_from_f32(acc, x)
    ok &= np.array_equal(f32(acc), want2)
    return _res("embed_scatter_add (sorted, deterministic) + accum_bf16", ok, max_abs=float(np.max(np.abs(outs[0] - want))))


def build_engine(cfg_name, seed=0):
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, seed)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    eng = MistralEncoderEngine.from_state_dict(EncoderConfig.from_dict(cfg), sd, DEV)
    return eng, cfg, w


def check_encoder_golden(cfg_name):
    """End-to-end encoder vs the fixture produced by the REFERENCE modeling file (fp32 and bf16 runs)."""
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    h = f32(eng.forward(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)))
    valid = mask.astype(bool)
    ref32, refb = g["last_hidden_state"], g["last_hidden_state_bf16"]
    rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
    r_ours, r_refb = rel(h, ref32), _yard()[f"encoder_{cfg_name}/rel_refbf16_vs_fp32"]
    out = dict(rel_ours_vs_fp32=r_ours, rel_refbf16_vs_fp32=r_refb)
    # hidden states: no further from the fp32 reference than the reference's own bf16 run is (frozen number: _yard)
    ok = r_ours < 1.25 * r_refb + 1e-3 and not np.isnan(h).any()
    tm = torch.from_numpy(mask).to(DEV)
    hb = torch.from_numpy(h).to(DEV).to(torch.bfloat16)
    for method in ("mean", "weightedmean"):
        e = f32(ops.pool_norm(hb, tm, method, True))
        ref, refb16 = g[f"emb_{method}"], g[f"emb_{method}_bf16"]
        one_minus_cos = float(np.max(1 - np.sum(e * ref, axis=1)))            # stated tolerance: < 1e-4
        one_minus_cos_b = float(np.max(1 - np.sum(e * refb16, axis=1)))       # vs the reference run in bf16
        cs_delta = float(np.max(np.abs(e @ e.T - ref @ ref.T)))               # pairwise q.d^T cosines
        cs_delta_ref = _yard()[f"encoder_{cfg_name}/pair_delta_of_bf16ref_{method}"]   # the bf16 reference's own noise (frozen)
        out[f"{method}_1-cos"] = one_minus_cos; out[f"{method}_1-cos_vs_bf16ref"] = one_minus_cos_b
        out[f"{method}_pair_delta"] = cs_delta; out[f"{method}_pair_delta_of_bf16ref"] = cs_delta_ref
        ok &= one_minus_cos < 1e-4 and one_minus_cos_b < 1e-4 and cs_delta < 1.5 * cs_delta_ref + 1e-4
    if not ok:
        _dump(f"encoder_{cfg_name}", h=h, ref=ref32)
    return _res(f"encoder[{cfg_name}] vs reference golden", ok, **out)


def check_encoder_fp32_residual(cfg_name):
    """The engine with the fp32 residual stream (residual_fp32: GRIT_EPI_RESIDUAL_F32, grit_rmsnorm_fwd_f32in, grit_embed_gather_f32)
    against the reference's FP32 run (fixture): hidden states must be CLOSER to fp32 than with the bf16 stream and than the reference's
    own bf16 run, embeddings within the north-star's 1e-4; padded == packed bit for bit."""
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    valid = mask.astype(bool)
    if "last_hidden_state" in g.files:
        ref32, refb = g["last_hidden_state"], g["last_hidden_state_bf16"]
        rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
        pick = lambda h: h
    else:                                                  # 7b-l1: probe rows only
        ref32, refb = g["probe_hidden"], g["probe_hidden_bf16"]
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        pick = lambda h: h.reshape(-1, h.shape[-1])[g["probe_rows"]]
    h_b = pick(f32(eng.forward(tid, tm)))
    eng.residual_fp32 = True
    h_f = pick(f32(eng.forward(tid, tm)))
    r_b, r_f, r_ref = rel(h_b, ref32), rel(h_f, ref32), rel(refb, ref32)
    out = dict(rel_bf16_stream=r_b, rel_fp32_stream=r_f, rel_refbf16=r_ref)
    ok = r_f < 1.02 * r_b and r_f < r_ref and not np.isnan(h_f).any()      # (the final RMSNorm's bf16 output rounding is in both)
    for method in ("mean", "weightedmean"):
        e_pad = eng.encode_pooled(tid, tm, method, True, packed=False)
        e_pack = eng.encode_pooled(tid, tm, method, True, packed=True)
        d = float(np.max(1 - np.sum(f32(e_pad) * g[f"emb_{method}"], axis=1)))
        out[f"{method}_1-cos"] = d
        ok &= d < 1e-4 and bool(torch.equal(e_pad, e_pack))
    return _res(f"encoder[{cfg_name}] fp32 residual stream vs reference fp32", ok, **out)


def check_encoder_7b_layer():
    """One layer at the TRUE GritLM-7B layer shape vs the fixture produced by the reference's MistralModel(is_causal=False)
    (tests/golden/encoder_7b-l1.npz: fp32 and bf16 runs).  Every GEMM runs at the bench's N and K (6144x4096, 4096x4096, 28672x4096,
    4096x14336); padded and packed (un-padded) paths.  Tolerance: no further from the fp32 reference than 1.1x the reference's OWN bf16 run."""
    g = np.load(os.path.join(GOLDEN, "encoder_7b-l1.npz"))
    eng, cfg, w = build_engine("7b-l1", int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    h = f32(eng.forward(tid, tm))
    probe = g["probe_rows"]
    hp = h.reshape(-1, h.shape[-1])[probe]
    ref32, refb = g["probe_hidden"], g["probe_hidden_bf16"]
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    r_ours, r_refb = rel(hp, ref32), _yard()["encoder_7b-l1/rel_refbf16_vs_fp32"]
    out = dict(rel_ours_vs_fp32=r_ours, rel_refbf16_vs_fp32=r_refb, max_abs_ours=float(np.abs(hp - ref32).max()),
               max_abs_refbf16=float(np.abs(refb - ref32).max()))
    ok = r_ours < 1.1 * r_refb and not np.isnan(h).any()
    for packed in (False, True):
        for method in ("mean", "weightedmean"):
            e = f32(eng.encode_pooled(tid, tm, method, True, packed=packed))
            one_minus_cos = float(np.max(1 - np.sum(e * g[f"emb_{method}"], axis=1)))              # stated tolerance: < 1e-4
            one_minus_cos_b = float(np.max(1 - np.sum(e * g[f"emb_{method}_bf16"], axis=1)))
            ref_own = float(np.max(1 - np.sum(g[f"emb_{method}_bf16"] * g[f"emb_{method}"], axis=1)))
            out[f"{method}{'_packed' if packed else ''}_1-cos"] = one_minus_cos
            out[f"{method}_1-cos_of_bf16ref"] = ref_own
            ok &= one_minus_cos < 1e-4 and one_minus_cos_b < 1e-4
    return _res("encoder[7b-l1] vs reference golden (7B layer shape)", ok, **out)


def check_gemm_fullshape(M, N, K, epi=EPI_STORE, seed=91, samples=4096):
    """Full-size GEMM launches (the bench's N, K and up to M = 131072: tiles_m = 512, XCD remap, K = 4096 / 14336 accumulation)
    spot-checked on `samples` random outputs against fp64 dot products of the SAME bf16 operands."""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    a = (torch.randn((M, K), device=DEV, generator=gen)).to(torch.bfloat16)
    w = (torch.randn((N, K), device=DEV, generator=gen) * 0.03).to(torch.bfloat16)
    res = torch.randn((M, N), device=DEV, generator=gen).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
    rng = np.random.default_rng(seed)
    rows = torch.from_numpy(np.concatenate([rng.integers(0, M, samples - 64), np.arange(M - 32, M), np.arange(32)])).to(DEV)
    I = N // 2
    ncols = I if epi == EPI_SWIGLU else N
    cols = torch.from_numpy(np.concatenate([rng.integers(0, ncols, samples - 64), np.arange(ncols - 32, ncols), np.arange(32)])).to(DEV)
    a64 = a[rows].double()
    if epi == EPI_SWIGLU:
        wi = swiglu_interleave(w[:I].contiguous(), w[I:].contiguous())
        out = ops.gemm_nt(a, wi, epilogue=epi)
        gt = (a64 * w[:I][cols].double()).sum(1).float().to(torch.bfloat16).double()
        ut = (a64 * w[I:][cols].double()).sum(1).float().to(torch.bfloat16).double()
        ref = (torch.nn.functional.silu(gt).float().to(torch.bfloat16).double() * ut)
    else:
        out = ops.gemm_nt(a, w, epilogue=epi, residual=res)
        ref = (a64 * w[cols].double()).sum(1)
        if epi == EPI_RESIDUAL:
            ref = ref.float().to(torch.bfloat16).double() + res[rows, cols].double()
    got = out[rows, cols].double()
    scale = float(ref.pow(2).mean().sqrt()) + 1e-9
    err = float(((got - ref).abs() / (1.2e-2 * ref.abs() + 1e-2 * scale)).max())
    return _res(f"gemm_fullshape[M={M},N={N},K={K},epi={epi}]", err < 1.0 and bool(torch.isfinite(out).all()), max_err_over_tol=err, samples=samples)


# ---------------------------------------------------------------------------------------------- sparse MoE (Mixtral)
# Full-depth bounds (1 - cos of the pooled embedding against the reference's FP32 run, 32 layers, 7B layer shape).  Numeric constants,
# not ratios to a yardstick (VERDICT r03 #1c).  What they are anchored on (profiles/r04_depth_parity.json, DESIGN section 2 "depth"):
# the reference's OWN bf16 run on this fixture is 4.6e-4 .. 6.0e-4 away from its fp32 run (stored in the fixture, generated by the
# reference); the engine with the reference's bf16 rounding points measures 4.3e-4 .. 5.4e-4, with the fp32 residual stream 2.9e-4 ..
# 3.4e-4.  The kernels are bit-reproducible, so the margin covers nothing but a re-generated fixture.  The north-star's 1e-4 is not
# reachable at depth 32 with bf16 MFMA operands: the error grows linearly with depth (1e-5 per layer, independent per-layer operand
# roundings) in BOTH modes and in the reference's own bf16 run.
# (the constants live in bench.py, which reports against the same numbers: FULL_DEPTH_BOUND_BF16_RESIDUAL = 7.0e-4, _FP32_RESIDUAL = 4.5e-4)


def check_full_depth_parity(residual_fp32=False, precision=None):
    """All 32 layers at the 7B layer shape (scripts/modeling_mistral_gritlm.py:936-1096) against embeddings the REFERENCE ITSELF produced
    (tests/golden/encoder_7b-depth32.npz: its fp32 run and its own bf16 run; `ragged` = 2 x 512 with one padded row -> the explicit-mask
    path, `full` = 1 x 512 all valid -> the mask-is-None path of :1017-1020).  The engine is held to a NUMERIC bound on 1 - cos against
    the reference's fp32 embeddings; the padded and the packed engine paths must agree bit for bit; the fp32 stock module on this GPU (the
    comparator bench.py uses on its own weights) must reproduce the fixture; the stock bf16 module under both mask rules is reported."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    import torch_reference as TR
    g = np.load(os.path.join(GOLDEN, "encoder_7b-depth32.npz"))
    layers = int(g["layers"])
    precision = precision or ("fp32_residual" if residual_fp32 else "bf16")
    residual_fp32 = precision != "bf16"
    bound = {"bf16": bench.FULL_DEPTH_BOUND_BF16_RESIDUAL, "fp32_residual": bench.FULL_DEPTH_BOUND_FP32_RESIDUAL,
             "f16_operands": bench.FULL_DEPTH_BOUND_F16_OPERANDS, "f16_stream": bench.FULL_DEPTH_BOUND_F16_OPERANDS}[precision]   # fp16 policies: the north-star's own 1e-4
    cosd = lambda a, b: float(np.max(1.0 - np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))))
    cfg, w, _, _ = bench.oracle_full_depth_case(sample_docs=1, seq=64, layers=layers)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    eng = MistralEncoderEngine.from_state_dict(EncoderConfig.from_dict(cfg), sd, DEV)
    eng.set_precision(precision)
    det, ok = {}, True
    embs = {}
    for tag in ("ragged", "full"):
        tid, tm = torch.from_numpy(g[f"{tag}_input_ids"]).to(DEV), torch.from_numpy(g[f"{tag}_attention_mask"]).to(DEV)
        e_pad = eng.encode_pooled(tid, tm, "mean", True, packed=False)
        e_pack = eng.encode_pooled(tid, tm, "mean", True, packed=True)
        same = bool(torch.equal(e_pad, e_pack))
        hip = cosd(f32(e_pad), g[f"{tag}_emb"])
        det[f"{tag}_vs_ref_fp32"] = hip
        det[f"{tag}_vs_ref_bf16"] = cosd(f32(e_pad), g[f"{tag}_emb_bf16"])
        det[f"{tag}_ref_bf16_vs_ref_fp32"] = float(g[f"{tag}_ref_bf16_one_minus_cos_vs_fp32"])
        det[f"{tag}_packed_identical"] = same
        ok = ok and same and bool(np.isfinite(f32(e_pad)).all()) and hip < bound
        embs[tag] = (tid, tm)
    if precision in ("f16_operands", "f16_stream"):
        det["overflow_flag"] = ops.f16_overflow_flag(eng.device)
        det["subnormal_weight_frac"] = eng.f16_weight_stats["subnormal"] / max(eng.f16_weight_stats["total"], 1)
        ok = ok and not det["overflow_flag"]
    del eng
    torch.cuda.empty_cache()
    if not residual_fp32:                    # the comparators, once
        hf = TR.build_model(cfg, torch.float32, DEV, state_dict=sd)
        for tag, (tid, tm) in embs.items():
            d = cosd(f32(TR.encode(hf, tid, tm)), g[f"{tag}_emb"])
            det[f"{tag}_stock_fp32_gpu_vs_ref_fp32"] = d
            ok = ok and d < 1e-6
        del hf
        torch.cuda.empty_cache()
        hb = TR.build_model(cfg, torch.bfloat16, DEV, state_dict=sd)
        tid, tm = embs["full"]
        det["full_stock_bf16_gpu_mask_none"] = cosd(f32(TR.encode(hb, tid, tm, mask_rule="reference")), g["full_emb"])
        det["full_stock_bf16_gpu_mask_4d"] = cosd(f32(TR.encode(hb, tid, tm, mask_rule="explicit")), g["full_emb"])
        del hb
        torch.cuda.empty_cache()
    return _res(f"full-depth parity [L={layers},precision={precision}]", ok, bound=bound, **det)


def check_moe_router(T=777, H=512, E=8):
    x = rnd((T, H), 61)
    gw = O.bf16_round(np.random.default_rng(62).standard_normal((E, H)).astype(np.float32) * 0.5)
    experts, weights, counts, row_token, rows = ops.moe_route(bf(x), bf(gw))
    xb = f32(bf(x))
    w_ref, sel_ref, logits = O.moe_router(xb, gw, 2, emulate_bf16=True)
    sel, wt = experts.cpu().numpy(), f32(weights)
    # a logit that lands within one bf16 ulp of a rounding boundary may round differently (different fp32 summation order):
    # judge the selection only where the top-3 probabilities are separated by more than that
    pr = np.sort(np.exp(logits - logits.max(1, keepdims=True)) / np.exp(logits - logits.max(1, keepdims=True)).sum(1, keepdims=True), axis=1)[:, ::-1]
    clear = (pr[:, 1] - pr[:, 2] > 0.02 * pr[:, 1]) & (pr[:, 0] - pr[:, 1] > 0.02 * pr[:, 0])
    same = (sel == sel_ref).all(1)
    ok = bool(same[clear].all()) and clear.mean() > 0.8
    werr = float(np.max(np.abs(wt - w_ref)[same])) if same.any() else 1.0
    ok &= werr <= 2 ** -7                                   # one bf16 ulp of a weight < 1
    # index: counts, stable sort, inverse map
    flat = sel.reshape(-1)
    cnt = counts.cpu().numpy()
    ok &= np.array_equal(cnt, np.bincount(flat, minlength=E))
    order = np.argsort(flat, kind="stable")
    ok &= np.array_equal(row_token.cpu().numpy(), (order // 2).astype(np.int32))
    inv = np.empty_like(order); inv[order] = np.arange(order.size)
    ok &= np.array_equal(rows.cpu().numpy().reshape(-1), inv.astype(np.int32))
    return _res(f"moe router+index [T={T},H={H},E={E}]", bool(ok), clear_frac=float(clear.mean()), agree_all=float(same.mean()), max_weight_err=werr)


def check_gemm_grouped(counts=(300, 0, 17, 256, 513, 1, 0, 64), N=384, K=256, epi=EPI_STORE, seed=65):
    """Grouped GEMM (device-side counts, gathered A rows) vs fp64 numpy per group; empty and tiny groups included."""
    E, M = len(counts), int(sum(counts))
    Tsrc = M // 2 + 5
    a = rnd((Tsrc, K), seed)
    w = rnd((E, N, K), seed + 1, 0.05)
    rng = np.random.default_rng(seed + 2)
    a_rows = rng.integers(0, Tsrc, size=M).astype(np.int32)
    tcounts = torch.tensor(counts, dtype=torch.int32, device=DEV)
    ta, trows = bf(a), torch.from_numpy(a_rows).to(DEV)
    if epi == EPI_SWIGLU:
        I = N // 2
        wi = torch.stack([swiglu_interleave(bf(w[e, :I]), bf(w[e, I:])) for e in range(E)]).contiguous()
        out = f32(ops.gemm_nt_grouped(ta, wi, tcounts, M, epilogue=epi, a_rows=trows))
    else:
        out = f32(ops.gemm_nt_grouped(ta, bf(w), tcounts, M, a_rows=trows))
    ref = np.zeros(out.shape, dtype=np.float64)
    off = 0
    for e, c in enumerate(counts):
        xa = a[a_rows[off:off + c]].astype(np.float64)
        full = xa @ w[e].astype(np.float64).T
        if epi == EPI_SWIGLU:
            g, u = O.bf16_round(full[:, :N // 2].astype(np.float32)), O.bf16_round(full[:, N // 2:].astype(np.float32))
            full = (O.bf16_round(O.silu(g.astype(np.float64)).astype(np.float32)) * u).astype(np.float64)
        ref[off:off + c] = full
        off += c
    scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-9
    err = float(np.max(np.abs(out - ref) / (1.2e-2 * np.abs(ref) + 1e-2 * scale)))
    # un-gathered variant == gathered variant on pre-permuted rows
    out2 = f32(ops.gemm_nt_grouped(bf(a[a_rows]), wi if epi == EPI_SWIGLU else bf(w), tcounts, M, epilogue=epi))
    ok = err < 1.0 and np.array_equal(out, out2)
    return _res(f"gemm_grouped[counts={list(counts)},N={N},K={K},epi={epi}]", bool(ok), max_err_over_tol=err)


def check_moe_block(cfg_name="moe-tiny", T=333):
    """Router -> index -> grouped w1|w3 + SwiGLU -> grouped w2 -> combine(+residual) vs the oracle's MixtralSparseMoeBlock on the
    same bf16 input, token by token where the routing agrees."""
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, 2)
    eng = MistralEncoderEngine.from_state_dict(EncoderConfig.from_dict(cfg), {k: torch.from_numpy(v) for k, v in w.items()}, DEV)
    H = cfg["hidden_size"]
    x, res = rnd((T, H), 71), rnd((T, H), 72)
    tx, h = bf(x), bf(res)
    eng.record_routing = []
    eng._mlp(eng.layers[0], tx, h, eng._workspace(T))
    got = f32(h)
    sel = np.sort(eng.record_routing[0].cpu().numpy(), axis=1)
    ref, sel_ref = O.moe_block(f32(tx), w, "layers.0.block_sparse_moe.", 2, emulate_bf16=True)
    ref = O.bf16_round(f32(bf(res)) + ref)
    same = (sel == np.sort(sel_ref, axis=1)).all(1)
    err = np.abs(got - ref)[same]
    scale = float(np.sqrt(np.mean(ref ** 2)))
    # bf16 outputs: the MoE term is a sum of two rounded products, the residual add rounds once more -> ~2 ulp of the result
    ok = same.mean() > 0.97 and float(err.max()) < 3 * 2 ** -8 * float(np.abs(ref).max()) and float(np.sqrt(np.mean(err ** 2))) < 4e-3 * scale
    return _res(f"moe block [{cfg_name},T={T}] vs oracle", bool(ok), routing_agree=float(same.mean()), max_abs_err=float(err.max()),
                rms_err_over_rms=float(np.sqrt(np.mean(err ** 2))) / scale)


def check_mixtral_golden(cfg_name="moe-tiny"):
    """Mixtral bidirectional encode vs the fixture produced by the reference's modeling_mixtral_gritlm.py: hidden states no further
    from the fp32 reference than the reference's own bf16 run, routing agreement, embeddings within 1e-4 cosine, packed == padded."""
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    eng.record_routing = []
    h = f32(eng.forward(tid, tm))
    routing = np.sort(np.stack([r.cpu().numpy() for r in eng.record_routing]).reshape(len(eng.layers), *ids.shape, 2), axis=-1)
    eng.record_routing = None
    valid = mask.astype(bool)
    ref32, refb = g["last_hidden_state"], g["last_hidden_state_bf16"]
    rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
    out = dict(rel_ours_vs_fp32=rel(h, ref32), rel_refbf16_vs_fp32=_yard()[f"encoder_{cfg_name}/rel_refbf16_vs_fp32"])
    agree = float((routing == np.sort(g["routing"], axis=-1)).all(-1)[:, valid].mean())
    agree_ref = _yard()[f"encoder_{cfg_name}/routing_agree_of_bf16_ref"]
    out["routing_agree_with_fp32_ref"], out["routing_agree_of_bf16_ref"] = agree, agree_ref
    ok = out["rel_ours_vs_fp32"] < 1.5 * out["rel_refbf16_vs_fp32"] + 2e-3 and not np.isnan(h).any() and agree > agree_ref - 0.02
    for method in ("mean", "weightedmean"):
        e = f32(eng.encode_pooled(tid, tm, method, True, packed=False))
        ep = f32(eng.encode_pooled(tid, tm, method, True, packed=True))
        c32 = float(np.max(1 - np.sum(e * g[f"emb_{method}"], axis=1)))
        cref = _yard()[f"encoder_{cfg_name}/1-cos_of_bf16ref_{method}"]                              # the bf16 reference's own distance (frozen)
        out[f"{method}_1-cos"] = c32; out[f"{method}_1-cos_of_bf16ref"] = cref
        ok &= c32 < max(1e-4, 2 * cref) and np.array_equal(e, ep)
    return _res(f"mixtral encoder[{cfg_name}] vs reference golden", bool(ok), **out)


def check_gemm_grouped_fullshape(N=28672, K=4096, epi=EPI_SWIGLU, rows=262144, seed=301, samples=4096):
    """grit_gemm_bf16_nt_grouped at BASELINE configs[3]'s launch shape (Mixtral-8x7B, 64 docs x 2048 tokens, top-2: 262144 sorted rows
    over 8 experts with UNEVEN counts incl. one that is no multiple of the 256-row tile and one tiny group; N 28672 / K 4096 with the
    SwiGLU epilogue and the token gather folded into the A loads, or N 4096 / K 14336 STORE), spot-checked on random (row, column) pairs
    of every group against fp64 dot products of the same bf16 operands (check_gemm_fullshape's tolerance)."""
    E = 8
    gen = torch.Generator(device=DEV).manual_seed(seed)
    frac = torch.tensor([0.16, 0.09, 0.125, 0.14, 0.11, 0.135, 0.1299, 0.0001])
    counts = (frac / frac.sum() * rows).long()
    counts[0] += rows - int(counts.sum())
    counts[3] += 77; counts[0] -= 77                                    # a count that is no multiple of anything
    assert int(counts.sum()) == rows and int(counts.min()) > 0
    T = rows // 2
    a = torch.randn((T, K), device=DEV, generator=gen).to(torch.bfloat16)
    a_rows = torch.randint(0, T, (rows,), device=DEV, generator=gen, dtype=torch.int32)
    w = torch.empty((E, N, K), dtype=torch.bfloat16, device=DEV)
    for e in range(E):
        w[e].copy_((torch.randn((N, K), device=DEV, generator=gen) * 0.03).to(torch.bfloat16))
    tcounts = counts.to(torch.int32).to(DEV)
    I = N // 2
    if epi == EPI_SWIGLU:
        wi = torch.stack([swiglu_interleave(w[e, :I].contiguous(), w[e, I:].contiguous()) for e in range(E)]).contiguous()
        out = ops.gemm_nt_grouped(a, wi, tcounts, rows, epilogue=epi, a_rows=a_rows)
        del wi
    else:
        out = ops.gemm_nt_grouped(a, w, tcounts, rows, a_rows=a_rows)
    ncols = I if epi == EPI_SWIGLU else N
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(counts.numpy())])
    worst, finite = 0.0, bool(torch.isfinite(out).all())
    for e in range(E):
        c = int(counts[e])
        n_s = max(64, samples // E)
        r_loc = np.unique(np.concatenate([rng.integers(0, c, n_s), np.arange(min(c, 16)), np.arange(max(c - 16, 0), c)]))
        rr = torch.from_numpy(off[e] + r_loc).to(DEV)
        cc = torch.from_numpy(rng.integers(0, ncols, rr.numel())).to(DEV)
        cc[:8] = torch.arange(8, device=DEV); cc[-8:] = torch.arange(ncols - 8, ncols, device=DEV)
        a64 = a[a_rows[rr].long()].double()
        if epi == EPI_SWIGLU:
            gt = (a64 * w[e, :I][cc].double()).sum(1).float().to(torch.bfloat16).double()
            ut = (a64 * w[e, I:][cc].double()).sum(1).float().to(torch.bfloat16).double()
            ref = torch.nn.functional.silu(gt).float().to(torch.bfloat16).double() * ut
        else:
            ref = (a64 * w[e][cc].double()).sum(1)
        got = out[rr, cc].double()
        scale = float(ref.pow(2).mean().sqrt()) + 1e-9
        worst = max(worst, float(((got - ref).abs() / (1.2e-2 * ref.abs() + 1e-2 * scale)).max()))
    return _res(f"gemm_grouped_fullshape[rows={rows},N={N},K={K},epi={epi}]", worst < 1.0 and finite, max_err_over_tol=worst,
                counts=counts.tolist())


def check_mixtral_layer_true_shape():
    """ONE layer at the TRUE Mixtral-8x7B layer shape (E 8, H 4096, I 14336, 32/8 heads) vs the fixture the REFERENCE's
    MixtralModel(is_causal=False) produced (tests/golden/encoder_8x7b-l1.npz: scripts/modeling_mixtral_gritlm.py:815-882; fp32 and bf16
    runs, probe rows, the routing of every token and the router's 2nd-vs-3rd margin).  The grouped GEMMs run at configs[3]'s N / K with 8
    uneven expert row counts.  Numeric criteria: (1) every valid token whose router margin (probability of the 2nd minus the 3rd choice
    in the fp32 run) exceeds 0.05 takes the fp32 reference's two experts -- the reference's OWN bf16 run re-routes tokens with margins up
    to 0.036 --, overall agreement >= 0.97 (the reference's bf16 run: 0.984); (2) probe rows that took the reference's experts, relative l2
    error PER ROW: median < 2.0e-2 (the reference's bf16 run: 1.39e-2; the dense 7B layer: 1.47e-2), 90th percentile < 8e-2, all rows
    together < 8e-2 -- the distribution is heavy-tailed in ANY bf16 run (reference: 90th percentile 3.7e-2, maximum 7.0e-2, aggregate
    2.4e-2): the synthetic router's logits have a standard deviation of ~30, so a token whose two best experts lie close takes routing
    WEIGHTS that move by several per cent under bf16 noise in x, and the aggregate is a statement about a handful of such rows; (3) pooled embeddings within
    1.5e-3 of the fp32 reference's (its own bf16 run: 3e-4 .. 7.7e-4 -- tokens that sit on a routing tie go to another expert in ANY
    bf16 run, so the 1e-4 of the dense fixtures does not apply); (4) packed == padded bit for bit."""
    g = np.load(os.path.join(GOLDEN, "encoder_8x7b-l1.npz"))
    eng, cfg, w = build_engine("8x7b-l1", int(g["seed_w"]))
    del w
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    eng.record_routing = []
    h = f32(eng.forward(tid, tm))
    routing = np.sort(eng.record_routing[0].cpu().numpy().reshape(-1, 2), axis=-1)
    eng.record_routing = None
    valid = mask.astype(bool).reshape(-1)
    r32 = np.sort(g["routing"], axis=-1).reshape(-1, 2)
    agree_tok = (routing == r32).all(-1)
    margin = g["router_margin_2nd_vs_3rd"].reshape(-1)
    clear = valid & (margin > 0.05)
    out = dict(routing_agree=float(agree_tok[valid].mean()), routing_agree_of_bf16_ref=float((np.sort(g["routing_bf16"], -1).reshape(-1, 2) == r32).all(-1)[valid].mean()),
               clear_margin_tokens=int(clear.sum()), clear_margin_agree=float(agree_tok[clear].mean()))
    ok = out["clear_margin_agree"] == 1.0 and out["routing_agree"] >= 0.97 and not np.isnan(h).any()
    probe = g["probe_rows"]
    pa = agree_tok[probe]
    hp = h.reshape(-1, h.shape[-1])[probe]
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    out["probe_rows_same_experts"] = int(pa.sum())
    out["rel_ours_vs_fp32"] = rel(hp[pa], g["probe_hidden"][pa]); out["rel_refbf16_vs_fp32"] = rel(g["probe_hidden_bf16"][pa], g["probe_hidden"][pa])
    rows = lambda a, b: np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
    pr, prr = rows(hp[pa], g["probe_hidden"][pa]), rows(g["probe_hidden_bf16"][pa], g["probe_hidden"][pa])
    out["row_rel_median"], out["row_rel_p90"], out["row_rel_max"] = float(np.median(pr)), float(np.quantile(pr, 0.9)), float(pr.max())
    out["row_rel_median_of_bf16ref"], out["row_rel_p90_of_bf16ref"] = float(np.median(prr)), float(np.quantile(prr, 0.9))
    ok &= pa.sum() >= 56 and out["row_rel_median"] < 2.0e-2 and out["row_rel_p90"] < 8e-2 and out["rel_ours_vs_fp32"] < 8e-2
    for method in ("mean", "weightedmean"):
        e = f32(eng.encode_pooled(tid, tm, method, True, packed=False))
        ep = f32(eng.encode_pooled(tid, tm, method, True, packed=True))
        c32 = float(np.max(1 - np.sum(e * g[f"emb_{method}"], axis=1)))
        out[f"{method}_1-cos"] = c32
        out[f"{method}_1-cos_of_bf16ref"] = float(np.max(1 - np.sum(g[f"emb_{method}_bf16"] * g[f"emb_{method}"], axis=1)))
        ok &= c32 < 1.5e-3 and np.array_equal(e, ep)
    return _res("mixtral layer at the true 8x7B shape vs reference golden", bool(ok), **out)


def check_inputs_embeds_and_layer_range(cfg_name="gqa"):
    """`forward(inputs_embeds=...)` (the reference's forward takes it too, modeling_mistral_gritlm.py:944) == the embedding lookup, bit for
    bit; `layer_range` + `final_norm=False` compose: layer 0 alone, then layers 1.. on its output, equals the whole forward (the bf16
    residual stream is what crosses the cut) -- the mechanism of the Mixtral leg's teacher-forced per-layer parity; dense and MoE."""
    det, ok = {}, True
    for name in (cfg_name, "moe-tiny"):
        eng, cfg, w = build_engine(name, 1)
        ids, mask = synth.make_batch(cfg, 3, 70, seed=11, min_len=20)
        tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
        full = eng.forward(tid, tm)
        emb = eng.embed[tid]                                                   # [B,S,H] bf16
        a = eng.forward(None, tm, inputs_embeds=emb)
        h1 = eng.forward(None, tm, inputs_embeds=emb, layer_range=(0, 1), final_norm=False)
        b = eng.forward(None, tm, inputs_embeds=h1, layer_range=(1, len(eng.layers)))
        det[f"{name}_inputs_embeds_identical"] = bool(torch.equal(a, full))
        det[f"{name}_layer_cut_identical"] = bool(torch.equal(b, full))
        ok &= det[f"{name}_inputs_embeds_identical"] and det[f"{name}_layer_cut_identical"]
    return _res("forward(inputs_embeds, layer_range, final_norm) composes to the full forward", ok, **det)


def check_encoder_vs_oracle_bf16(cfg_name="tiny", B=3, S=130):
    """Different shape than the golden (S not a multiple of 64), against the bf16-emulating oracle."""
    eng, cfg, w = build_engine(cfg_name, 5)
    ids, mask = synth.make_batch(cfg, B, S, seed=77, min_len=33)
    h = f32(eng.forward(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)))
    ref = O.mistral_encode(w, cfg, ids, mask, emulate_bf16=True)
    ref32 = O.mistral_encode(w, cfg, ids, mask)
    valid = mask.astype(bool)
    rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
    r1, r2, r3 = rel(h, ref), rel(h, ref32), rel(ref, ref32)
    return _res(f"encoder[{cfg_name},B={B},S={S}] vs oracle", r2 < max(2.0 * r3, 1.5e-2), rel_vs_bf16_oracle=r1, rel_vs_fp32_oracle=r2,
                bf16_oracle_vs_fp32=r3)


def check_gritlm_native_encode():
    """gritlm_amd.GritLM(...).encode() on the GPU (tokenise -> native engine -> fused pool/normalise) vs the
    REFERENCE GritLM.encode() outputs recorded on CPU in fp32 and in bf16 (tests/golden/gritlm_encode.npz)."""
    import tempfile
    from gritlm_amd import GritLM
    g = np.load(os.path.join(GOLDEN, "gritlm_encode.npz"))
    sents = [str(x) for x in g["sentences"]]
    instr = str(g["instruction"]) + " "
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16)
        ok &= m.engine is not None
        for key, kw in (("mean_instr", dict(instruction=instr)), ("mean", {})):
            e = m.encode(sents[:12], batch_size=5, max_length=64, **kw)
            ok &= e.dtype == np.float32 and e.shape == (12, 256)
            r32, r16 = g[f"mistral_fp32_{key}"], g[f"mistral_bf16_{key}"]
            c32 = float(np.max(1 - np.sum(e * r32, axis=1))); c16 = float(np.max(1 - np.sum(e * r16, axis=1)))
            pd = float(np.max(np.abs(e @ e.T - r32 @ r32.T))); pd_ref = _yard()[f"gritlm_encode/pair_delta_of_bf16ref_{key}"]
            out[f"{key}_1-cos_fp32ref"] = c32; out[f"{key}_1-cos_bf16ref"] = c16
            out[f"{key}_pair_delta"] = pd; out[f"{key}_pair_delta_of_bf16ref"] = pd_ref
            ok &= c32 < 1e-4 and c16 < 1e-4 and pd < 1.5 * pd_ref + 1e-4
        t = m.encode(sents[:3], max_length=64, convert_to_tensor=True)
        ok &= t.is_cuda and t.dtype == torch.float32
        m.pooling_method = "weightedmean"
        mask = torch.ones((2, 5), dtype=torch.int64, device="cuda"); mask[1, 3:] = 0
        m.pooling(torch.randn((2, 5, 256), device="cuda").to(torch.bfloat16), mask)
        ok &= mask.cpu().tolist() == [[1, 2, 3, 4, 5], [1, 2, 3, 0, 0]]        # in-place side effect kept (:211)
    return _res("GritLM.encode native vs reference GritLM.encode goldens", ok, **out)


# ---------------------------------------------------------------------------------------------- fp16-operand precision policy (round 5)
def f16r(x: np.ndarray) -> np.ndarray:
    """fp32 values rounded to IEEE fp16 (RNE), returned as fp32"""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def fh(x: np.ndarray) -> torch.Tensor:
    """fp16-representable fp32 numpy -> fp16 cuda tensor (exact)"""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV).to(torch.float16)


def _f16_flag(clear=True) -> bool:
    return ops.f16_overflow_flag(torch.device(DEV, torch.cuda.current_device()), clear)


def check_gemm_f16(M, N, K, epi=EPI_STORE, seed=107, subnormal_weights=False):
    """grit_gemm_f16_nt against fp64 products of the SAME fp16 operands.  STORE: one fp16 rounding of the accumulator (1 ulp = 2^-11 of
    the value); SWIGLU: fp16(silu(gate) * up) evaluated in fp32 on the accumulators (ONE rounding, unlike the bf16 epilogue's four);
    RESIDUAL_F32: fp32 out, nothing rounded.  subnormal_weights: a third of the weights below 2^-14 -- the fp16 MFMA must not flush its
    subnormal inputs (bf16 checkpoints hold such weights; the engine counts them)."""
    rng = np.random.default_rng(seed)
    a = f16r(rng.standard_normal((M, K), dtype=np.float32))
    w = rng.standard_normal((N, K), dtype=np.float32) * 0.05
    if subnormal_weights:
        w[rng.random((N, K)) < 0.33] *= 2.0 ** -13                   # |w| ~ 6e-6: fp16 subnormals (spacing 6e-8)
    w = f16r(w)
    n_sub = int(((np.abs(w) < 2.0 ** -14) & (w != 0)).sum())
    _f16_flag()
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    if epi == EPI_RESIDUAL_F32:
        r = rng.standard_normal((M, N)).astype(np.float32) * 3.0
        rt = torch.from_numpy(r).to(DEV)
        ops.gemm_nt(fh(a), fh(w), out=rt, epilogue=epi, residual=rt)
        got = rt.cpu().numpy().astype(np.float64)
        ref = ref + r
        scale = float(np.sqrt(np.mean(ref ** 2)))
        err = float(np.max(np.abs(got - ref)) / scale) / 2e-5
    elif epi == EPI_RESIDUAL:             # fp16 residual stream: C = f16(f16(acc) + residual), in place
        r = f16r(rng.standard_normal((M, N)).astype(np.float32) * 3.0)
        rt = fh(r)
        ops.gemm_nt(fh(a), fh(w), out=rt, epilogue=epi, residual=rt)
        assert rt.dtype == torch.float16
        got = rt.float().cpu().numpy().astype(np.float64)
        exact = f16r((f16r(ref.astype(np.float32)).astype(np.float64) + r).astype(np.float32)).astype(np.float64)   # the two roundings, from the fp64 product
        scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-12
        tol = 2.0 ** -11 * (np.abs(ref) + np.abs(ref + r)) + 6.0e-8 + 2e-5 * scale
        err = float(np.max(np.abs(got - (ref + r)) / tol))
        err = max(err, 0.0 if float(np.mean(got == exact)) > 0.99 else 2.0)                                           # and bit-equal to the emulated roundings almost everywhere
    else:
        if epi == EPI_SWIGLU:
            I = N // 2
            wi = swiglu_interleave(fh(w[:I]), fh(w[I:]))
            out = ops.gemm_nt(fh(a), wi, epilogue=epi)
            g, u = ref[:, :I], ref[:, I:]
            ref = g / (1.0 + np.exp(-g)) * u
        else:
            out = ops.gemm_nt(fh(a), fh(w))
        assert out.dtype == torch.float16
        got = out.float().cpu().numpy().astype(np.float64)
        scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-12
        # fp16 output: half an ulp = 2^-12 of the value (+ the subnormal spacing); fp32 accumulation and exp: 1e-5 of the rms
        err = float(np.max(np.abs(got - ref) / (2.0 ** -11 * np.abs(ref) + 6.0e-8 + 2e-5 * scale)))
    flag = _f16_flag()
    ok = err < 1.0 and not flag
    return _res(f"gemm_f16[M={M},N={N},K={K},epi={epi},sub={int(subnormal_weights)}]", ok, max_err_over_tol=err, subnormal_weights=n_sub, overflow_flag=flag)


def check_gemm_f16_rope(M=300, nq=4, nkv=2, K=256, S=77, packed=False):
    """grit_gemm_f16_nt_rope: q | k rotated in fp32 on the ACCUMULATORS with unrounded fp32 tables, one fp16 rounding -- against the fp64
    product rotated in fp64 with the same fp32 tables; v heads = the plain fp16 GEMM, bit for bit."""
    from gritlm_amd.encoder import rope_tables
    d = 128
    N = (nq + 2 * nkv) * d
    rng = np.random.default_rng(171)
    a, w = f16r(rng.standard_normal((M, K), dtype=np.float32)), f16r(rng.standard_normal((N, K), dtype=np.float32) * 0.05)
    cos, sin = rope_tables(max(S, 128), d, 10000.0, False, DEV)
    if packed:
        pos_np = (np.arange(M) * 7 % max(S, 128)).astype(np.int32)
        got = ops.gemm_nt_rope(fh(a), fh(w), cos, sin, (nq + nkv) * d, positions=torch.from_numpy(pos_np).to(DEV))
    else:
        pos_np = (np.arange(M) % S).astype(np.int32)
        got = ops.gemm_nt_rope(fh(a), fh(w), cos, sin, (nq + nkv) * d, S=S)
    plain = ops.gemm_nt(fh(a), fh(w))
    acc = (a.astype(np.float64) @ w.astype(np.float64).T).reshape(M, nq + 2 * nkv, d)
    c = cos.cpu().numpy().astype(np.float64)[pos_np][:, None, :]
    sn = sin.cpu().numpy().astype(np.float64)[pos_np][:, None, :]
    x1, x2 = acc[:, :nq + nkv, :d // 2], acc[:, :nq + nkv, d // 2:]
    ref = acc.copy()
    ref[:, :nq + nkv, :d // 2] = x1 * c - x2 * sn
    ref[:, :nq + nkv, d // 2:] = x2 * c + x1 * sn
    ref = ref.reshape(M, N)
    g64 = got.float().cpu().numpy().astype(np.float64)
    scale = float(np.sqrt(np.mean(ref ** 2)))
    err = float(np.max(np.abs(g64 - ref) / (2.0 ** -11 * np.abs(ref) + 6.0e-8 + 2e-5 * scale)))
    v_same = bool(torch.equal(got[:, (nq + nkv) * d:], plain[:, (nq + nkv) * d:]))
    return _res(f"gemm_f16+rope [M={M},nq={nq},nkv={nkv},K={K},packed={int(packed)}]", err < 1.0 and v_same and got.dtype == torch.float16,
                max_err_over_tol=err, v_heads_identical=v_same)


def check_gemm_f16_fullshape(M, N, K, epi=EPI_STORE, seed=191, samples=4096):
    """Full-size fp16 launches (the persistent form, the bench's N and K) spot-checked on random outputs against fp64 dot products."""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.randn((M, K), device=DEV, generator=gen).to(torch.float16)
    w = (torch.randn((N, K), device=DEV, generator=gen) * 0.03).to(torch.float16)
    rng = np.random.default_rng(seed)
    rows = torch.from_numpy(np.concatenate([rng.integers(0, M, samples - 64), np.arange(M - 32, M), np.arange(32)])).to(DEV)
    I = N // 2
    ncols = I if epi == EPI_SWIGLU else N
    cols = torch.from_numpy(np.concatenate([rng.integers(0, ncols, samples - 64), np.arange(ncols - 32, ncols), np.arange(32)])).to(DEV)
    a64 = a[rows].double()
    _f16_flag()
    if epi == EPI_SWIGLU:
        out = ops.gemm_nt(a, swiglu_interleave(w[:I].contiguous(), w[I:].contiguous()), epilogue=epi)
        gt, ut = (a64 * w[:I][cols].double()).sum(1), (a64 * w[I:][cols].double()).sum(1)
        ref = torch.nn.functional.silu(gt) * ut
        got = out[rows, cols].double()
    elif epi == EPI_RESIDUAL_F32:
        res = torch.randn((M, N), device=DEV, generator=gen) * 2.0
        ref = (a64 * w[cols].double()).sum(1) + res[rows, cols].double()
        out = ops.gemm_nt(a, w, out=res, epilogue=epi, residual=res)
        got = out[rows, cols].double()
    else:
        out = ops.gemm_nt(a, w)
        ref = (a64 * w[cols].double()).sum(1)
        got = out[rows, cols].double()
    scale = float(ref.pow(2).mean().sqrt()) + 1e-12
    tol = (2e-5 * scale) if epi == EPI_RESIDUAL_F32 else (2.0 ** -11 * ref.abs() + 6.0e-8 + 2e-5 * scale)
    err = float(((got - ref).abs() / tol).max())
    flag = _f16_flag()
    return _res(f"gemm_f16_fullshape[M={M},N={N},K={K},epi={epi}]", err < 1.0 and bool(torch.isfinite(out).all()) and not flag, max_err_over_tol=err,
                samples=samples, overflow_flag=flag)


def check_f16_overflow_flag():
    """A result beyond the fp16 range must set the device's overflow flag (and only then): GEMM STORE / SWIGLU / RoPE epilogues and the
    fp16 RMSNorm; the flag is sticky until cleared; RESIDUAL_F32 (fp32 out) never sets it."""
    from gritlm_amd.encoder import rope_tables
    det, ok = {}, True
    M, N, K = 300, 512, 128
    small = fh(np.full((M, K), 1.0, dtype=np.float32))
    big = fh(np.full((M, K), 200.0, dtype=np.float32))
    w = fh(np.full((N, K), 4.0, dtype=np.float32))              # 128 * 200 * 4 = 102400 > 65504; 128 * 1 * 4 = 512
    _f16_flag()
    ops.gemm_nt(small, w)
    det["store_in_range"] = _f16_flag(); ok &= not det["store_in_range"]
    o = ops.gemm_nt(big, w)
    det["store_overflow"] = _f16_flag(clear=False); ok &= det["store_overflow"] and bool(torch.isinf(o).any())
    det["sticky"] = _f16_flag(); ok &= det["sticky"]
    det["cleared"] = _f16_flag(); ok &= not det["cleared"]
    h = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    ops.gemm_nt(big, w, out=h, epilogue=EPI_RESIDUAL_F32, residual=h)
    det["residual_f32_no_flag"] = _f16_flag(); ok &= not det["residual_f32_no_flag"] and bool(torch.isfinite(h).all())
    ops.gemm_nt(big, swiglu_interleave(w[:N // 2].contiguous(), w[N // 2:].contiguous()), epilogue=EPI_SWIGLU)   # silu(102400) * 102400
    det["swiglu_overflow"] = _f16_flag(); ok &= det["swiglu_overflow"]
    cos, sin = rope_tables(128, 128, 10000.0, False, DEV)
    ops.gemm_nt_rope(big, w, cos, sin, 256, S=100)
    det["rope_overflow"] = _f16_flag(); ok &= det["rope_overflow"]
    stream = fh(np.full((M, N), 6.54e4, dtype=np.float32))                     # fp16 residual stream near the top of the range: 65408 + 512 > 65504
    ops.gemm_nt(small, w, out=stream, epilogue=EPI_RESIDUAL, residual=stream)
    det["f16_stream_residual_overflow"] = _f16_flag(); ok &= det["f16_stream_residual_overflow"]
    stream = fh(np.full((M, N), 1.0e3, dtype=np.float32))
    ops.gemm_nt(small, w, out=stream, epilogue=EPI_RESIDUAL, residual=stream)
    det["f16_stream_residual_in_range"] = _f16_flag(); ok &= not det["f16_stream_residual_in_range"]
    x = torch.full((5, 256), 3.0, dtype=torch.float32, device=DEV)
    wn = torch.full((256,), 3.0e4, dtype=torch.bfloat16, device=DEV)           # 1.0 * 3e4 fits; x row has rms 3 -> normalised 1.0
    y = torch.empty((5, 256), dtype=torch.float16, device=DEV)
    ops.rmsnorm(x, wn, 1e-5, out=y)
    det["rmsnorm_in_range"] = _f16_flag(); ok &= not det["rmsnorm_in_range"]
    x[2, 7] = 1.0e3                                                             # one outlier: normalised ~ 15.6 x 3e4 > 65504
    ops.rmsnorm(x, wn, 1e-5, out=y)
    det["rmsnorm_overflow"] = _f16_flag(); ok &= det["rmsnorm_overflow"]
    # round 6: the fp16-stream norm (fp16 rows in) and the grouped fp16 GEMM (MoE experts), STORE and the stacked SwiGLU
    from gritlm_amd._lib import EPI_SWIGLU_STACKED
    xh = torch.full((5, 512), 3.0, dtype=torch.float16, device=DEV)
    wn5 = torch.full((512,), 3.0e4, dtype=torch.bfloat16, device=DEV)
    yh = torch.empty((5, 512), dtype=torch.float16, device=DEV)
    ops.rmsnorm(xh, wn5, 1e-5, out=yh)
    det["rmsnorm_f16in_in_range"] = _f16_flag(); ok &= not det["rmsnorm_f16in_in_range"]
    xh[2, 7] = 1.0e3
    ops.rmsnorm(xh, wn5, 1e-5, out=yh)
    det["rmsnorm_f16in_overflow"] = _f16_flag(); ok &= det["rmsnorm_f16in_overflow"]
    cnt = torch.tensor([200, 100], dtype=torch.int32, device=DEV)
    w3 = torch.stack([w, w]).contiguous()
    ops.gemm_nt_grouped(small, w3, cnt, M)
    det["grouped_in_range"] = _f16_flag(); ok &= not det["grouped_in_range"]
    mixed = small.clone(); mixed[250:] = 200.0                                 # only rows of the SECOND group overflow
    ops.gemm_nt_grouped(mixed, w3, cnt, M)
    det["grouped_store_overflow"] = _f16_flag(); ok &= det["grouped_store_overflow"]
    ops.gemm_nt_grouped(mixed, w3, cnt, M, epilogue=EPI_SWIGLU_STACKED)
    det["grouped_swiglu_stacked_overflow"] = _f16_flag(); ok &= det["grouped_swiglu_stacked_overflow"]
    return _res("fp16 overflow flag (set by STORE / SWIGLU / RoPE / RMSNorm / grouped GEMM, sticky, clearable)", ok, **det)


def check_f16_stream_ops(T=37, H=4096):
    """grit_rmsnorm_fwd_f32in_f16: fp32 rows -> fp16, ONE rounding, vs fp64."""
    x = np.random.default_rng(3).standard_normal((T, H)).astype(np.float32) * 2.0
    w = O.bf16_round(1 + 0.1 * rnd((H,), 4))
    y = torch.empty((T, H), dtype=torch.float16, device=DEV)
    ops.rmsnorm(torch.from_numpy(x).to(DEV), bf(w), 1e-5, out=y)
    y = y.float().cpu().numpy()
    x64 = x.astype(np.float64)
    ref = w * (x64 * (1.0 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5)))
    exact = float(np.mean(y == f16r(ref.astype(np.float32))))
    err = float(np.max(np.abs(y - ref) / (np.abs(ref) + 1e-3)))
    return _res(f"f16_stream_ops[T={T},H={H}]", err < 6e-4 and exact > 0.99, rms_max_rel=err, rms_exact_frac=exact)


def check_f16_stream_norm_and_gather(T=37, H=4096):
    """The fp16 residual stream's own kernels: grit_rmsnorm_fwd_f16in (fp16 rows -> fp16 operand / bf16 last_hidden_state, ONE rounding,
    vs fp64) and the embedding gather on an fp16 copy of the table (bit-exact rows)."""
    rng = np.random.default_rng(13)
    x = f16r(rng.standard_normal((T, H)).astype(np.float32) * 2.0)
    w = O.bf16_round(1 + 0.1 * rnd((H,), 4))
    x64 = x.astype(np.float64)
    ref = w * (x64 * (1.0 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-5)))
    det, ok = {}, True
    for dt, rnd_fn, name, tol in ((torch.float16, f16r, "f16", 6e-4), (torch.bfloat16, O.bf16_round, "bf16", 5e-3)):
        y = torch.empty((T, H), dtype=dt, device=DEV)
        ops.rmsnorm(fh(x), bf(w), 1e-5, out=y)
        y = y.float().cpu().numpy()
        det[f"{name}_exact_frac"] = float(np.mean(y == rnd_fn(ref.astype(np.float32))))
        det[f"{name}_max_rel"] = float(np.max(np.abs(y - ref) / (np.abs(ref) + 1e-3)))
        ok &= det[f"{name}_exact_frac"] > 0.99 and det[f"{name}_max_rel"] < tol
    tab = f16r(rng.standard_normal((97, H)).astype(np.float32))
    ids = rng.integers(0, 97, size=(T,))
    out = torch.empty((T, H), dtype=torch.float16, device=DEV)
    ops.embed_gather(fh(tab), torch.from_numpy(ids).to(DEV), out=out)
    det["gather_exact"] = bool(np.array_equal(out.float().cpu().numpy(), tab[ids])); ok &= det["gather_exact"]
    return _res(f"f16 stream: rmsnorm_f16in + gather [T={T},H={H}]", ok, **det)


def check_attention_f16(B=2, S=200, nq=4, nkv=2, mask_kind="ragged", seed=211, packed_lens=None, causal=False, window=0):
    """grit_attn_bidir_f16_fwd / _varlen_f16_fwd (and, ``causal``, grit_attn_causal_f16_fwd / _varlen_f16_fwd, optionally windowed) vs the
    fp64 oracle on the same fp16 inputs: the output carries one fp16 rounding of P (2^-12 relative per term, averaging down) and one of
    O; LSE is fp32.  With packed_lens: the packed launch must reproduce the padded rows bit for bit (what keeps packed == padded in the
    f16 policy)."""
    d = 128
    width = (nq + 2 * nkv) * d
    rng = np.random.default_rng(seed)
    qkv = f16r(rng.standard_normal((B * S, width), dtype=np.float32))
    mask = np.ones((B, S), dtype=np.int64)
    if packed_lens is not None:
        for b, L in enumerate(packed_lens):
            mask[b, L:] = 0
    elif mask_kind == "ragged":
        for b in range(1, B):
            mask[b, rng.integers(S // 3, S):] = 0
    elif mask_kind == "holes":
        mask = (rng.random((B, S)) < 0.7).astype(np.int64); mask[:, 0] = 1
    x = qkv.reshape(B, S, nq + 2 * nkv, d).transpose(0, 2, 1, 3)
    q, k, v = x[:, :nq], x[:, nq:nq + nkv], x[:, nq + nkv:]
    ref = O.attention_bidirectional(q, k, v, mask, causal=causal, window=window)
    lse_t = torch.empty((B, nq, S), dtype=torch.float32, device=DEV)
    bits = ops.mask_pack(torch.from_numpy(mask).to(DEV))
    t = fh(qkv)
    out_t = ops.attn_bidir(t, bits, B, S, nq, nkv, d, lse=lse_t, causal=causal, window=window)
    out = out_t.float().cpu().numpy().reshape(B, S, nq * d)
    kk = np.repeat(k, nq // nkv, axis=1)
    sc = np.einsum("bhqd,bhkd->bhqk", q.astype(np.float64), kk.astype(np.float64)) / np.sqrt(d)
    allowed = np.broadcast_to(mask.astype(bool)[:, None, None, :], sc.shape)
    if causal:
        allowed = allowed & O.causal_window_mask(S, window)[None, None]
    sc = np.where(allowed, sc, -np.inf)
    sees = allowed.any(-1)                      # (a padding query behind a sliding window sees no key: output 0, lse -inf)
    mx = np.where(sees, sc.max(-1), 0.0)[..., None]
    with np.errstate(divide="ignore"):
        lse_ref = mx[..., 0] + np.log(np.exp(sc - mx).sum(-1))
    err = float(np.max(np.abs(out - ref)))
    lerr = float(np.max(np.abs(f32(lse_t) - lse_ref)[sees]))
    det = dict(max_abs=err, lse_abs=lerr)
    ok = err < 2.5e-3 and lerr < 2e-3 and not np.isnan(out).any() and out_t.dtype == torch.float16      # (bf16 kernel: 2e-2)
    if packed_lens is not None:
        keep = torch.from_numpy(mask.astype(bool).reshape(-1)).to(DEV)
        cu = torch.zeros((B + 1,), dtype=torch.int32, device=DEV)
        cu[1:] = torch.cumsum(torch.tensor(packed_lens, dtype=torch.int32, device=DEV), 0)
        po = ops.attn_bidir_varlen(t[keep].contiguous(), cu, max(packed_lens), nq, nkv, d, causal=causal, window=window)
        det["packed_identical"] = bool(torch.equal(po, out_t[keep]))
        ok &= det["packed_identical"]
    return _res(f"attention_f16[B={B},S={S},nq={nq},nkv={nkv},{mask_kind if packed_lens is None else 'packed'},causal={int(causal)},window={window}]", ok, **det)


def check_encoder_causal_f16(cfg_name="gqa", B=3, S=96):
    """The engine with CAUSAL attention ('cc' embedding attention; the prompt pass of a unified / generative model) under the fp16
    policies against the fp32 oracle: hidden states at least 10x closer than the bf16 policy on the same inputs, pooled lasttoken /
    weightedmean embeddings within 1e-5 (1 - cos), padded == packed bit for bit, K/V handed out in fp16."""
    eng, cfg, w = build_engine(cfg_name, 4)
    ids, mask = synth.make_batch(cfg, B, S, seed=61, min_len=S // 3)
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    ref = O.mistral_encode(w, cfg, ids, mask, causal=True)
    valid = mask.astype(bool)
    rel = lambda a: float(np.linalg.norm((a - ref)[valid]) / np.linalg.norm(ref[valid]))
    eng.causal = True
    out, ok = {}, True
    eng.set_precision("bf16")
    out["rel_hidden_bf16"] = rel(f32(eng.forward(tid, tm)))
    emb_ref = {m: O.l2_normalize(O.pooling(ref, mask, m)) for m in ("lasttoken", "weightedmean")}
    for pol in ("f16_operands", "f16_stream"):
        eng.set_precision(pol)
        _f16_flag()
        h, kv = eng.forward(tid, tm, return_kv=True, kv_dtype=None)
        out[f"rel_hidden_{pol}"] = rel(f32(h))
        ok &= kv[0][0].dtype == torch.float16 and not _f16_flag()
        for m in ("lasttoken", "weightedmean"):
            e_pad = eng.encode_pooled(tid, tm, m, True, packed=False)
            e_pack = eng.encode_pooled(tid, tm, m, True, packed=True)
            dcos = float(np.max(1 - np.sum(f32(e_pad).astype(np.float64) * emb_ref[m].astype(np.float64), axis=1)))
            out[f"{m}_1-cos_{pol}"] = dcos
            ok &= dcos < 1e-5 and bool(torch.equal(e_pad, e_pack))
    # last_hidden_state is bf16 under every policy (the pooling kernels' input format): its rounding floors the per-element error
    ok &= out["rel_hidden_f16_operands"] <= out["rel_hidden_bf16"] and out["rel_hidden_f16_stream"] <= out["rel_hidden_bf16"]
    return _res(f"causal engine under the fp16 policies [{cfg_name}] vs the fp32 oracle", bool(ok), **out)


def check_encoder_f16_operands(cfg_name, policy="f16_operands"):
    """The engine under precision='f16_operands' (fp32 residual stream, every MFMA operand fp16) against the reference's FP32 run
    (reference-generated fixture): pooled embeddings within 1e-5 of the fp32 reference (north-star: 1e-4) AND at least 10x closer than
    the fp32-residual policy with bf16 operands on the same inputs (the emulation, profiles/r05_precision_budget.json, predicts ~64x in
    1 - cos: 3 mantissa bits), hidden states no further than that policy's (both carry the bf16 rounding of last_hidden_state, the
    pooling kernels' input format, which dominates the per-element error of the fp16 policy), padded == packed bit for bit, no overflow
    flagged, and the weight conversion exact apart from the counted subnormals."""
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    valid = mask.astype(bool)
    if "last_hidden_state" in g.files:
        ref32 = g["last_hidden_state"]
        rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
        pick = lambda h: h
    else:                                                  # 7b-l1: probe rows only
        ref32 = g["probe_hidden"]
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        pick = lambda h: h.reshape(-1, h.shape[-1])[g["probe_rows"]]
    eng.set_precision("fp32_residual")
    r_b = rel(pick(f32(eng.forward(tid, tm))), ref32)
    omc = lambda e, m: float(np.max(1 - np.sum(f32(e).astype(np.float64) * g[f"emb_{m}"].astype(np.float64), axis=1)))
    d_b = {m: omc(eng.encode_pooled(tid, tm, m, True, packed=False), m) for m in ("mean", "weightedmean")}
    eng.set_precision(policy)
    _f16_flag()
    h_f = pick(f32(eng.forward(tid, tm)))
    r_f = rel(h_f, ref32)
    out = dict(rel_hidden_bf16_operands=r_b, rel_hidden_f16_operands=r_f)
    ok = r_f <= r_b and not np.isnan(h_f).any()
    for method in ("mean", "weightedmean"):
        e_pad = eng.encode_pooled(tid, tm, method, True, packed=False)
        e_pack = eng.encode_pooled(tid, tm, method, True, packed=True)
        d = omc(e_pad, method)
        out[f"{method}_1-cos"] = d; out[f"{method}_1-cos_bf16_operands"] = d_b[method]
        ok &= d < 1e-5 and d < 0.1 * d_b[method] and bool(torch.equal(e_pad, e_pack))
    st = eng.f16_weight_stats
    out["subnormal_weights"], out["overflow_flag"] = st["subnormal"], _f16_flag()
    ok &= st["overflow"] == 0 and not out["overflow_flag"]
    try:
        eng.check_f16_overflow()
        out["check_passes"] = True
    except Exception:      # noqa: BLE001
        out["check_passes"] = False
    ok &= out["check_passes"]
    return _res(f"encoder[{cfg_name}] {policy} policy vs reference fp32", ok, **out)


def check_f16_policy_raises_on_overflow():
    """An activation beyond the fp16 range must surface as an error, not as a saturated embedding: the tiny model with its down_proj input
    scaled up (gate / up weights x 300 -> SwiGLU activations ~1e5) raises from check_f16_overflow() and from GritLM.encode(); the default
    policy on the same weights is unaffected; causal engines take the policy (round 6: grit_attn_causal_f16_fwd) and raise alike,
    sparse-MoE engines take 'f16_operands' only."""
    from gritlm_amd._lib import GritHipError
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, 3)
    for k in list(w):
        if "gate_proj" in k or "up_proj" in k:
            w[k] = O.bf16_round(w[k] * 300.0)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    eng = MistralEncoderEngine.from_state_dict(EncoderConfig.from_dict(cfg), sd, DEV)
    ids = torch.from_numpy(synth.make_batch(cfg, 3, 40, seed=5)[0]).to(DEV)
    mask = torch.ones_like(ids)
    det, ok = {}, True
    e = eng.encode_pooled(ids, mask, "mean", True)
    det["bf16_policy_finite"] = bool(torch.isfinite(e).all()); ok &= det["bf16_policy_finite"]
    eng.set_precision("f16_operands")
    _f16_flag()
    eng.encode_pooled(ids, mask, "mean", True)
    try:
        eng.check_f16_overflow()
        det["raised"] = False
    except GritHipError as ex:
        det["raised"] = "fp16 range" in str(ex)
    ok &= det["raised"] is True
    try:
        eng.check_f16_overflow()                       # the flag was cleared by the failing check
        det["cleared_after_raise"] = True
    except GritHipError:
        det["cleared_after_raise"] = False
    ok &= det["cleared_after_raise"]
    eng.causal = True
    eng.set_precision("f16_operands")
    eng.encode_pooled(ids, mask, "lasttoken", True)
    try:
        eng.check_f16_overflow(); det["causal_raised"] = False
    except GritHipError:
        det["causal_raised"] = True
    ok &= det["causal_raised"] and eng.supported_precisions()[0] == "f16_stream"
    meng, _, _ = build_engine("moe-tiny", 0)
    try:
        meng.set_precision("f16_stream"); det["moe_refuses_f16_stream"] = False          # (the sparse-MoE engine routes on the fp32 stream:
    except GritHipError:                                                                  #  'f16_operands' only, round 6)
        det["moe_refuses_f16_stream"] = True
    ok &= det["moe_refuses_f16_stream"]
    return _res("f16_operands: overflow raises (bidirectional and causal engines), MoE engines refuse f16_stream", ok, **det)


def check_gritlm_f16_operands():
    """GritLM(..., precision='f16_operands').encode() through the drop-in API vs the REFERENCE GritLM.encode() fp32 outputs
    (tests/golden/gritlm_encode.npz): 1 - cos < 1e-5 (the default policy is held to 1e-4), and closer to the fp32 reference than the
    default policy on every row set."""
    import tempfile
    from gritlm_amd import GritLM
    g = np.load(os.path.join(GOLDEN, "gritlm_encode.npz"))
    sents = [str(x) for x in g["sentences"]]
    instr = str(g["instruction"]) + " "
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m0 = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16)
        m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16, precision="f16_operands")
        ok &= m.engine is not None and m.engine.precision == "f16_operands" and m0.engine.precision == "bf16"
        for key, kw in (("mean_instr", dict(instruction=instr)), ("mean", {})):
            e = m.encode(sents[:12], batch_size=5, max_length=64, **kw)
            e0 = m0.encode(sents[:12], batch_size=5, max_length=64, **kw)
            r32 = g[f"mistral_fp32_{key}"].astype(np.float64)
            c = float(np.max(1 - np.sum(e * r32, axis=1))); c0 = float(np.max(1 - np.sum(e0 * r32, axis=1)))
            out[f"{key}_1-cos_f16"] = c; out[f"{key}_1-cos_default"] = c0
            ok &= e.dtype == np.float32 and c < 1e-5 and c <= c0
        try:
            GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16, precision="fp8")
            out["bad_precision_rejected"] = False
        except ValueError:
            out["bad_precision_rejected"] = True
        ok &= out["bad_precision_rejected"]
    return _res("GritLM(precision='f16_operands').encode vs reference fp32 goldens", ok, **out)



def check_moe_router_bwd(T=1531, H=512, E=8, aux=True):
    """grit_moe_router_bwd / grit_moe_router_wgrad (csrc/moe.hip) against torch autograd of the reference's routing arithmetic
    (scripts/modeling_mixtral_gritlm.py:843-849: softmax in fp32 -> top-2 -> renormalise) through the gate Linear, in fp32 on the same
    bf16 operands -- the torch expression the training engine evaluated until round 3.  dlogits to 2e-5 relative, dx and the gate gradient
    to one bf16 ulp of their values; a second run must reproduce the gate gradient bit for bit (fixed two-level summation order)."""
    g = torch.Generator(device=DEV).manual_seed(71)
    x = (torch.randn((T, H), generator=g, device=DEV) * 1.0).to(torch.bfloat16)
    wg = (torch.randn((E, H), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    dw = torch.randn((T, 2), generator=g, device=DEV)
    dx_in = torch.randn((T, H), generator=g, device=DEV).to(torch.bfloat16)
    auxg = torch.randn((T, E), generator=g, device=DEV) * 0.1 if aux else None
    grad0 = (torch.randn((E, H), generator=g, device=DEV) * 0.5).to(torch.bfloat16)
    experts = ops.moe_route(x, wg)[0]
    with torch.enable_grad():
        xf = x.float().requires_grad_(True)
        wf = wg.float().requires_grad_(True)
        logits = xf @ wf.t()
        logits.retain_grad()
        p = torch.softmax(logits, dim=-1)
        sel = torch.gather(p, 1, experts.to(torch.int64))
        w = sel / sel.sum(dim=-1, keepdim=True)
        obj = (w * dw).sum()
        if aux:
            obj = obj + (logits * auxg).sum()
        obj.backward()
    dx_ref = (dx_in.float() + xf.grad)
    g_ref = grad0.float() + wf.grad.to(torch.bfloat16).float()
    outs = []
    for _ in range(2):
        gg = grad0.clone()
        dx, dl = ops.moe_router_bwd(x, wg, experts, dw, dx_in, gg, aux_dlogits=auxg, return_dlogits=True)
        outs.append((dx, dl, gg))
    dx, dl, gg = outs[0]
    e_dl = float((dl - logits.grad).abs().max() / logits.grad.abs().max())
    e_dx = float(((dx.float() - dx_ref).abs() / (dx_ref.abs() * 2 ** -7 + 1e-3)).max())         # in bf16 ulps of the value (+ a floor)
    e_g = float(((gg.float() - g_ref).abs() / (g_ref.abs() * 2 ** -7 + 2e-2 * wf.grad.abs().mean())).max())
    same = bool(torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0]))
    # no dx_in: the gate's input gradient alone
    dx_only = ops.moe_router_bwd(x, wg, experts, dw, None, grad0.clone(), aux_dlogits=auxg)
    e_only = float(((dx_only.float() - xf.grad).abs() / (xf.grad.abs() * 2 ** -7 + 1e-3)).max())
    ok = e_dl < 2e-5 and e_dx <= 1.0 and e_g <= 1.0 and e_only <= 1.0 and same
    return _res(f"moe_router_bwd[T={T},H={H},E={E},aux={aux}]", ok, dlogits_rel=e_dl, dx_ulps=e_dx, gate_grad_ulps=e_g, dx_only_ulps=e_only,
                reproducible=same)


def check_gritlm_multi_gpu_in_process():
    """In-process multi-GPU encode (gritlm/gritlm.py:69-75, :106-107: ONE GritLM in ONE process over every GPU, batch_size x num_gpus):
    mode='embedding' with a device string that names no index builds one engine replica per listed GPU.  A one-GPU box can run (a) the
    one-element list (no replicas, num_gpus 1), (b) TWO replicas that both live on cuda:0 -- a real `replica()` copy of the weights, the
    DataParallel row split, both launch sequences issued back to back with no synchronisation, the cross-replica concatenation -- and
    must get the single-engine embeddings back BIT FOR BIT (batch-invariant kernels, host-side geometry), in sentence order."""
    import tempfile
    from gritlm_amd import GritLM
    sents = synth.make_sentences(23, seed=9, min_words=2, max_words=40)
    instr = "Represent the sentence: "
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m1 = GritLM(d16, mode="embedding", pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, devices=["cuda:0"])
        out["one_device_list"] = bool(m1.engine is not None and m1.num_gpus == 1 and m1.engines == [])
        ok &= out["one_device_list"]
        base = m1.encode(sents, batch_size=8, max_length=64, instruction=instr)
        m2 = GritLM(d16, mode="embedding", pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, devices=["cuda:0"])
        rep = m2.engine.replica("cuda:0")
        out["replica_is_a_copy"] = bool(rep.embed.data_ptr() != m2.engine.embed.data_ptr() and torch.equal(rep.layers[0].wqkv, m2.engine.layers[0].wqkv))
        ok &= out["replica_is_a_copy"]
        m2.engines, m2.num_gpus = [m2.engine, rep], 2
        two = m2.encode(sents, batch_size=4, max_length=64, instruction=instr)          # 4 x 2 replicas: the same batches of 8
        out["bit_identical_to_one_engine"] = bool(np.array_equal(base, two))
        t = m2.encode(sents[:5], batch_size=2, max_length=64, convert_to_tensor=True)
        out["odd_split_max_abs_diff"] = float(np.max(np.abs(t.float().cpu().numpy() - m1.encode(sents[:5], batch_size=4, max_length=64))))
        ok &= out["bit_identical_to_one_engine"] and out["odd_split_max_abs_diff"] == 0.0 and t.shape == (5, 256)
    return _res("GritLM in-process multi-GPU encode (two replicas, DataParallel row split)", ok, **out)


def check_gritlm_api_variants():
    """The drop-in surface beyond plain mean pooling (gritlm/gritlm.py:92-176, :178-218): every pooling method, `recast`, `embed_eos`,
    `add_special_tokens=False`, `max_length` truncation under an instruction, `convert_to_tensor`, a single string, `encode_corpus` on
    title / text dicts, `encode_queries` -- the native engine against the SAME wrapper on its Hugging Face path (`native=False`; that
    path is the reference's code shape and is pinned on the reference's own outputs on the CPU, tests/test_gritlm_cpu.py).  Output
    dtype / shape / device must agree exactly, values to 1 - cos < 1e-4 (2-layer model: both bf16 paths are within 2e-5 of fp32)."""
    import tempfile
    from gritlm_amd import GritLM
    sents = synth.make_sentences(13, seed=21, min_words=1, max_words=70)
    instr = "Given a query, retrieve passages: "
    det, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        d32 = synth.build_mistral_dir(os.path.join(td, "m32"), "tiny", 0, "float32")
        nat = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16)
        hf = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, native=False)
        f32m = GritLM(d32, pooling_method="mean", attn="bbcc", device="cuda", native=False)          # the same wrapper in fp32: the truth
        ok &= nat.engine is not None and hf.engine is None and f32m.engine is None
        eos = nat.tokenizer.eos_token or ""
        cosd = lambda x, y: float(np.max(1 - np.sum(x * y, axis=-1) / (np.linalg.norm(x, axis=-1) * np.linalg.norm(y, axis=-1))))

        def cmp(tag, **kw):
            """Averaging poolings: the two bf16 paths within 1e-4 of each other.  Single-token poolings ('cls', 'lasttoken'; also everything
            that is rounded to bf16 before the normalisation) carry ONE token's bf16 noise un-averaged -- the BOS position of this model is
            4e-3 from fp32 in the Hugging Face module's own bf16 run -- so there the native path is held to the fp32 run of the wrapper: no
            further from it than 1.5x the Hugging Face bf16 path is (+ 1e-5)."""
            nonlocal ok
            for m_ in (nat, hf, f32m):
                m_.pooling_method, m_.normalized, m_.embed_eos = nat.pooling_method, nat.normalized, nat.embed_eos
            a, b, c = (m_.encode(sents, batch_size=5, **kw) for m_ in (nat, hf, f32m))
            same_kind = type(a) is type(b) and a.shape == b.shape and a.dtype == b.dtype
            af, bf_, cf = (x.float().cpu().numpy() if torch.is_tensor(x) else x for x in (a, b, c))
            d, d_nat, d_hf = cosd(af, bf_), cosd(af, cf), cosd(bf_, cf)
            det[tag] = d
            single = nat.pooling_method in ("cls", "lasttoken") or kw.get("recast")
            if single:
                det[tag + "_vs_fp32"], det[tag + "_hf_vs_fp32"] = d_nat, d_hf
            ok &= bool(same_kind) and bool(np.isfinite(af).all()) and ((d_nat <= 1.5 * d_hf + 1e-5) if single else d < 1e-4)

        for method in ("mean", "weightedmean", "lasttoken", "cls"):
            nat.pooling_method = method
            cmp(f"{method}", max_length=64)
            cmp(f"{method}_instr", max_length=64, instruction=instr)
        nat.pooling_method = "mean"
        cmp("recast_tensor", max_length=64, recast=True, convert_to_tensor=True)
        cmp("truncated_under_instruction", max_length=12, instruction=instr)
        cmp("no_special_tokens", max_length=64, add_special_tokens=False)
        cmp("embed_instruction", max_length=64, instruction=instr, embed_instruction=True)
        if eos and eos in nat.tokenizer.vocab:
            nat.embed_eos = eos
            cmp("embed_eos", max_length=64)
            nat.embed_eos = ""
        nat.normalized = False
        cmp("not_normalized", max_length=64)
        nat.normalized = True
        for m_ in (hf, f32m):
            m_.pooling_method, m_.normalized, m_.embed_eos = nat.pooling_method, nat.normalized, nat.embed_eos
        one_n, one_h = nat.encode(sents[3], max_length=64), hf.encode(sents[3], max_length=64)
        ok &= one_n.shape == one_h.shape == (256,) and float(1 - np.sum(one_n * one_h)) < 1e-4
        docs = [{"title": "T " + s[:10], "text": s} if i % 2 else {"text": s} for i, s in enumerate(sents)]
        cn, ch = nat.encode_corpus(docs, batch_size=4, max_length=64), hf.encode_corpus(docs, batch_size=4, max_length=64)
        det["corpus_dicts"] = float(np.max(1 - np.sum(cn * ch, axis=1)))
        qn, qh = nat.encode_queries(sents[:4], max_length=64, instruction=instr), hf.encode_queries(sents[:4], max_length=64, instruction=instr)
        det["queries"] = float(np.max(1 - np.sum(qn * qh, axis=1)))
        ok &= det["corpus_dicts"] < 1e-4 and det["queries"] < 1e-4
    return _res("GritLM API variants: native engine vs the wrapper's Hugging Face path", ok, **det)


def check_gritlm_native_mixtral():
    """gritlm_amd.GritLM on a (tiny) Mixtral checkpoint directory: model_type 'mixtral' binds the native MoE engine (from the
    installed transformers' fused expert parameters) and encode() matches the oracle on the same tokens."""
    import tempfile
    from gritlm_amd import GritLM
    sents = synth.make_sentences(10, seed=3)
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mixtral_dir(os.path.join(td, "x16"), "moe-tiny", 0, "bfloat16")
        m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16)
        ok &= m.engine is not None and m.engine.cfg.num_local_experts == 8
        e = m.encode(sents, batch_size=4, max_length=64)
        tok = m.tokenizer(sents, padding=True, truncation=True, return_tensors="np", max_length=64, add_special_tokens=True)
        ids, mask = tok["input_ids"].astype(np.int64), tok["attention_mask"].astype(np.int64)
    cfg = synth.CONFIGS["moe-tiny"]
    w = synth.make_weights(cfg, 0)
    ref = O.encode_core(w, cfg, ids, mask, "mean", True)
    refb = O.l2_normalize(O.pooling(O.mistral_encode(w, cfg, ids, mask, emulate_bf16=True), mask, "mean"))
    out["1-cos_vs_fp32_oracle"] = float(np.max(1 - np.sum(e * ref, axis=1)))
    out["1-cos_of_bf16_oracle"] = float(np.max(1 - np.sum(refb * ref, axis=1)))
    ok &= e.shape == (10, 256) and out["1-cos_vs_fp32_oracle"] < max(1e-4, 2 * out["1-cos_of_bf16_oracle"])
    return _res("GritLM.encode native on a Mixtral directory vs oracle", bool(ok), **out)


# Training pins.  The fixtures carry the reference's fp32 run AND its own bf16 run (model.to(bfloat16), CPU).  A bf16
# implementation is held to "no further from fp32 than 1.25x the reference's own bf16 run" per parameter (+ GRAD_FLOOR for quantities
# whose reference error is ~0: gradient NORMS of the bf16 reference agree with fp32 to 1e-5..6e-4 because its rounding noise averages
# out of a norm), and the loss to LOSS_VS_F32_REF of the reference's fp32 loss (at the tiny model, whose loss is 15.17 and whose
# reference bf16 run is itself 8.3e-3 away, to 1.25x that run + LOSS_VS_F32_REF).
GRAD_FLOOR = 2e-3
LOSS_VS_F32_REF = 1e-3

_NAMES = ["layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight", "norm.weight", "layers.0.input_layernorm.weight",
          "layers.1.self_attn.v_proj.weight", "embed_tokens.weight"]


def check_train_step(mode="direct"):
    """Native contrastive step (HIP forward + backward + InfoNCE kernel) vs the REFERENCE's loss and parameter gradients
    (tests/golden/gradcache_tiny.npz: GritLMTrainModel.forward + backward, and the vendored GradCache, fp32 on CPU)."""
    import tempfile
    from gritlm_amd.training import GradCacheStep, GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        q = {"input_ids": torch.from_numpy(g["q_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["q_mask"]).to(DEV)}
        p = {"input_ids": torch.from_numpy(g["p_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["p_mask"]).to(DEV)}
        if mode == "direct":
            o = m(query=q, passage=p)
            loss = o.loss
            loss.backward()
            qr = f32(o.q_reps)
            out["q_reps_1-cos"] = float(np.max(1 - np.sum(qr * g["q_reps"], axis=1)))
            ok &= out["q_reps_1-cos"] < 1e-4
        else:
            # GradCache's pass 1 is a no-grad forward: it must not keep activations (round 3 found it saving all of them: inside an
            # autograd.Function grad mode is always off and ctx.needs_input_grad is True under no_grad too)
            saves = []
            orig_fp = m.train_engine.forward_pooled
            def spy(*a, **k):
                saves.append(bool(k.get("save", False)))
                return orig_fp(*a, **k)
            m.train_engine.forward_pooled = spy
            loss = GradCacheStep(m, chunk_size=2)(q, p)
            m.train_engine.forward_pooled = orig_fp
            n_chunks = (g["q_ids"].shape[0] + 1) // 2 + (g["p_ids"].shape[0] + 1) // 2
            # pass 1 runs several chunks per call (gradcache.pass1_chunk_rows: default 4 x the chunk)
            from gritlm_amd.training.gradcache import pass1_chunk_rows
            p1 = pass1_chunk_rows(m, 2)
            n_calls1 = -(-g["q_ids"].shape[0] // p1) + -(-g["p_ids"].shape[0] // p1)
            out["pass1_calls"] = n_calls1
            out["pass1_saves"] = sum(saves[:n_calls1]); out["pass2_saves"] = sum(saves[n_calls1:])
            ok &= len(saves) == n_calls1 + n_chunks and out["pass1_saves"] == 0 and out["pass2_saves"] == n_chunks
        key = "direct" if mode == "direct" else "gradcache"
        ref_loss, ref_loss16 = float(g[f"loss_{key}"]), float(g[f"loss_{key}_bf16"])
        out["loss"] = float(loss.item()); out["loss_ref"] = ref_loss; out["loss_ref_bf16"] = ref_loss16
        # the InfoNCE kernel itself holds 1e-3 ABSOLUTE on identical fp32 reps (check_infonce); around a bf16 encoder the yardstick is the
        # reference's own bf16 run of the step (15.1819 vs 15.1736 in fp32)
        ok &= abs(out["loss"] - ref_loss) <= 1.25 * _yard()[f"gradcache_tiny/loss_gap_bf16_{key}"] + LOSS_VS_F32_REF
        out["loss_minus_f32ref"] = out["loss"] - ref_loss; out["loss_minus_bf16ref"] = out["loss"] - ref_loss16
        sd = dict(m._backbone().named_parameters())
        worst = worst_ratio = 0.0
        for n in _NAMES:
            ref, ref16 = g[f"grad_{key}/" + n], g[f"grad_{key}_bf16/" + n]
            got = f32(sd[n].grad)
            rel = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-20))
            rel16 = _yard()[f"gradcache_tiny/grad_rel_bf16_{key}/{n}"]
            out[n.replace("layers.", "L").replace(".weight", "")] = rel
            worst = max(worst, rel)
            worst_ratio = max(worst_ratio, rel / (1.25 * rel16 + GRAD_FLOOR))
        out["grad_err_over_bound"] = worst_ratio                 # bound per parameter: 1.25 x the reference's own bf16 error + floor
        ok &= worst_ratio <= 1.0
        if mode == "gradcache":
            # transposed-weight cache follows in-place parameter updates without an explicit weights_updated()
            eng = m.train_engine
            eng.cache_transposed_weights = True
            L0 = eng.layers[0]
            t1 = eng._wt(0, "qkv", L0.wqkv); t1b = eng._wt(0, "qkv", L0.wqkv)
            with torch.no_grad():
                L0.mods[0].k_proj.weight.add_(1.0)
            t2 = eng._wt(0, "qkv", L0.wqkv)
            ok &= (t1 is t1b) and (t2 is not t1) and bool(torch.equal(t2, L0.wqkv.t().contiguous()))
        # state_dict keeps the reference names although q/k/v and gate/up live in packed storage
        ok &= all(k in m.model.state_dict() for k in ("layers.0.self_attn.k_proj.weight", "layers.1.mlp.up_proj.weight"))
    return _res(f"native train step [{mode}] vs reference loss+grads", ok, **out)


def check_train_causal_embedding(pooling="weightedmean"):
    """'cc' embedding attention (the reference's attn='cccc' / 'cc': the stock causal forward, gritlm/training/model.py:146-148 without
    ``is_causal=False``) on the native training engine -- round 6 -- against the SAME model class on its Hugging Face path in fp32 (the
    reference's own steps on the stock module, autograd): representations, loss, every parameter's gradient, for the direct step and for
    GradCache with a bf16 and an fp16 pass 1 (the fp16 one must land closer to the fp32 loss)."""
    import tempfile
    from gritlm_amd.training import GradCacheStep, GritLMTrainModel
    cfg = synth.CONFIGS["tiny"]
    qi, qm = synth.make_batch(cfg, 4, 24, seed=71, min_len=6)
    pi, pm = synth.make_batch(cfg, 8, 40, seed=72, min_len=9)
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        d32 = synth.build_mistral_dir(os.path.join(td, "m32"), "tiny", 0, "float32")
        mk = lambda d, dt: GritLMTrainModel(model_name_or_path=d, mode="embedding", pooling_method=pooling, normalized=True, attn="cccc",
                                            temperature=0.02, negatives_cross_device=False, device="cuda", torch_dtype=dt)
        feats = lambda: ({"input_ids": torch.from_numpy(qi).to(DEV), "attention_mask": torch.from_numpy(qm).to(DEV)},
                         {"input_ids": torch.from_numpy(pi).to(DEV), "attention_mask": torch.from_numpy(pm).to(DEV)})
        ref = mk(d32, torch.float32)
        ref.model.to(DEV)
        q, p = feats()
        o = ref(query=q, passage=p)
        o.loss.backward()
        ref_g = {n: f32(t.grad) for n, t in ref._backbone().named_parameters()}
        ref_loss, ref_q, ref_p = float(o.loss.detach()), f32(o.q_reps), f32(o.p_reps)
        # and the bidirectional forward of the same weights must NOT match: the check would pass vacuously if 'cc' were ignored
        m = mk(d16, torch.bfloat16)
        m.enable_native()
        q, p = feats()
        o2 = m(query=q, passage=p)
        o2.loss.backward()
        omc = lambda a, b: float(np.max(1 - np.sum(a * b, axis=1)))
        out["q_reps_1-cos"], out["p_reps_1-cos"] = omc(f32(o2.q_reps), ref_q), omc(f32(o2.p_reps), ref_p)
        out["loss_ref_fp32"], out["loss_direct"] = ref_loss, float(o2.loss.detach())
        sd = dict(m._backbone().named_parameters())

        def grads_vs_ref(tag):
            worst_rel, worst_cos = 0.0, 1.0
            for n, g0 in ref_g.items():
                got = f32(sd[n].grad)
                worst_rel = max(worst_rel, float(np.linalg.norm(got - g0) / (np.linalg.norm(g0) + 1e-20)))
                worst_cos = min(worst_cos, float(np.sum(got * g0) / (np.linalg.norm(got) * np.linalg.norm(g0) + 1e-20)))
            out[f"grad_rel_l2_worst[{tag}]"], out[f"grad_cos_min[{tag}]"] = worst_rel, worst_cos
            return worst_rel < 0.04 and worst_cos > 0.999            # (measured 1.4e-2 / 0.9999)
        ok &= out["q_reps_1-cos"] < 1e-4 and out["p_reps_1-cos"] < 1e-4 and abs(out["loss_direct"] - ref_loss) < 2e-2 and grads_vs_ref("direct")
        m.attn = "bbcc"
        with torch.no_grad():
            q, p = feats()
            out["bidirectional_q_reps_1-cos"] = omc(f32(m.encode(q)), ref_q)
        m.attn = "cccc"
        ok &= out["bidirectional_q_reps_1-cos"] > 20 * out["q_reps_1-cos"]
        for pol in ("bf16", "f16_operands"):
            for t in sd.values():
                t.grad = None
            m.train_engine.prepare_grads() if hasattr(m.train_engine, "prepare_grads") else None
            q, p = feats()
            step = GradCacheStep(m, chunk_size=2, precision=pol)
            loss = float(step(q, p))
            out[f"loss_gradcache[{pol}]"] = loss
            ok &= grads_vs_ref(f"gradcache_{pol}") and abs(loss - ref_loss) < 2e-2
            if pol != "bf16":
                out["reps_1-cos_f16_pass1"] = max(omc(f32(step.last_reps[0]), ref_q), omc(f32(step.last_reps[1]), ref_p)) if getattr(step, "last_reps", None) else -1.0
        ok &= abs(out["loss_gradcache[f16_operands]"] - ref_loss) <= abs(out["loss_gradcache[bf16]"] - ref_loss) + 1e-4
        ok &= 0 <= out["reps_1-cos_f16_pass1"] < 2e-5
    return _res(f"'cc' embedding attention on the native training engine [{pooling}] vs the stock module in fp32", bool(ok), **out)


def check_train_step_7b_layer():
    """One contrastive step at the TRUE 7B layer shape (H 4096, 32/8 heads, I 14336, one layer; every dgrad/wgrad GEMM at the bench's
    N and K) vs the reference's direct forward + backward in fp32 (tests/golden/train_7b-l1.npz, from GritLMTrainModel.forward).
    Three native schedules against the same reference: direct, GradCache (chunk 2), GradCache with layer recompute.  The fixture also
    holds the reference's OWN bf16 run of the step: loss within LOSS_VS_F32_REF = 1e-3 of the reference's fp32 loss (the distance to the
    reference's bf16 loss -- itself 6.3e-3 off -- is reported); reps 1-cos < 1e-4; every parameter's gradient probe and gradient norm no further from fp32 than 1.25x the
    reference's bf16 run of that parameter (+ GRAD_FLOOR)."""
    import tempfile
    from gritlm_amd.training import GradCacheStep, GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "train_7b-l1.npz"))
    out, ok = {}, True
    q = {"input_ids": torch.from_numpy(g["q_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["q_mask"]).to(DEV)}
    p = {"input_ids": torch.from_numpy(g["p_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["p_mask"]).to(DEV)}
    ref_loss, ref_loss16 = float(g["loss"]), float(g["loss_bf16"])
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "7b-l1", 0, "bfloat16")
        for sched in ("direct", "gradcache", "gradcache+recompute"):
            m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                                 temperature=float(g["tau"]), negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
            m.enable_native()
            if sched == "direct":
                o = m(query=dict(q), passage=dict(p))
                loss = o.loss
                loss.backward()
                for nm, r in (("q", o.q_reps), ("p", o.p_reps)):
                    c = float(np.max(1 - np.sum(f32(r) * g[nm + "_reps"], axis=1)))
                    out[f"{nm}_reps_1-cos"] = c
                    ok &= c < 1e-4
            else:
                if sched.endswith("recompute"):
                    m.gradient_checkpointing_enable()
                loss = GradCacheStep(m, chunk_size=2)(dict(q), dict(p))
            lv = float(loss.item())
            out[f"loss[{sched}]"] = lv
            # the north-star's bound, against the reference's fp32 loss: |d loss| < 1e-3 (measured 4.4e-4; the reference's OWN bf16 run of
            # this step is 6.3e-3 away from its fp32 loss, so "within 1e-3 of the bf16-reference loss" would reward the larger error:
            # the distance to it is reported, not asserted)
            ok &= abs(lv - ref_loss) < LOSS_VS_F32_REF
            out[f"loss_minus_f32ref[{sched}]"] = lv - ref_loss
            out[f"loss_minus_bf16ref[{sched}]"] = lv - ref_loss16
            worst_probe, worst_norm, worst_ratio, worst_nratio = 0.0, 0.0, 0.0, 0.0
            for n, t in m._backbone().named_parameters():
                got = t.grad
                ref_n, ref_n16 = float(g["gnorm/" + n]), float(g["gnorm_bf16/" + n])
                en = abs(float(got.double().norm().item()) - ref_n) / (ref_n + 1e-20)
                worst_norm = max(worst_norm, en)
                worst_nratio = max(worst_nratio, en / (1.25 * _yard()[f"train_7b-l1/gnorm_rel_bf16/{n}"] + GRAD_FLOOR))
                ref, ref16 = g["probe/" + n], g["probe_bf16/" + n]
                if n == "embed_tokens.weight":
                    gp = f32(got[torch.from_numpy(g["probe_rows/" + n]).to(DEV)])
                elif got.dim() == 2:
                    gp = f32(got[:8])
                else:
                    gp = f32(got)
                e = float(np.linalg.norm(gp - ref) / (np.linalg.norm(ref) + 1e-20))
                e16 = _yard()[f"train_7b-l1/probe_rel_bf16/{n}"]
                worst_probe = max(worst_probe, e)
                worst_ratio = max(worst_ratio, e / (1.25 * e16 + GRAD_FLOOR))
            out[f"grad_probe_rel[{sched}]"] = worst_probe
            out[f"grad_norm_rel[{sched}]"] = worst_norm
            out[f"probe_err_over_bound[{sched}]"] = worst_ratio          # bound per parameter: 1.25 x the reference's own bf16 error + floor
            out[f"norm_err_over_bound[{sched}]"] = worst_nratio
            ok &= worst_ratio <= 1.0 and worst_nratio <= 1.0
            del m
            torch.cuda.empty_cache()
    out["loss_ref"] = ref_loss
    out["loss_ref_bf16"] = ref_loss16
    out["ref_bf16_worst_probe_rel"] = float(g["ref_bf16_vs_f32_worst_probe_rel_l2"])
    return _res("native train step [7b-l1: direct / gradcache / recompute] vs reference loss+grads (fp32 and its own bf16 run)", bool(ok), **out)


def check_grouped_training_epilogues(E=4, H=256, I=512, counts=(300, 0, 129, 71)):
    """grit_gemm_bf16_nt_grouped_epi (the expert MLP of Mixtral training) against the dense kernel group by group, bit for bit:
    SWIGLU_STACKED_SAVE with the token gather (a_rows), STORE, SWIGLU_BWD; an empty group in the middle; + moe_combine_bwd vs torch."""
    from gritlm_amd._lib import EPI_SWIGLU_BWD, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE
    M = int(sum(counts))
    T = M // 2
    rng = np.random.default_rng(77)
    x = bf(rnd((T, H), 1))
    a_rows = torch.from_numpy(rng.integers(0, T, M).astype(np.int32)).to(DEV)
    wgu, wdn = bf(rnd((E, 2 * I, H), 2, 0.05)), bf(rnd((E, H, I), 3, 0.05))
    cnt = torch.tensor(counts, dtype=torch.int32, device=DEV)
    gu = torch.zeros((M, 2 * I), dtype=torch.bfloat16, device=DEV)
    act = ops.gemm_nt_grouped_epi(x, wgu, cnt, M, EPI_SWIGLU_STACKED_SAVE, residual=gu, a_rows=a_rows)
    act2 = ops.gemm_nt_grouped_epi(x, wgu, cnt, M, EPI_SWIGLU_STACKED, a_rows=a_rows)
    y = ops.gemm_nt_grouped_epi(act, wdn, cnt, M, EPI_STORE)
    dy = bf(rnd((M, H), 4))
    wdnT = torch.stack([ops.transpose(wdn[e]) for e in range(E)])               # [E, I, H]
    dgu = ops.gemm_nt_grouped_epi(dy, wdnT, cnt, M, EPI_SWIGLU_BWD, residual=gu)
    ok, off = bool(torch.equal(act, act2)), 0
    xg = x.index_select(0, a_rows.to(torch.int64))
    for e, n in enumerate(counts):
        if n == 0:
            continue
        seg = slice(off, off + n)
        gu_e = torch.zeros((n, 2 * I), dtype=torch.bfloat16, device=DEV)
        act_e = ops.gemm_nt(xg[seg].contiguous(), wgu[e], epilogue=EPI_SWIGLU_STACKED_SAVE, residual=gu_e)
        ok &= bool(torch.equal(act_e, act[seg])) and bool(torch.equal(gu_e, gu[seg]))
        ok &= bool(torch.equal(ops.gemm_nt(act[seg].contiguous(), wdn[e]), y[seg]))
        ok &= bool(torch.equal(ops.gemm_nt(dy[seg].contiguous(), wdnT[e], epilogue=EPI_SWIGLU_BWD, residual=gu[seg].contiguous()), dgu[seg]))
        off += n
    # combine backward vs torch: routed rows of token t are rows[t,0], rows[t,1]
    Tt = 37
    perm = torch.from_numpy(rng.permutation(2 * Tt).astype(np.int32)).to(DEV)
    rows = perm.view(Tt, 2).contiguous()
    row_token = torch.empty((2 * Tt,), dtype=torch.int32, device=DEV)
    row_token[rows.view(-1).to(torch.int64)] = torch.arange(Tt, device=DEV, dtype=torch.int32).repeat_interleave(2)
    wts = torch.from_numpy(rng.random((Tt, 2), dtype=np.float32)).to(DEV)
    dout, yy = bf(rnd((Tt, H), 5)), bf(rnd((2 * Tt, H), 6))
    dyy, dw = ops.moe_combine_bwd(dout, yy, row_token, rows, wts)
    w_sorted = torch.empty((2 * Tt,), dtype=torch.float32, device=DEV)
    w_sorted[rows.view(-1).to(torch.int64)] = wts.view(-1)
    dg = dout.index_select(0, row_token.to(torch.int64)).float()
    dy_ref = (dg * w_sorted[:, None]).to(torch.bfloat16)
    dw_ref = (dg * yy.float()).sum(-1)[rows.view(-1).to(torch.int64)].view(Tt, 2)
    ok &= bool(torch.equal(dyy, dy_ref))
    e_dw = float((dw - dw_ref).abs().max() / (dw_ref.abs().max() + 1e-9))
    ok &= e_dw < 1e-5
    return _res("grouped training epilogues == dense per group; combine backward vs torch", ok, dw_rel=e_dw)


def check_train_step_mixtral(mode="direct"):
    """Native contrastive step on the bidirectional MIXTRAL (sparse-MoE MLP forward with saved pre-activations, MoE backward: combine
    backward, grouped dgrads with the SwiGLU backward in the epilogue, per-expert weight gradients, router backward) vs the
    REFERENCE's loss, representations and parameter gradients (tests/golden/train_moe-tiny.npz: GritLMTrainModel.forward + backward
    around scripts/modeling_mixtral_gritlm.py, fp32 and bf16 on CPU).  Gradients: relative l2 vs the fp32 run below 6e-2 (the
    reference's own bf16 run is 2.8e-2 away); every parameter's gradient norm within 5e-2."""
    import tempfile
    from gritlm_amd.training import GradCacheStep, GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "train_moe-tiny.npz"))
    I = synth.CONFIGS["moe-tiny"]["intermediate_size"]
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mixtral_dir(os.path.join(td, "m16"), "moe-tiny", 0, "bfloat16")
        m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=float(g["tau"]), negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        from gritlm_amd.training.engine import MixtralTrainEngine
        ok &= isinstance(m.train_engine, MixtralTrainEngine)
        q = {"input_ids": torch.from_numpy(g["q_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["q_mask"]).to(DEV)}
        p = {"input_ids": torch.from_numpy(g["p_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["p_mask"]).to(DEV)}
        if mode == "direct":
            o = m(query=q, passage=p)
            loss = o.loss
            loss.backward()
            for nm, r in (("q", o.q_reps), ("p", o.p_reps)):
                c = float(np.max(1 - np.sum(f32(r) * g[nm + "_reps_f32"], axis=1)))
                out[f"{nm}_reps_1-cos"] = c
                ok &= c < 2e-4          # bf16 routing flips a few near-tied tokens (the reference's own bf16 run: see the fixture)
            out["ref_bf16_q_1-cos"] = float(np.max(1 - np.sum(g["q_reps_bf16"] * g["q_reps_f32"], axis=1)))
        else:
            if mode == "recompute":
                m.gradient_checkpointing_enable()
            loss = GradCacheStep(m, chunk_size=2)(q, p)
        lv = float(loss.item())
        out["loss"], out["loss_ref_f32"], out["loss_ref_bf16"] = lv, float(g["loss_f32"]), float(g["loss_bf16"])
        ok &= abs(lv - float(g["loss_f32"])) < 2e-3 * max(1.0, abs(float(g["loss_f32"])))
        sd = dict(m._backbone().named_parameters())

        def ours(ref_name):
            """gradient of the reference-named parameter out of the fused transformers >= 5 parameters"""
            if "block_sparse_moe" not in ref_name:
                return f32(sd[ref_name].grad)
            pre, rest = ref_name.split(".block_sparse_moe.")
            if rest == "gate.weight":
                return f32(sd[pre + ".mlp.gate.weight"].grad)
            _, e, w, _ = rest.split(".")            # experts.E.w1.weight
            e = int(e)
            if w == "w2":
                return f32(sd[pre + ".mlp.experts.down_proj"].grad[e])
            gu = sd[pre + ".mlp.experts.gate_up_proj"].grad[e]
            return f32(gu[:I] if w == "w1" else gu[I:])

        worst, worst_name, worst_norm = 0.0, "", 0.0
        for k in g.files:
            if k.startswith("grad_f32/"):
                n = k[len("grad_f32/"):]
                ref = g[k]
                rel = float(np.linalg.norm(ours(n) - ref) / (np.linalg.norm(ref) + 1e-20))
                if rel > worst:
                    worst, worst_name = rel, n
            elif k.startswith("gnorm_f32/"):
                n = k[len("gnorm_f32/"):]
                rn = float(g[k])
                if rn > 1e-6:
                    worst_norm = max(worst_norm, abs(float(np.linalg.norm(ours(n))) - rn) / rn)
        out["worst_grad_rel_l2"], out["worst_grad"], out["worst_norm_rel"] = worst, worst_name.replace("block_sparse_moe", "moe"), worst_norm
        out["ref_bf16_vs_f32"] = float(g["ref_bf16_vs_f32_worst_rel_l2"])
        ok &= worst < 6e-2 and worst_norm < 5e-2
    return _res(f"native Mixtral train step [{mode}] vs reference loss+grads", bool(ok), **out)


def check_generative_mixtral():
    """Generative branch on a Mixtral (the reference takes the model's own loss there: token-sum cross entropy / batch * factor +
    router_aux_loss_coef * load_balancing_loss_func) vs the reference's MixtralForCausalLM.forward + backward
    (tests/golden/generative_moe-tiny.npz, fp32 on CPU, aux coefficient 0.5 so that the router gradients are dominated by the
    auxiliary term): loss, and every stored gradient incl. the routers' and lm_head's."""
    import tempfile
    from gritlm_amd.training import GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "generative_moe-tiny.npz"))
    I = synth.CONFIGS["moe-tiny"]["intermediate_size"]
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mixtral_dir(os.path.join(td, "mixtral-tiny"), "moe-tiny", 0, "bfloat16")     # "mixtral" in the path: model's own loss
        m = GritLMTrainModel(model_name_or_path=d16, mode="unified", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=False, loss_gen_type="token", loss_gen_factor=float(g["factor"]),
                             device="cuda", torch_dtype=torch.bfloat16)
        ok &= m.gen_loss_fn is None
        m.enable_native()
        mk_gen = lambda: {"input_ids": torch.from_numpy(g["input_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["attention_mask"]).to(DEV),
                          "labels": torch.from_numpy(g["labels"]).to(DEV)}
        # (1) without the auxiliary term
        m.model.config.router_aux_loss_coef = 0.0
        o0 = m(generative=mk_gen())
        o0.loss.backward()
        sd0 = dict(m.model.named_parameters())
        worst0 = 0.0
        for k in g.files:
            if k.startswith("grad_noaux/"):
                n = k[len("grad_noaux/"):]
                nn = n.replace("block_sparse_moe.gate.weight", "mlp.gate.weight")
                got = f32(sd0[nn].grad) if "experts" not in n else f32(sd0[n.split(".block_sparse_moe.")[0] + ".mlp.experts.gate_up_proj"].grad[0][:I])
                rel = float(np.linalg.norm(got - g[k]) / (np.linalg.norm(g[k]) + 1e-20))
                if os.environ.get("GRIT_CHECK_VERBOSE"):
                    print(f"   [no aux] {n:60s} rel {rel:.3e}")
                worst0 = max(worst0, rel)
        out["loss_noaux"], out["loss_noaux_ref"], out["worst_grad_noaux"] = float(o0.loss_gen.item()), float(g["loss_noaux"]), worst0
        ok &= worst0 < 6e-2
        m.model.zero_grad(set_to_none=True)
        # (2) with it
        m.model.config.router_aux_loss_coef = float(g["router_aux_loss_coef"])
        if "routing" in g.files:          # routing agreement with the reference's fp32 run (real tokens, either order of the two experts)
            eng = m.train_engine
            eng._router_log = []
            with torch.no_grad():
                eng.forward(mk_gen()["input_ids"], mk_gen()["attention_mask"], save=False, packed=True, causal=True)
            log, eng._router_log = eng._router_log, None
            keep = g["attention_mask"].reshape(-1) != 0
            agree = []
            for li, (_, ex) in enumerate(log):
                ref = np.sort(g["routing"][li][keep], axis=-1)
                agree.append(float((np.sort(ex.cpu().numpy(), axis=-1) == ref).all(-1).mean()))
            out["routing_agreement_per_layer"] = str([round(a, 4) for a in agree])
        o = m(generative=mk_gen())
        o.loss.backward()
        lv = float(o.loss_gen.item())
        out["loss"], out["loss_ref"], out["aux_ref"] = lv, float(g["loss"]), float(g["aux_loss"])
        ok &= abs(lv - float(g["loss"])) < 2e-3 * abs(float(g["loss"]))
        sd = dict(m.model.named_parameters())

        def ours(ref_name):
            if "block_sparse_moe" not in ref_name:
                return f32(sd[ref_name].grad)
            pre, rest = ref_name.split(".block_sparse_moe.")
            if rest == "gate.weight":
                return f32(sd[pre + ".mlp.gate.weight"].grad)
            _, e, w, _ = rest.split(".")
            e = int(e)
            if w == "w2":
                return f32(sd[pre + ".mlp.experts.down_proj"].grad[e])
            gu = sd[pre + ".mlp.experts.gate_up_proj"].grad[e]
            return f32(gu[:I] if w == "w1" else gu[I:])

        worst, worst_name, worst_gate = 0.0, "", 0.0
        for k in g.files:
            if k.startswith("grad/"):
                n = k[len("grad/"):]
                ref = g[k]
                rel = float(np.linalg.norm(ours(n) - ref) / (np.linalg.norm(ref) + 1e-20))
                if "gate.weight" in n:
                    worst_gate = max(worst_gate, rel)
                if os.environ.get("GRIT_CHECK_VERBOSE"):
                    print(f"   {n:60s} rel {rel:.3e}  |ref| {np.linalg.norm(ref):.3e} |ours| {np.linalg.norm(ours(n)):.3e}")
                if rel > worst:
                    worst, worst_name = rel, n
        out["worst_grad_rel_l2"], out["worst_grad"], out["worst_router_grad_rel_l2"] = worst, worst_name.replace("block_sparse_moe", "moe"), worst_gate
        ok &= worst < 6e-2
    return _res("native Mixtral generative loss (CE + router auxiliary loss) vs reference loss+grads", bool(ok), **out)


def check_train_packed_vs_padded(cfg_name="gqa"):
    """One contrastive step with the packed (un-padded) training path vs the padded one: identical reps and loss, parameter
    gradients equal up to the bf16 accumulation order of the wgrad GEMMs (K = tokens, padded rows contribute exact zeros)."""
    from gritlm_amd.training.engine import MistralTrainEngine, SyntheticBackbone
    from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel
    cfg = EncoderConfig.from_dict(synth.CONFIGS[cfg_name])
    idq, mq = synth.make_batch(synth.CONFIGS[cfg_name], 4, 48, seed=5, min_len=9)
    idp, mp_ = synth.make_batch(synth.CONFIGS[cfg_name], 8, 150, seed=6, min_len=20)
    q = {"input_ids": torch.from_numpy(idq).to(DEV), "attention_mask": torch.from_numpy(mq).to(DEV), "instruction_lens": [3, 0, 5, 2]}
    p = {"input_ids": torch.from_numpy(idp).to(DEV), "attention_mask": torch.from_numpy(mp_).to(DEV)}
    res = {}
    for packed in (False, True):
        bb = SyntheticBackbone(cfg, DEV, seed=3)
        m = GritLMTrainModel.__new__(GritLMTrainModel)
        torch.nn.Module.__init__(m)
        m.model, m.projection, m.pooling_method, m.normalized, m.attn, m.embedding_attr = bb, None, "mean", True, "bbcc", None
        m.emb_loss_fn = DistributedContrastiveLoss(0.02, False)
        m.train_engine = MistralTrainEngine(bb, cfg, DEV)
        m.native_packed = packed
        o = m(query=dict(q), passage=dict(p))
        o.loss.backward()
        res[packed] = (float(o.loss.item()), f32(o.q_reps), f32(o.p_reps),
                       {n: f32(t.grad) for n, t in bb.named_parameters()}, m.train_engine)
    out, ok = {}, True
    ok &= res[True][0] == res[False][0] and np.array_equal(res[True][1], res[False][1]) and np.array_equal(res[True][2], res[False][2])
    out["loss"] = res[True][0]
    worst = 0.0
    for n, g_pad in res[False][3].items():
        g_pk = res[True][3][n]
        rel = float(np.linalg.norm(g_pk - g_pad) / (np.linalg.norm(g_pad) + 1e-20))
        worst = max(worst, rel)
    out["worst_grad_rel_l2"] = worst
    ok &= worst < 1e-2
    # the embedding gradient has no token-count contraction (row sums in token order, no atomics since round 3): padded rows add exact
    # zeros, so both layouts must give the SAME BITS -- and so must a repetition of the step
    emb_same = bool(np.array_equal(res[True][3]["embed_tokens.weight"], res[False][3]["embed_tokens.weight"]))
    out["embedding_grad_identical"] = emb_same
    ok &= emb_same
    ok &= res[True][4]._tbuf and all(k[1] > 0 for k in res[True][4]._tbuf)
    return _res(f"packed training step == padded training step [{cfg_name}]", bool(ok), **out)


def check_gradcache_pass1_superchunks(cfg_name="gqa"):
    """GradCache pass 1 in calls of 4 chunks (gradcache.pass1_chunk_rows) against pass 1 chunk by chunk: the kernels are
    batch-invariant, so the loss and EVERY parameter gradient must come out with the same bits."""
    from gritlm_amd.training import GradCacheStep
    from gritlm_amd.training.engine import MistralTrainEngine, SyntheticBackbone
    from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel
    cfg = EncoderConfig.from_dict(synth.CONFIGS[cfg_name])
    idq, mq = synth.make_batch(synth.CONFIGS[cfg_name], 6, 48, seed=15, min_len=9)
    idp, mp_ = synth.make_batch(synth.CONFIGS[cfg_name], 18, 150, seed=16, min_len=20)
    q = {"input_ids": torch.from_numpy(idq).to(DEV), "attention_mask": torch.from_numpy(mq).to(DEV), "instruction_lens": [3, 0, 5, 2, 1, 0]}
    p = {"input_ids": torch.from_numpy(idp).to(DEV), "attention_mask": torch.from_numpy(mp_).to(DEV)}
    res, calls = {}, {}
    for p1 in (2, 8, 5):                      # chunk by chunk / 4 chunks per call / a size that cuts across the pass-2 chunks
        bb = SyntheticBackbone(cfg, DEV, seed=3)
        m = GritLMTrainModel.__new__(GritLMTrainModel)
        torch.nn.Module.__init__(m)
        m.model, m.projection, m.pooling_method, m.normalized, m.attn, m.embedding_attr = bb, None, "mean", True, "bbcc", None
        m.emb_loss_fn = DistributedContrastiveLoss(0.02, False)
        m.train_engine = MistralTrainEngine(bb, cfg, DEV)
        n = []
        orig = m.train_engine.forward_pooled
        def spy(*a, _o=orig, _n=n, **k):
            _n.append(int(a[0].shape[0]))
            return _o(*a, **k)
        m.train_engine.forward_pooled = spy
        gc = GradCacheStep(m, chunk_size=2, pass1_chunk_size=p1)
        loss = gc(dict(q), dict(p), sync=False)
        res[p1] = (float(loss.item()), {k_: f32(t.grad) for k_, t in bb.named_parameters()})
        calls[p1] = n
    ok = True
    for p1 in (8, 5):
        ok &= res[p1][0] == res[2][0] and all(np.array_equal(res[p1][1][k_], v) for k_, v in res[2][1].items())
    ok &= len(calls[2]) == 2 * 12 and len(calls[8]) == 1 + 3 + 12 and max(calls[8][:4]) == 8 and max(calls[8][4:]) == 2
    default_rows = GradCacheStep(m, chunk_size=2).pass1_chunk_size
    ok &= default_rows == 8
    return _res(f"GradCache pass 1 in super-chunks == chunk by chunk, bit for bit [{cfg_name}]", bool(ok), loss=res[2][0],
                pass1_calls={k_: len(v) - 12 for k_, v in calls.items()})


def check_swiglu_stacked(M=300, I=512, K=256):
    """GRIT_EPI_SWIGLU_STACKED ([gate; up] weights, interleave folded into the LDS-DMA source rows) must be bit-identical to
    GRIT_EPI_SWIGLU on the pre-interleaved copy of the same weights."""
    from gritlm_amd._lib import EPI_SWIGLU_STACKED
    a, wg, wu = bf(rnd((M, K), 3)), bf(rnd((I, K), 4, 0.05)), bf(rnd((I, K), 5, 0.05))
    ref = ops.gemm_nt(a, swiglu_interleave(wg, wu), epilogue=EPI_SWIGLU)
    got = ops.gemm_nt(a, torch.cat([wg, wu], dim=0).contiguous(), epilogue=EPI_SWIGLU_STACKED)
    return _res(f"swiglu stacked == interleaved [M={M},I={I},K={K}]", bool(torch.equal(ref, got)), max_abs=float((ref.float() - got.float()).abs().max()))


def check_swiglu_fused_train_epilogues(M=700, I=1024, K=512):
    """The two training epilogues of the GEMM against the kernels they replace, bit for bit:
    SWIGLU_STACKED_SAVE  == (SWIGLU_STACKED activation, bf16 [gate | up] of the plain GEMM on the stacked weights);
    SWIGLU_BWD           == grit_swiglu_bwd(saved [gate | up], bf16 d_act of the plain GEMM)."""
    from gritlm_amd._lib import EPI_SWIGLU_BWD, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE
    a, wgu = bf(rnd((M, K), 3)), bf(rnd((2 * I, K), 4, 0.05))
    gu_ref = ops.gemm_nt(a, wgu)                                           # [M, 2I] = [gate | up]
    act_ref = ops.gemm_nt(a, wgu, epilogue=EPI_SWIGLU_STACKED)
    gu = torch.full((M, 2 * I), 7.0, dtype=torch.bfloat16, device=DEV)
    act = ops.gemm_nt(a, wgu, epilogue=EPI_SWIGLU_STACKED_SAVE, residual=gu)
    ok1 = bool(torch.equal(act, act_ref)) and bool(torch.equal(gu, gu_ref))
    dh, wdT = bf(rnd((M, K), 8)), bf(rnd((I, K), 9, 0.05))                 # d_act = dh @ wdT^T  (N = I)
    dact = ops.gemm_nt(dh, wdT)
    dgu_ref = ops.swiglu_bwd(gu_ref, dact)
    dgu = ops.gemm_nt(dh, wdT, epilogue=EPI_SWIGLU_BWD, residual=gu_ref)
    ok2 = bool(torch.equal(dgu, dgu_ref))
    return _res(f"fused SwiGLU training epilogues == un-fused kernels [M={M},I={I},K={K}]", ok1 and ok2, save_ok=ok1, bwd_ok=ok2,
                bwd_max_abs=float((dgu.float() - dgu_ref.float()).abs().max()))


def check_train_recompute(cfg_name="gqa"):
    """--gradient_checkpointing on the native engine (keep only the layer inputs, re-run each layer inside backward) vs the default
    keep-everything policy: same reps and loss (bit for bit), parameter gradients equal up to bf16 rounding of the re-computed
    activations (the recompute re-runs the same kernels, so in practice bit-identical)."""
    from gritlm_amd.training.engine import MistralTrainEngine, SyntheticBackbone
    from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel
    cfg = EncoderConfig.from_dict(synth.CONFIGS[cfg_name])
    idq, mq = synth.make_batch(synth.CONFIGS[cfg_name], 4, 48, seed=5, min_len=9)
    idp, mp_ = synth.make_batch(synth.CONFIGS[cfg_name], 8, 150, seed=6, min_len=20)
    q = {"input_ids": torch.from_numpy(idq).to(DEV), "attention_mask": torch.from_numpy(mq).to(DEV)}
    p = {"input_ids": torch.from_numpy(idp).to(DEV), "attention_mask": torch.from_numpy(mp_).to(DEV)}
    res = {}
    for rc in (False, True):
        for packed in (True, False):
            bb = SyntheticBackbone(cfg, DEV, seed=3)
            m = GritLMTrainModel.__new__(GritLMTrainModel)
            torch.nn.Module.__init__(m)
            m.model, m.projection, m.pooling_method, m.normalized, m.attn, m.embedding_attr = bb, None, "mean", True, "bbcc", None
            m.emb_loss_fn = DistributedContrastiveLoss(0.02, False)
            m.train_engine = MistralTrainEngine(bb, cfg, DEV)
            m.native_packed = packed
            if rc:
                m.gradient_checkpointing_enable()
                assert m.train_engine.recompute
            import gc
            gc.collect()                                 # garbage of the previous configuration freed INSIDE the forward would hide the peak
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            o = m(query=dict(q), passage=dict(p))
            peak = torch.cuda.max_memory_allocated() - base
            o.loss.backward()
            res[(rc, packed)] = (float(o.loss.item()), f32(o.q_reps), f32(o.p_reps), {n: f32(t.grad) for n, t in bb.named_parameters()}, peak)
    out, ok, worst = {}, True, 0.0
    for packed in (True, False):
        a, b = res[(False, packed)], res[(True, packed)]
        ok &= abs(a[0] - b[0]) < 1e-3 and float(np.max(1 - np.sum(a[1] * b[1], axis=1))) < 1e-5 and float(np.max(1 - np.sum(a[2] * b[2], axis=1))) < 1e-5
        for n, g0 in a[3].items():
            worst = max(worst, float(np.linalg.norm(b[3][n] - g0) / (np.linalg.norm(g0) + 1e-20)))
        out[f"fwd_activation_bytes_keep_vs_recompute{'_packed' if packed else ''}"] = f"{a[4]}/{b[4]}"
        ok &= b[4] < 0.5 * a[4]
    out["worst_grad_rel_l2"] = worst
    ok &= worst < 1e-2
    return _res(f"recompute (gradient checkpointing) step == keep-all step [{cfg_name}]", bool(ok), **out)


def _rccl_world1_worker(rank, port, model_dir, ret):
    """ONE rank, backend nccl (= RCCL): every collective of the data-parallel GradCache step is issued for real on this GPU."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GRIT_DIST_WORLD1"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        from gritlm_amd.training import gradcache as gcm
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        calls = {"gather": 0, "allreduce": 0}
        orig_g, orig_r = dist.all_gather_into_tensor, dist.all_reduce
        def cg(*a, **k):
            calls["gather"] += 1; return orig_g(*a, **k)
        def cr(*a, **k):
            calls["allreduce"] += 1; return orig_r(*a, **k)
        dist.all_gather_into_tensor, dist.all_reduce = cg, cr
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=True, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        q = {"input_ids": torch.from_numpy(g["q_ids"]).cuda(), "attention_mask": torch.from_numpy(g["q_mask"]).cuda()}
        p = {"input_ids": torch.from_numpy(g["p_ids"]).cuda(), "attention_mask": torch.from_numpy(g["p_mask"]).cuda()}
        loss = GradCacheStep(m, chunk_size=2)(q, p, sync=True)
        torch.cuda.synchronize()
        sd = dict(m._backbone().named_parameters())
        ret["loss"] = float(loss.item()); ret["backend"] = dist.get_backend(); ret["calls"] = dict(calls)
        ret["grads"] = {n: sd[n].grad.float().cpu().numpy() for n in ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight")}
    finally:
        dist.destroy_process_group()


def check_rccl_world1_step():
    """The cross-device GradCache step on a ONE-rank RCCL process group: chunk-wise all_gather_into_tensor of the reps (ChunkGather) and
    the layer-wise gradient all-reduce (OverlappedGradSync) run on backend 'nccl' on this GPU; with one rank the result must equal the
    reference's local-negatives step (tests/golden/gradcache_tiny.npz).  (Two ranks cannot share one GPU under RCCL: 'Duplicate GPU
    detected' -- the 2-rank paths are covered on gloo, tests/test_dist_gloo.py and check_overlapped_grad_sync.)"""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_rccl_world1_worker, args=(port, d16, ret), nprocs=1, join=True)
    ref_loss = float(g["loss_gradcache"])
    worst = max(float(np.linalg.norm(ret["grads"][n] - g["grad_gradcache/" + n]) / np.linalg.norm(g["grad_gradcache/" + n])) for n in ret["grads"])
    ok = ret["backend"] == "nccl" and abs(ret["loss"] - ref_loss) < 2e-3 * abs(ref_loss) and worst < 6e-2 \
        and ret["calls"]["gather"] >= 2 and ret["calls"]["allreduce"] >= 8
    return _res("GradCache step on a 1-rank RCCL group (gather + all-reduce issued on the GPU)", ok, loss=ret["loss"], loss_ref=ref_loss,
                worst_grad_rel=worst, gathers=ret["calls"]["gather"], allreduces=ret["calls"]["allreduce"])


def _native_comm_worker(rank, port, model_dir, ret):
    """ONE rank; the rep gathers go through the C ABI (grit_comm_*: ncclCommInitRank + grouped ncclAllGather on a side stream)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GRIT_DIST_WORLD1"] = "1"
    os.environ["GRIT_NATIVE_COMM"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from gritlm_amd import comm
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        from gritlm_amd.training.model import packed_all_gather
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        nc = comm.NativeComm.get("cuda:0")
        calls = {"n": 0}
        orig = nc.allgather_packed
        def counted(q, p):
            calls["n"] += 1
            return orig(q, p)
        nc.allgather_packed = counted
        tg = torch.Generator(device="cuda").manual_seed(3)
        a, b = torch.randn((5, 256), generator=tg, device="cuda"), torch.randn((40, 256), generator=tg, device="cuda")
        qa, pa = packed_all_gather(a, b, 1)
        ret["identity"] = bool(torch.equal(qa, a) and torch.equal(pa, b))
        masked = comm.NativeComm("cuda:0", cu_mask=32)                      # a second communicator on a CU-masked side stream
        h = masked.allgather_packed(a, None)
        ret["masked_identity"] = bool(torch.equal(h.wait()[0], a) and h.p_all is None)
        masked.close()
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=True, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        q = {"input_ids": torch.from_numpy(g["q_ids"]).cuda(), "attention_mask": torch.from_numpy(g["q_mask"]).cuda()}
        p = {"input_ids": torch.from_numpy(g["p_ids"]).cuda(), "attention_mask": torch.from_numpy(g["p_mask"]).cuda()}
        from gritlm_amd.training import gradcache as gcm
        seen = []
        orig_init = gcm.ChunkGather.__init__
        def spy(self, n_local, width, dtype, device, *a, **k):
            orig_init(self, n_local, width, dtype, device, *a, **k)
            seen.append((str(dtype), str(device), self.native is not None))
        gcm.ChunkGather.__init__ = spy
        gcs = GradCacheStep(m, chunk_size=2)
        loss = gcs(q, p, sync=True)
        torch.cuda.synchronize()
        sd = dict(m._backbone().named_parameters())
        # one native gather per pass-1 call (pass 1 runs several chunks per call: gradcache.pass1_chunk_rows) of either tower
        ret["pass1_calls"] = -(-q["input_ids"].shape[0] // gcs.pass1_chunk_size) + -(-p["input_ids"].shape[0] // gcs.pass1_chunk_size)
        ret["loss"] = float(loss.item()); ret["gathers"] = calls["n"]; ret["chunk_gathers"] = list(seen)
        ret["grads"] = {n: sd[n].grad.float().cpu().numpy() for n in ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight")}
        nc.close()
    finally:
        dist.destroy_process_group()


def check_native_comm():
    """grit_comm_* (gritlm/training/model.py:49-60 on RCCL below torch.distributed) on a one-rank communicator: the packed
    gather returns its inputs (also on a CU-masked side stream), and the cross-device GradCache step with every rep gather routed
    through it reproduces the reference's step (tests/golden/gradcache_tiny.npz)."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_native_comm_worker, args=(port, d16, ret), nprocs=1, join=True)
    ref_loss, ref_loss16 = float(g["loss_gradcache"]), float(g["loss_gradcache_bf16"])
    worst = max(float(np.linalg.norm(ret["grads"][n] - g["grad_gradcache/" + n]) / np.linalg.norm(g["grad_gradcache/" + n])) for n in ret["grads"])
    ok = ret["identity"] and ret["masked_identity"] and abs(ret["loss"] - ref_loss) <= 1.25 * _yard()["gradcache_tiny/loss_gap_bf16_gradcache"] + LOSS_VS_F32_REF \
        and worst < 3e-2 and ret["gathers"] == 1 + ret["pass1_calls"] and ret["pass1_calls"] >= 2
    return _res("grit_comm_* on a 1-rank RCCL communicator (packed gather, CU-masked stream, GradCache step)", ok, loss=ret["loss"],
                loss_ref=ref_loss, worst_grad_rel=worst, native_gathers=ret["gathers"], identity=ret["identity"],
                masked_identity=ret["masked_identity"], chunk_gathers=str(ret["chunk_gathers"]))


def check_wgrad_accumulation_drift(cfg_name="gqa"):
    """Weight gradients accumulate in bf16 across GradCache chunks (the wgrad GEMM's residual epilogue adds into the packed .grad
    storage, which is what `param.grad +=` does in the reference's bf16 run).  MEASURE the rounding this costs: the same 8 x 8 batch
    as ONE chunk (a single accumulation) vs 16 chunks of 4 rows (a 17-term bf16 running sum), relative L2 per parameter."""
    from gritlm_amd.training.engine import MistralTrainEngine, SyntheticBackbone
    from gritlm_amd.training.gradcache import GradCacheStep
    from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel
    cfg = EncoderConfig.from_dict(synth.CONFIGS[cfg_name])
    idq, mq = synth.make_batch(synth.CONFIGS[cfg_name], 8, 40, seed=15, min_len=9)
    idp, mp_ = synth.make_batch(synth.CONFIGS[cfg_name], 64, 120, seed=16, min_len=20)
    q = {"input_ids": torch.from_numpy(idq).to(DEV), "attention_mask": torch.from_numpy(mq).to(DEV)}
    p = {"input_ids": torch.from_numpy(idp).to(DEV), "attention_mask": torch.from_numpy(mp_).to(DEV)}
    res = {}
    for chunk in (64, 4):
        bb = SyntheticBackbone(cfg, DEV, seed=3)
        m = GritLMTrainModel.__new__(GritLMTrainModel)
        torch.nn.Module.__init__(m)
        m.model, m.projection, m.pooling_method, m.normalized, m.attn, m.embedding_attr = bb, None, "mean", True, "bbcc", None
        m.emb_loss_fn = DistributedContrastiveLoss(0.02, False)
        m.train_engine = MistralTrainEngine(bb, cfg, DEV)
        loss = GradCacheStep(m, chunk)(dict(q), dict(p), sync=False)
        res[chunk] = (float(loss.item()), {n: f32(t.grad) for n, t in bb.named_parameters()})
    worst, worst_name = 0.0, ""
    for n, g1 in res[64][1].items():
        rel = float(np.linalg.norm(res[4][1][n] - g1) / (np.linalg.norm(g1) + 1e-20))
        if rel > worst:
            worst, worst_name = rel, n
    ok = abs(res[64][0] - res[4][0]) < 1e-4 and worst < 2e-2
    return _res("bf16 wgrad accumulation: 1 chunk vs 17-term running sum", ok, loss=res[4][0], worst_rel_l2=worst, worst_param=worst_name)


def check_ce(T=300, V=1003):
    """Fused vocabulary cross entropy (grit_ce_fwd / grit_ce_bwd) vs fp64 numpy; ignore_index rows, V not a multiple of 8 via ld."""
    rng = np.random.default_rng(81)
    ld = (V + 7) // 8 * 8
    logits = (rng.standard_normal((T, ld)) * 3).astype(np.float32)
    labels = rng.integers(0, V, size=T).astype(np.int64)
    labels[::7] = -100
    tl = bf(logits)
    view = tl[:, :V]
    tlab = torch.from_numpy(labels).to(DEV)
    lse, loss_row = ops.ce_fwd(view, tlab)
    z = f32(tl)[:, :V].astype(np.float64)
    mx = z.max(1, keepdims=True)
    lse_ref = mx[:, 0] + np.log(np.exp(z - mx).sum(1))
    keep = labels >= 0
    nll_ref = np.where(keep, lse_ref - z[np.arange(T), np.where(keep, labels, 0)], 0.0)
    e1 = float(np.max(np.abs(f32(lse) - lse_ref))); e2 = float(np.max(np.abs(f32(loss_row) - nll_ref)))
    dev_scale = torch.tensor([0.5], dtype=torch.float32, device=DEV)
    ops.ce_bwd_(view, tlab, lse, 3.0, dev_scale)
    p = np.exp(z - lse_ref[:, None]); p[np.arange(T), np.where(keep, labels, 0)] -= keep
    gref = p * 1.5 * keep[:, None]
    got = f32(tl)[:, :V]
    e3 = float(np.max(np.abs(got - gref) / (2.0 ** -8 * np.abs(gref) + 1e-6)))
    return _res(f"cross entropy fwd/bwd [T={T},V={V}]", e1 < 1e-4 and e2 < 1e-4 and e3 < 1.01, lse_abs=e1, nll_abs=e2, grad_err_over_ulp=e3)


def check_generative_step(kind="mixed"):
    """Native generative branch (causal attention fwd/bwd, lm_head GEMMs, fused CE) vs the REFERENCE's loss_gen and parameter
    gradients (tests/golden/generative_tiny.npz: GritLMTrainModel.forward(generative=...) on the reference's MistralForCausalLM)."""
    import tempfile
    from gritlm_amd.training import GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "generative_tiny.npz"))
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLMTrainModel(model_name_or_path=d16, mode="unified", pooling_method="mean", normalized=True, attn="bbcc", temperature=0.02,
                             negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16, loss_gen_type=kind,
                             loss_gen_factor=float(g[f"factor_{kind}"]))
        m.enable_native()
        ok &= m.train_engine.lm_head is not None
        for packed in (True, False):
            m.native_packed = packed
            m.model.zero_grad(set_to_none=True)
            gen = {"input_ids": torch.from_numpy(g["input_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["attention_mask"]).to(DEV),
                   "labels": torch.from_numpy(g["labels"]).to(DEV)}
            o = m(generative=gen)
            o.loss_gen.backward()
            ref_loss = float(g[f"loss_gen_{kind}"])
            tag = "packed" if packed else "padded"
            out[f"loss_{tag}"] = float(o.loss_gen.item()); out["loss_ref"] = ref_loss
            ok &= abs(out[f"loss_{tag}"] - ref_loss) < 1e-2 * max(1.0, abs(ref_loss))
            sd = dict(m.model.named_parameters())
            worst = 0.0
            for key in g.files:
                if not key.startswith(f"grad_{kind}/"):
                    continue
                n = key.split("/", 1)[1]
                ref = g[key]; got = f32(sd[n].grad)
                rel = float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-20))
                worst = max(worst, rel)
                if packed:
                    out[n.replace("model.", "").replace("layers.", "L").replace(".weight", "")] = rel
            out[f"worst_grad_rel_{tag}"] = worst
            ok &= worst < 6e-2
    return _res(f"native generative step [{kind}] vs reference loss_gen+grads", bool(ok), **out)


def check_gemv(B=3, N=1030, K=512, epi=EPI_STORE, seed=85):
    x, w = rnd((B, K), seed), rnd((N, K), seed + 1, 0.05)
    ref = f32(bf(x)).astype(np.float64) @ f32(bf(w)).astype(np.float64).T
    if epi == EPI_RESIDUAL:
        r = rnd((B, N), seed + 2)
        out = f32(ops.gemv(bf(x), bf(w), epilogue=epi, residual=bf(r)))
        ref = O.bf16_round(ref.astype(np.float32)).astype(np.float64) + f32(bf(r))
    elif epi == EPI_SWIGLU:
        I = N // 2
        wi = swiglu_interleave(bf(w[:I]), bf(w[I:]))
        out = f32(ops.gemv(bf(x), wi, epilogue=epi))
        gq, uq = O.bf16_round(ref[:, :I].astype(np.float32)), O.bf16_round(ref[:, I:].astype(np.float32))
        ref = (O.bf16_round(O.silu(gq.astype(np.float64)).astype(np.float32)) * uq).astype(np.float64)
        # the fused GEMM epilogue must agree with the GEMV on the same rows (the prefill and the decode path of one model)
        big = f32(ops.gemm_nt(bf(np.repeat(x, 8, axis=0)), wi, epilogue=epi))[::8]
        same = float(np.max(np.abs(big - out)))
    else:
        out = f32(ops.gemv(bf(x), bf(w)))
    scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-9
    err = float(np.max(np.abs(out - ref) / (1.2e-2 * np.abs(ref) + 1e-2 * scale)))
    return _res(f"gemv[B={B},N={N},K={K},epi={epi}]", err < 1.0, max_err_over_tol=err)


def check_decode_fused_ops(B=2, H=512, N=768, nq=4, nkv=2, Lmax=256):
    """rmsnorm+gemv == gemv(rmsnorm) bit for bit; rope+kv-append == rope_qk_pos_ + kv_append bit for bit."""
    d = 128
    x, lnw, w = bf(rnd((B, H), 91)), bf(1.0 + 0.1 * rnd((H,), 92)), bf(rnd((N, H), 93, 0.05))
    a = ops.rmsnorm_gemv(x, lnw, 1e-5, w)
    b = ops.gemv(ops.rmsnorm(x, lnw, 1e-5), w)
    ok = bool(torch.equal(a, b))
    wi = swiglu_interleave(w[:N // 2], w[N // 2:])
    ok &= bool(torch.equal(ops.rmsnorm_gemv(x, lnw, 1e-5, wi, epilogue=EPI_SWIGLU), ops.gemv(ops.rmsnorm(x, lnw, 1e-5), wi, epilogue=EPI_SWIGLU)))
    from gritlm_amd.encoder import rope_tables
    cos, sin = rope_tables(Lmax, d, 10000.0, True, DEV)
    qkv = bf(rnd((B, (nq + 2 * nkv) * d), 94))
    lens = torch.tensor([200, 7][:B], dtype=torch.int32, device=DEV)
    ck1, cv1 = torch.zeros((B, nkv, Lmax, d), dtype=torch.bfloat16, device=DEV), torch.zeros((B, nkv, Lmax, d), dtype=torch.bfloat16, device=DEV)
    ck2, cv2 = ck1.clone(), cv1.clone()
    q1, q2 = qkv.clone(), qkv.clone()
    ops.rope_kv_append(q1, cos, sin, ck1, cv1, lens, nq, nkv, d)
    ops.rope_qk_pos_(q2, cos, sin, lens, nq, nkv, d)
    ops.kv_append(q2, ck2, cv2, lens, nq, nkv, d)
    ok &= bool(torch.equal(q1[:, :nq * d], q2[:, :nq * d])) and bool(torch.equal(ck1, ck2)) and bool(torch.equal(cv1, cv2)) and float(ck1.abs().sum()) > 0
    # attention with RoPE + append folded in == rope_kv_append followed by attn_decode, bit for bit (caches included)
    ck3, cv3 = bf(rnd((B, nkv, Lmax, d), 95)), bf(rnd((B, nkv, Lmax, d), 96))
    for bi in range(B):            # everything from the new token's slot on is NaN: the fused kernel requests its K / V rows before it
        ck3[bi, :, int(lens[bi]):] = float("nan"); cv3[bi, :, int(lens[bi]):] = float("nan")     # appends -- nothing stale may leak
    ck4, cv4 = ck3.clone(), cv3.clone()
    qa, qb = qkv.clone(), qkv.clone()
    ws = ops.attn_decode_workspace(B, nq, nkv, Lmax, DEV)
    o3 = torch.empty((B, nq * d), dtype=torch.bfloat16, device=DEV); o4 = torch.empty_like(o3)
    ops.rope_kv_append(qa, cos, sin, ck3, cv3, lens, nq, nkv, d)
    ops.attn_decode(qa, ck3, cv3, lens, o3, ws, nq, nkv, d)
    ops.attn_decode_rope(qb, cos, sin, ck4, cv4, lens, o4, ws, nq, nkv, d)
    nn = torch.nan_to_num
    ok &= bool(torch.equal(o3, o4)) and bool(torch.isfinite(o4.float()).all()) and bool(torch.equal(nn(ck3), nn(ck4))) and bool(torch.equal(nn(cv3), nn(cv4)))
    ok &= all(bool(torch.isfinite(ck4[bi, :, :int(lens[bi]) + 1].float()).all()) for bi in range(B))
    return _res("decode fused ops (rmsnorm+gemv, rope+kv-append, rope+append+attention) == unfused kernels", ok)


def check_decode_deferred_norm(B=2, H=4096, N=1024):
    """grit_rmsnorm_gemv_bf16_deferred: out = rsqrt(mean x^2 + eps) * (W (x * w_ln)) against the same expression in fp64 -- the only
    rounding is the bf16 output (2^-9 relative), the sum of squares is complete (every split-K quarter contributed), for 1 / 2 / 3 / 8 rows
    and for the SwiGLU epilogue; and it stays within the bf16 noise of the exact fused form (which rounds x_n twice)."""
    ok, det = True, {}
    lnw, w = bf(1.0 + 0.1 * rnd((H,), 192)), bf(rnd((N, H), 193, 0.05))
    wi = swiglu_interleave(w[:N // 2], w[N // 2:])
    for Bi in (1, 2, 3, 8)[:4 if B >= 2 else 1]:
        x = bf(3.0 * rnd((Bi, H), 191 + Bi))
        xd, ld, wd = x.double(), lnw.double(), w.double()
        inv = torch.rsqrt((xd * xd).mean(dim=1, keepdim=True) + 1e-5)
        ref = ((xd * ld) @ wd.T) * inv
        got = ops.rmsnorm_gemv(x, lnw, 1e-5, w, deferred=True).double()
        scale = float(ref.pow(2).mean().sqrt())
        e = float(((got - ref).abs() / (2.0 ** -8 * ref.abs() + 1e-3 * scale)).max())
        exact = ops.rmsnorm_gemv(x, lnw, 1e-5, w).double()
        e_exact = float(((exact - ref).abs() / (2.0 ** -8 * ref.abs() + 1e-3 * scale)).max())
        d_forms = float((got - exact).norm() / exact.norm())
        g, u = ref[:, :N // 2].float(), ref[:, N // 2:].float()
        rb = lambda t: t.to(torch.bfloat16).float()
        ref_sw = (rb(torch.nn.functional.silu(rb(g))) * rb(u)).double()
        got_sw = ops.rmsnorm_gemv(x, lnw, 1e-5, wi, epilogue=EPI_SWIGLU, deferred=True).double()
        sc_sw = float(ref_sw.pow(2).mean().sqrt())
        e_sw = float(((got_sw - ref_sw).abs() / (2.0 ** -6 * ref_sw.abs() + 1e-2 * sc_sw)).max())
        det[f"B{Bi}"] = dict(err_over_tol=e, exact_form_err_over_tol=e_exact, rel_diff_to_exact_form=d_forms, swiglu_err_over_tol=e_sw)
        ok &= e < 1.0 and e_sw < 1.0 and d_forms < 1e-2 and bool(torch.isfinite(got).all())
    return _res("deferred rmsnorm+gemv == rsqrt(mean x^2) * W (x * w_ln) in fp64 up to the output rounding", ok, **det)


def check_attn_decode(B=3, nq=8, nkv=2, Lmax=768, lens=(700, 0, 255)):
    d = 128
    rng = np.random.default_rng(87)
    ck, cv = rnd((B, nkv, Lmax, d), 88, 0.7), rnd((B, nkv, Lmax, d), 89)
    q = rnd((B, (nq + 2 * nkv) * d), 90, 0.7)
    tl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    tck, tcv, tq = bf(ck), bf(cv), bf(q)
    out = torch.empty((B, nq * d), dtype=torch.bfloat16, device=DEV)
    ops.attn_decode(tq, tck, tcv, tl, out, ops.attn_decode_workspace(B, nq, nkv, Lmax, DEV), nq, nkv, d)
    got = f32(out).reshape(B, nq, d)
    worst = 0.0
    for b in range(B):
        L = lens[b] + 1
        for h in range(nq):
            kk, vv = f32(tck)[b, h // (nq // nkv), :L].astype(np.float64), f32(tcv)[b, h // (nq // nkv), :L].astype(np.float64)
            sc = kk @ f32(tq)[b, h * d:(h + 1) * d].astype(np.float64) / np.sqrt(d)
            p = np.exp(sc - sc.max()); p /= p.sum()
            worst = max(worst, float(np.max(np.abs(got[b, h] - p @ vv))))
    return _res(f"attn_decode[B={B},nq={nq},nkv={nkv},lens={list(lens)}]", worst < 1e-2 and not np.isnan(got).any(), max_abs=worst)


def check_native_generate(cfg_name="tiny", P=21, new=10, rows=2, tol=0.06, fuse_norm=None):
    """Greedy generation on the native decoder: logits of every generated position vs the fp32 oracle run over the same token
    sequence (a) from a plain prompt (causal prefill), (b) on top of the cached K/V of a bidirectionally encoded document (the RAG
    doc-caching flow); tokens agree with the oracle's argmax wherever its top-2 margin is clear; HIP-graph replay == eager launches."""
    from gritlm_amd.decoder import MistralDecoder
    eng, cfg, w = build_engine(cfg_name, 6)
    rng = np.random.default_rng(97)
    lm = O.bf16_round((rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"])) * 0.05).astype(np.float32))
    dec = MistralDecoder(eng, torch.from_numpy(lm))
    if fuse_norm is not None:      # "none" / "qkv" / "all": the exact forms (rows > 2 then take the un-fused RMSNorm decode step); default: deferred
        dec.fuse_norm = fuse_norm
    prompt = rng.integers(3, cfg["vocab_size"], size=(rows, P)).astype(np.int64)
    ok, out = True, {}

    def judge(tag, toks, logits, ref_logits):
        nonlocal ok
        err = float(np.max(np.abs(logits - ref_logits)))
        srt = np.sort(ref_logits, axis=-1)
        clear = (srt[..., -1] - srt[..., -2]) > 4 * err + 1e-3
        agree = (toks == ref_logits.argmax(-1))[clear]
        out[f"{tag}_logit_abs_err"] = err; out[f"{tag}_logit_std"] = float(ref_logits.std()); out[f"{tag}_clear_frac"] = float(clear.mean())
        ok &= err < tol * float(ref_logits.std()) + 2e-2 and bool(agree.all())

    # (a) plain prompt
    toks, lg = dec.generate(torch.from_numpy(prompt).to(DEV), new, return_logits=True)
    toks, lg = toks.cpu().numpy(), f32(lg)
    for b in range(rows):
        seq = np.concatenate([prompt[b], toks[b]])[None]
        h = O.mistral_encode(w, cfg, seq, np.ones_like(seq), causal=True)
        judge(f"prompt{b}", toks[b], lg[b], (h[0] @ lm.T)[P - 1:P - 1 + new])
    dec.use_graph = True
    t_graph = dec.generate(torch.from_numpy(prompt).to(DEV), new).cpu().numpy()
    dec.use_graph = False
    t_eager = dec.generate(torch.from_numpy(prompt).to(DEV), new).cpu().numpy()
    ok &= np.array_equal(t_graph, toks) and np.array_equal(t_eager, toks)
    # (b) document prefix (bidirectional) + query continuation
    doc = rng.integers(3, cfg["vocab_size"], size=(1, 70)).astype(np.int64)
    _, kv = eng.forward(torch.from_numpy(doc).to(DEV), torch.ones((1, 70), dtype=torch.int64, device=DEV), return_kv=True)
    toks2, lg2 = dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kv, return_logits=True)
    toks2, lg2 = toks2.cpu().numpy()[0], f32(lg2)[0]
    _, kv_ref = O.mistral_encode(w, cfg, doc, np.ones_like(doc), return_layers="kv")
    ref2 = O.mistral_continue(w, cfg, kv_ref, 70, np.concatenate([prompt[0], toks2]), lm)[P - 1:P - 1 + new]
    judge("doc_prefix", toks2, lg2, ref2)
    dec.use_graph = True
    ok &= np.array_equal(dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kv).cpu().numpy()[0], toks2)
    return _res(f"native greedy generation [{cfg_name}] vs oracle (prompt prefill, document-KV prefix, graph replay)", bool(ok), **out)


def check_gemv_f16(N=1030, K=512, seed=185):
    """The fp16-operand GEMV forms (grit_gemv_f16, grit_rmsnorm_gemv_f16_deferred) against fp64 on the SAME fp16 operands, for 1 / 2 / 3 / 8
    rows: STORE and RESIDUAL write fp32 (only the fp32 accumulation separates them from fp64), RESIDUAL's out16 is the fp16 rounding of
    exactly that fp32 value, SWIGLU writes fp16(silu(g) u) -- one rounding; an activation beyond 65504 raises the device's overflow flag."""
    ok, det = True, {}
    h16 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.float16)
    w = h16(rnd((N, K), seed + 1, 0.05))
    Ns = N // 32 * 32
    wi = swiglu_interleave(w[:Ns // 2], w[Ns // 2:Ns])
    lnw = bf(1.0 + 0.1 * rnd((K,), seed + 2))
    ops.f16_overflow_flag(DEV, clear=True)
    for B in (1, 2, 3, 8):
        x16 = h16(rnd((B, K), seed + 10 + B))
        x32 = h16(3.0 * np.random.default_rng(seed + 20 + B).standard_normal((B, K)).astype(np.float32))       # (the stream's fp16 copy)
        r32 = torch.from_numpy(np.random.default_rng(seed + 30 + B).standard_normal((B, N)).astype(np.float32)).to(DEV)
        ref = x16.double() @ w.double().T
        sc = float(ref.pow(2).mean().sqrt())
        got = ops.gemv(x16, w)
        e_store = float(((got.double() - ref).abs() / (4e-6 * ref.abs() + 4e-6 * sc)).max())
        got = ops.gemv(x16, w, epilogue=EPI_RESIDUAL, residual=r32)
        e_res = float(((got.double() - (ref + r32.double())).abs() / (4e-6 * (ref.abs() + r32.abs().double()) + 4e-6 * sc)).max())
        rr = r32.clone(); r16 = torch.empty((B, N), dtype=torch.float16, device=DEV)
        ops.gemv(x16, w, out=rr, epilogue=EPI_RESIDUAL, residual=rr, out16=r16)                      # in place + the fp16 copy, as the decode step runs it
        same = bool(torch.equal(rr, got)) and bool(torch.equal(r16, got.to(torch.float16)))
        g, u = ref[:, :Ns // 2], ref[:, Ns // 2:Ns]
        ref_sw = torch.nn.functional.silu(g) * u
        got_sw = ops.gemv(x16, wi, epilogue=EPI_SWIGLU)
        e_sw = float(((got_sw.double() - ref_sw).abs() / (2.0 ** -11 * 1.02 * ref_sw.abs() + 1e-5 * float(ref_sw.pow(2).mean().sqrt()) + 6e-8)).max())
        # deferred norm on the fp32 stream
        xd = x32.double()
        inv = torch.rsqrt((xd * xd).mean(dim=1, keepdim=True) + 1e-5)
        refn = ((xd * lnw.double()) @ w.double().T) * inv
        scn = float(refn.pow(2).mean().sqrt())
        gotn = ops.rmsnorm_gemv(x32, lnw, 1e-5, w, deferred=True)
        e_n = float(((gotn.double() - refn).abs() / (8e-6 * refn.abs() + 8e-6 * scn)).max())
        refn_sw = torch.nn.functional.silu(refn[:, :Ns // 2]) * refn[:, Ns // 2:Ns]
        gotn_sw = ops.rmsnorm_gemv(x32, lnw, 1e-5, wi, epilogue=EPI_SWIGLU, deferred=True)
        e_nsw = float(((gotn_sw.double() - refn_sw).abs() / (2.0 ** -11 * 1.02 * refn_sw.abs() + 1e-5 * float(refn_sw.pow(2).mean().sqrt()) + 6e-8)).max())
        det[f"B{B}"] = dict(store=e_store, residual=e_res, swiglu=e_sw, norm_store=e_n, norm_swiglu=e_nsw)
        ok &= max(e_store, e_res, e_sw, e_n, e_nsw) < 1.0 and same and got.dtype == torch.float32 and got_sw.dtype == torch.float16
    clean = not ops.f16_overflow_flag(DEV, clear=True)
    big = ops.gemv(h16(np.full((1, K), 60.0, dtype=np.float32)), swiglu_interleave(h16(np.full((16, K), 8.0)), h16(np.full((16, K), 8.0))), epilogue=EPI_SWIGLU)
    raised = ops.f16_overflow_flag(DEV, clear=True) and bool(torch.isinf(big).any())
    ok &= clean and raised
    return _res(f"gemv on fp16 operands [N={N},K={K}] vs fp64 (err over tol; fp32 out 4e-6, fp16 out one rounding)", ok, flag_clean=clean,
                flag_raised=raised, **{k: max(v.values()) for k, v in det.items()})


def check_gemv_expert(E=5, N=1024, K=512, big=False):
    """grit_gemv_{bf16,f16}_expert: x times the matrix of an [E,N,K] stack chosen by an int32 in DEVICE memory == the plain GEMV on that
    slice, bit for bit, for every expert, both operand formats, STORE and SWIGLU, 1 and 3 rows; ``big``: the Mixtral-8x7B w13 stack's
    strides (8 x 28672 x 4096: the index times 117 M elements).  And grit_moe_decode_combine_f32 against the same expression in torch."""
    if big:
        E, N, K = 8, 28672, 4096
    ok, det = True, {}
    g = torch.Generator(device=DEV).manual_seed(9)
    stack_bf = (torch.randn((E, N, K), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
    for dt in (torch.bfloat16, torch.float16):
        stack = stack_bf if dt == torch.bfloat16 else stack_bf.to(torch.float16)
        for B in (1, 3):
            x = torch.randn((B, K), generator=g, device=DEV).to(dt)
            idx = torch.zeros((E, 2), dtype=torch.int32, device=DEV)
            for e in (range(E) if not big else (0, E - 1)):
                idx[e, 1] = e
                for epi in (EPI_STORE, EPI_SWIGLU):
                    got = ops.gemv_expert(x, stack, idx[e, 1:2], epilogue=epi)
                    ref = ops.gemv(x, stack[e], epilogue=epi)
                    same = bool(torch.equal(got, ref)) and got.dtype == ref.dtype
                    ok &= same
                    det[f"{str(dt)[6:]}_B{B}_e{e}_epi{epi}"] = same
    del stack_bf, stack
    B, H = 3, 1000
    h = torch.randn((B, H), generator=g, device=DEV); y = torch.randn((2 * B, H), generator=g, device=DEV); w = torch.rand((B, 2), generator=g, device=DEV)
    want = h + (y.view(B, 2, H) * w.unsqueeze(-1)).sum(dim=1)
    h2, h16 = h.clone(), torch.empty((B, H), dtype=torch.float16, device=DEV)
    ops.f16_overflow_flag(DEV, clear=True)
    ops.moe_decode_combine_f32(h2, h16, y, w)
    comb = float((h2 - want).abs().max()) < 1e-6 and bool(torch.equal(h16, h2.to(torch.float16))) and not ops.f16_overflow_flag(DEV, clear=True)
    h3 = h.clone(); h3[1, 7] = 1e6
    ops.moe_decode_combine_f32(h3, h16, y, w)
    comb &= ops.f16_overflow_flag(DEV, clear=True)
    ok &= comb
    return _res(f"gemv_expert[E={E},N={N},K={K}] == gemv on the slice (bit for bit); moe_decode_combine_f32", bool(ok), all_equal=all(det.values()), cases=len(det),
                combine_ok=comb)


def check_attn_decode_f16(B=3, nq=8, nkv=2, Lmax=768, lens=(700, 0, 255)):
    """grit_attn_decode_rope_f16: fp32 q|k|v row, fp16 caches.  The appended rows are fp16(rope(k)) and fp16(v) -- ONE rounding from the
    fp32 row --, the context equals fp64 attention of the UNROUNDED rotated q over the cache as it then stands, up to the fp16 output
    rounding; slots behind the new token are NaN before the call and nothing of them leaks."""
    from gritlm_amd.encoder import rope_tables
    d = 128
    G = nq // nkv
    cos, sin = rope_tables(Lmax, d, 10000.0, False, DEV)
    rng = np.random.default_rng(287)
    ck = torch.from_numpy(rng.standard_normal((B, nkv, Lmax, d)).astype(np.float32) * 0.7).to(DEV).to(torch.float16)
    cv = torch.from_numpy(rng.standard_normal((B, nkv, Lmax, d)).astype(np.float32)).to(DEV).to(torch.float16)
    qkv = torch.from_numpy(rng.standard_normal((B, (nq + 2 * nkv) * d)).astype(np.float32) * 0.7).to(DEV)
    tl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    for b in range(B):
        ck[b, :, lens[b]:] = float("nan"); cv[b, :, lens[b]:] = float("nan")
    out = torch.empty((B, nq * d), dtype=torch.float16, device=DEV)
    ops.f16_overflow_flag(DEV, clear=True)
    ops.attn_decode_rope(qkv, cos, sin, ck, cv, tl, out, ops.attn_decode_workspace(B, nq, nkv, Lmax, DEV), nq, nkv, d)
    got = out.double().cpu().numpy().reshape(B, nq, d)
    c64, s64 = cos.double().cpu().numpy(), sin.double().cpu().numpy()
    row = qkv.double().cpu().numpy()

    def rot(x, pos):            # x*cos + rotate_half(x)*sin
        x1, x2 = x[..., :d // 2], x[..., d // 2:]
        return np.concatenate([x1 * c64[pos] - x2 * s64[pos], x2 * c64[pos] + x1 * s64[pos]], axis=-1)

    worst, worst_k, worst_v, ok = 0.0, 0.0, 0.0, True
    K, V = ck.double().cpu().numpy(), cv.double().cpu().numpy()
    for b in range(B):
        pos = lens[b]
        kn = rot(row[b, nq * d:(nq + nkv) * d].reshape(nkv, d), pos)
        vn = row[b, (nq + nkv) * d:].reshape(nkv, d)
        worst_k = max(worst_k, float(np.max(np.abs(K[b, :, pos] - kn.astype(np.float16).astype(np.float64)))))      # RNE of the fp64 rotation: at most
        worst_v = max(worst_v, float(np.max(np.abs(V[b, :, pos] - vn.astype(np.float16).astype(np.float64)))))      # one fp16 ulp from fp16(fp32 rotation)
        ok &= bool(np.isfinite(K[b, :, :pos + 1]).all()) and bool(np.isnan(K[b, :, pos + 1:]).all())
        for h in range(nq):
            qh = rot(row[b, h * d:(h + 1) * d], pos)
            sc = K[b, h // G, :pos + 1] @ qh / np.sqrt(d)
            p = np.exp(sc - sc.max()); p /= p.sum()
            worst = max(worst, float(np.max(np.abs(got[b, h] - p @ V[b, h // G, :pos + 1]))))
    ok &= worst < 1.5e-3 and worst_k < 4e-3 and worst_v == 0.0 and not np.isnan(got).any() and not ops.f16_overflow_flag(DEV, clear=True)
    # a new key beyond the fp16 range raises the flag
    q2 = qkv.clone(); q2[0, nq * d] = 1e6
    ops.attn_decode_rope(q2, cos, sin, ck, cv, tl, out, ops.attn_decode_workspace(B, nq, nkv, Lmax, DEV), nq, nkv, d)
    raised = ops.f16_overflow_flag(DEV, clear=True)
    return _res(f"attn_decode_rope on fp16 K/V [B={B},nq={nq},nkv={nkv},lens={list(lens)}] vs fp64", bool(ok and raised), ctx_max_abs=worst,
                new_k_max_abs=worst_k, new_v_max_abs=worst_v, flag_raised=raised)


def check_argmax_f32(B=3, V=32003):
    lg = torch.from_numpy(np.random.default_rng(311).standard_normal((B, (V + 7) // 8 * 8)).astype(np.float32)).to(DEV)[:, :V]
    lg[0, 17] = lg[0, V // 2 + 3] = 9.0           # a tie: the lower index wins
    lg[1, V - 1] = 11.0                           # the tail beyond the last full chunk of 8
    nxt = torch.zeros((B,), dtype=torch.int64, device=DEV); lens = torch.arange(5, 5 + B, dtype=torch.int32, device=DEV)
    hist = torch.zeros((B, 4), dtype=torch.int64, device=DEV); step = torch.tensor([2], dtype=torch.int32, device=DEV)
    ops.argmax_advance(lg, nxt, lens, hist, step)
    ref = lg.argmax(dim=1); ref[0] = 17
    ok = bool(torch.equal(nxt, ref)) and lens.tolist() == list(range(6, 6 + B)) and bool(torch.equal(hist[:, 2], ref)) and int(step) == 3
    return _res("argmax_advance on fp32 logits (ties, tail, history, lens)", ok)


def check_native_generate_f16(cfg_name="tiny", P=21, new=10, rows=2, policy="f16_operands", cos_bound=1e-5):
    """The decoder on fp16 operands (engine policy ``f16_operands`` / ``f16_stream``): teacher-forced logits of every generated position against
    the fp32 oracle at the NORTH-STAR's level -- 1 - cos(logits row) < ``cos_bound`` (<= 1e-4 / 10), at least 10x below the bf16 decoder on
    the same weights and tokens -- (a) from a prompt (one causal pass under the policy), (b) on top of the fp16 document K/V of
    encode(get_cache=True) under the policy (kv_dtype=None), the RAG flow; greedy tokens agree wherever the oracle's margin is clear;
    graph replay == eager; rows 1..3; overflow: on_overflow='bf16' falls back, 'raise' raises."""
    from gritlm_amd._lib import GritHipError
    from gritlm_amd.decoder import MistralDecoder
    eng, cfg, w = build_engine(cfg_name, 6)
    rng = np.random.default_rng(97)
    lm = O.bf16_round((rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"])) * 0.05).astype(np.float32))
    dec = MistralDecoder(eng, torch.from_numpy(lm))
    prompt = rng.integers(3, cfg["vocab_size"], size=(rows, P)).astype(np.int64)
    doc = rng.integers(3, cfg["vocab_size"], size=(1, 70)).astype(np.int64)
    ok, out = True, {}
    omc = lambda a, b: float(np.max(1.0 - np.sum(a * b, axis=-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))))
    res = {}
    for pol in ("bf16", policy):
        eng.set_precision(pol)
        ones = torch.ones((1, 70), dtype=torch.int64, device=DEV)
        _, kv = eng.forward(torch.from_numpy(doc).to(DEV), ones, return_kv=True, kv_dtype=None)
        toks2, lg2 = dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kv, return_logits=True)
        ok &= dec.last_precision == ("bf16" if pol == "bf16" else "f16") and kv[0][0].dtype == (torch.bfloat16 if pol == "bf16" else torch.float16)
        ok &= lg2.dtype == (torch.bfloat16 if pol == "bf16" else torch.float32)
        toks2, lg2 = toks2.cpu().numpy()[0], f32(lg2)[0].astype(np.float64)
        _, kv_ref = O.mistral_encode(w, cfg, doc, np.ones_like(doc), return_layers="kv")
        ref2 = O.mistral_continue(w, cfg, kv_ref, 70, np.concatenate([prompt[0], toks2]), lm)[P - 1:P - 1 + new].astype(np.float64)
        res[pol] = dict(doc=omc(lg2, ref2), doc_abs=float(np.max(np.abs(lg2 - ref2))), std=float(ref2.std()))
        srt = np.sort(ref2, axis=-1)
        clear = (srt[..., -1] - srt[..., -2]) > 4 * res[pol]["doc_abs"] + 1e-4
        ok &= bool((toks2 == ref2.argmax(-1))[clear].all())
        res[pol]["clear_frac"] = float(clear.mean())
        if pol != "bf16":
            dec.use_graph = False
            ok &= np.array_equal(dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kv).cpu().numpy()[0], toks2)
            dec.use_graph = True
            ok &= np.array_equal(dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kv).cpu().numpy()[0], toks2)
            # the same prefix handed over as the reference's bf16 cache: narrowed exactly, decode continues on fp16 operands
            _, kvb = eng.forward(torch.from_numpy(doc).to(DEV), ones, return_kv=True)
            _, lgb = dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, past_key_values=kvb, return_logits=True)
            res[pol]["doc_bf16_cache"] = omc(f32(lgb)[0].astype(np.float64), ref2)
            # several rows on top of no prefix: the prompt rides on the decode path (empty past), every row against the oracle
            empty = [(torch.zeros((rows, cfg["num_key_value_heads"], 0, eng.cfg.head_dim), dtype=torch.float16, device=DEV),) * 2] * cfg["num_hidden_layers"]
            toks, lg = dec.generate(torch.from_numpy(prompt).to(DEV), new, past_key_values=empty, return_logits=True)
            toks, lg = toks.cpu().numpy(), f32(lg).astype(np.float64)
            worst = 0.0
            for b in range(rows):
                seq = np.concatenate([prompt[b], toks[b]])[None]
                hh = O.mistral_encode(w, cfg, seq, np.ones_like(seq), causal=True)
                worst = max(worst, omc(lg[b], (hh[0] @ lm.T)[P - 1:P - 1 + new].astype(np.float64)))
            res[pol]["prompt_on_decode_path"] = worst
            # plain prompt: ONE causal pass under the fp16 policy (grit_attn_causal_f16_fwd), fp16 K/V straight into the cache, then decode
            toks_p, lg_p = dec.generate(torch.from_numpy(prompt[:1]).to(DEV), new, return_logits=True)
            seq = np.concatenate([prompt[0], toks_p.cpu().numpy()[0]])[None]
            hh = O.mistral_encode(w, cfg, seq, np.ones_like(seq), causal=True)
            res[pol]["prompt_prefill"] = omc(f32(lg_p)[0].astype(np.float64), (hh[0] @ lm.T)[P - 1:P - 1 + new].astype(np.float64))
            ok &= eng.precision == pol and not eng.causal
    f, b = res[policy], res["bf16"]
    ok &= f["doc"] < cos_bound and f["prompt_on_decode_path"] < cos_bound and f["doc"] * 10 < b["doc"] and f["doc_bf16_cache"] < 1e-4
    ok &= f["prompt_prefill"] < cos_bound
    # overflow: a huge lm_head-independent activation -- scale one layer's down_proj so that act * W stays finite but the NEXT act overflows
    eng2, cfg2, w2 = build_engine(cfg_name, 6)
    eng2.set_precision(policy)
    dec2 = MistralDecoder(eng2, torch.from_numpy(lm))
    (eng2.layers[0].w13 if cfg.get("num_local_experts") else eng2.layers[0].wgu).mul_(400.0)       # |gate|, |up| x 400: silu(g) u beyond 65504 for some elements
    fell = raised = False
    try:
        dec2.generate(torch.from_numpy(prompt[:1]).to(DEV), 3, past_key_values=[(k[:, :, :0], v[:, :, :0]) for k, v in kv])
    except GritHipError:
        raised = True
    t_fb = dec2.generate(torch.from_numpy(prompt[:1]).to(DEV), 3, past_key_values=[(k[:, :, :0], v[:, :, :0]) for k, v in kv], on_overflow="bf16")
    fell = dec2.last_precision == "bf16 (f16 overflow)" and t_fb.shape == (1, 3)
    ok &= raised and fell
    return _res(f"native decode on fp16 operands [{cfg_name}, {policy}] vs fp32 oracle: 1-cos of the logits", bool(ok), f16_doc=f["doc"], bf16_doc=b["doc"],
                f16_prompt_on_decode_path=f["prompt_on_decode_path"], f16_doc_from_bf16_cache=f["doc_bf16_cache"],
                f16_prompt_prefill=f["prompt_prefill"], f16_logit_abs_err=f["doc_abs"], bf16_logit_abs_err=b["doc_abs"], logit_std=f["std"],
                clear_frac=f["clear_frac"], overflow_raised=raised, overflow_fell_back=fell)


def check_native_generate_prompt_chunk(cfg_name="gqa", B=2, P=11, new=5, policy="bf16"):
    """The prompt tokens handed to generate() next to past_key_values, all at once (MistralDecoder._prompt_rows: the decode step's kernels
    over B x P rows, grit_rope_kv_append_rows + grit_attn_decode_rows) against the token-by-token loop: the SAME bits -- the logits of every
    generated position, the tokens, and through them the K/V the chunk appended -- for sequences with different prefix lengths, row
    counts that are not a multiple of 8, and both decode arithmetics; and against the fp32 oracle for one sequence."""
    from gritlm_amd.decoder import MistralDecoder
    eng, cfg, w = build_engine(cfg_name, 6)
    eng.set_precision(policy)
    rng = np.random.default_rng(131)
    lm = O.bf16_round((rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"])) * 0.05).astype(np.float32))
    dec = MistralDecoder(eng, torch.from_numpy(lm))
    doc = rng.integers(3, cfg["vocab_size"], size=(B, 70)).astype(np.int64)
    plens = np.array([70, 41, 55, 70][:B], dtype=np.int32)
    dmask = (np.arange(70)[None] < plens[:, None]).astype(np.int64)
    _, kv = eng.forward(torch.from_numpy(doc).to(DEV), torch.from_numpy(dmask).to(DEV), return_kv=True, kv_dtype=None)
    prompt = torch.from_numpy(rng.integers(3, cfg["vocab_size"], size=(B, P)).astype(np.int64)).to(DEV)
    pl = torch.from_numpy(plens).to(DEV)
    out, ok = {}, True
    res = {}
    for chunk in (True, False):
        dec.prompt_chunk = chunk
        toks, lg = dec.generate(prompt, new, past_key_values=kv, past_lens=pl, return_logits=True)
        dec.use_graph = True
        toks_g = dec.generate(prompt, new, past_key_values=kv, past_lens=pl)
        res[chunk] = (toks, lg, toks_g)
    out["logits_identical"] = bool(torch.equal(res[True][1], res[False][1]))
    out["tokens_identical"] = bool(torch.equal(res[True][0], res[False][0])) and bool(torch.equal(res[True][2], res[False][2])) \
        and bool(torch.equal(res[True][0], res[True][2]))
    ok &= out["logits_identical"] and out["tokens_identical"] and dec.last_precision == ("bf16" if policy == "bf16" else "f16")
    # one sequence against the fp32 oracle (its own fp32 document K/V): the chunk path is a correct continuation, not merely a consistent one
    _, kv_ref = O.mistral_encode(w, cfg, doc[:1], dmask[:1], return_layers="kv")
    seq = np.concatenate([prompt[0].cpu().numpy(), res[True][0][0].cpu().numpy()])
    ref = O.mistral_continue(w, cfg, kv_ref, int(plens[0]), seq, lm)[P - 1:P - 1 + new].astype(np.float64)
    got = f32(res[True][1])[0].astype(np.float64)
    out["1-cos_vs_fp32_oracle"] = float(np.max(1.0 - np.sum(got * ref, axis=-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(ref, axis=-1))))
    ok &= out["1-cos_vs_fp32_oracle"] < (2e-4 if policy == "bf16" else 1e-5)
    return _res(f"prompt chunk on cached K/V == token-by-token prompt [{cfg_name}, B={B}, P={P}, {policy}]", bool(ok), **out)


def check_knn_topk(Q=5, N=10000, H=256, k=10, transposed=False):
    """Index search (rag/index.py:97-104: queries @ embeddings, torch.topk) on the f32-MFMA similarity GEMM + chunked bitonic top-k vs
    numpy: same neighbours (scores compared; indices wherever the score has no tie), descending order, both index layouts."""
    rng = np.random.default_rng(101)
    q = rng.standard_normal((Q, H)).astype(np.float32)
    e = rng.standard_normal((N, H)).astype(np.float32)
    e[7] = e[3]                                             # an exact tie: the lower index must come first
    tq = torch.from_numpy(q).to(DEV)
    te = torch.from_numpy(np.ascontiguousarray(e.T) if transposed else e).to(DEV)
    sc, ix = ops.knn_topk(tq, te, k, transposed=transposed)
    sc, ix = f32(sc), ix.cpu().numpy()
    ref = q.astype(np.float64) @ e.astype(np.float64).T
    order = np.argsort(-ref, axis=1, kind="stable")[:, :k]
    ref_sc = np.take_along_axis(ref, order, axis=1)
    ok = float(np.max(np.abs(sc - ref_sc))) < 1e-3 and bool((np.diff(sc, axis=1) <= 0).all())
    gap = np.abs(np.diff(np.sort(ref, axis=1)[:, ::-1][:, :k + 1], axis=1)).min(axis=1) > 1e-4        # rows whose top-(k+1) has no near-tie
    ok &= bool((ix[gap] == order[gap]).all()) and (gap.any() or k == N)      # k == N: the planted tie is in every row
    got_sc = np.take_along_axis(ref, ix, axis=1)
    ok &= float(np.max(np.abs(got_sc - ref_sc))) < 1e-3    # every returned index really has a top-k score
    # tie order: wherever both 3 and 7 are returned, 3 precedes 7
    for r in range(Q):
        a, b = np.where(ix[r] == 3)[0], np.where(ix[r] == 7)[0]
        if a.size and b.size:
            ok &= a[0] < b[0]
    return _res(f"knn_topk[Q={Q},N={N},H={H},k={k},transposed={transposed}]", bool(ok), max_score_err=float(np.max(np.abs(sc - ref_sc))))


def check_rag_distributed_index():
    """gritlm_amd.rag.DistributedIndex on the GPU (grit_knn_topk) against the REFERENCE's rag/index.py on the same saved index
    (tests/golden/rag_index/: three shard files written by the reference's save_index, its search_knn result on four queries): the same
    passages in the same order, scores to 1e-5; in-place column fill as rag/eval.py:145 does it; save -> load round trip on the GPU; a bf16
    index; topk larger than the index."""
    import json
    import tempfile
    from gritlm_amd import rag
    gold = os.path.join(GOLDEN, "rag_index")
    exp = json.load(open(os.path.join(gold, "expected.json")))
    idx = rag.DistributedIndex()
    idx.load_index(gold, exp["shards"])
    ok = idx.is_in_gpu and idx.embeddings.is_cuda and idx.embeddings.shape == (exp["dim"], exp["n_passages"])
    q = torch.tensor(exp["queries"], device=DEV)
    docs, scores = idx.search_knn(q, exp["topk"])
    ok &= [[d["id"] for d in row] for row in docs] == exp["docs_ids"]
    err = float((torch.tensor(scores) - torch.tensor(exp["scores"])).abs().max())
    ok &= err < 1e-5
    # built the way rag/eval.py builds it: zero matrix, encode() output transposed into column blocks
    passages = [p for p in exp["loaded"] if p is not None]
    built = rag.DistributedIndex()
    built.init_embeddings(passages, exp["dim"])
    emb_rows = torch.tensor(exp["embeddings"], device=DEV).t().contiguous()          # [N, dim] as encode_corpus returns it
    built.embeddings[:, 0:4] = emb_rows[0:4].T.to(built.dtype)
    built.embeddings[:, 4:] = emb_rows[4:].T.to(built.dtype)
    docs2, scores2 = built.search_knn(q.cpu(), exp["topk"])                           # host queries are moved to the index
    ok &= [[d["id"] for d in row] for row in docs2] == exp["docs_ids"] and scores2 == scores
    with tempfile.TemporaryDirectory() as td:
        built.save_index(td, 2)
        back = rag.DistributedIndex()
        back.load_index(td, 2)
        ok &= bool(torch.equal(back.embeddings, built.embeddings)) and back.doc_map == built.doc_map
    everything, sc_all = built.search_knn(q, 64)
    ok &= all(len(r) == len(passages) for r in everything) and all(r == sorted(r, reverse=True) for r in sc_all)
    half = rag.DistributedIndex(dtype=torch.bfloat16)
    half.init_embeddings(passages, exp["dim"])
    half.embeddings[:, :] = emb_rows.T.to(torch.bfloat16)
    d16, s16 = half.search_knn(q, 1)
    ref16 = (q @ half.embeddings.float()).cpu()
    ok &= [r[0]["id"] for r in d16] == [passages[int(c)]["id"] for c in ref16.argmax(dim=1)]
    ok &= float((torch.tensor(s16)[:, 0] - ref16.max(dim=1).values).abs().max()) < 1e-5
    return _res("rag_distributed_index", bool(ok), max_score_err_vs_reference=err)


def check_cli_native(arch="mistral"):
    """python -m gritlm.training.run on the GPU: bf16 tiny Mistral (or Mixtral), (instruction, text) rows, GradCache switch, native engine."""
    import json
    import tempfile
    from gritlm.training.run import main
    W = synth.WORDS
    with tempfile.TemporaryDirectory() as td:
        if arch == "mixtral":
            d16 = synth.build_mixtral_dir(os.path.join(td, "m16"), "moe-tiny", 0, "bfloat16")
        else:
            d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        rows = []
        for i in range(0, 64, 2):
            negs = [["w3", " ".join(W[j:j + 7])] for j in range(i + 20, i + 27)]
            rows.append({"query": ["w1 w2", " ".join(W[i:i + 5])], "pos": [["w3", " ".join(W[i + 1:i + 9])]], "neg": negs})
        data = os.path.join(td, "toy.jsonl")
        open(data, "w").write("\n".join(json.dumps(r) for r in rows))
        out = os.path.join(td, "out")
        common = ["--model_name_or_path", d16, "--train_data", data, "--output_dir", out, "--bf16", "--per_device_train_batch_size", "2",
                  "--gradient_accumulation_steps", "4", "--no_gen_gas", "--no_emb_gas", "--train_group_size", "8", "--pooling_method", "mean",
                  "--learning_rate", "2e-4", "--query_max_len", "24", "--passage_max_len", "40", "--report_to", "none", "--logging_steps", "1"]
        l1 = main(common + ["--max_steps", "1"])
        l8 = main(common + ["--max_steps", "8"])
        files = os.listdir(out)
        # --pass1_precision (round 6, not a reference flag): GradCache pass 1 on fp16 operands -- the first step's loss moves by the bf16
        # policy's rep error only, training still converges
        l1h = main(common + ["--max_steps", "1", "--pass1_precision", "f16_operands"])
        l8h = main(common + ["--max_steps", "8", "--pass1_precision", "f16_operands"])
    ok = np.isfinite(l1) and np.isfinite(l8) and l8 < l1 and "config.json" in files
    ok = ok and np.isfinite(l1h) and np.isfinite(l8h) and l8h < l1h and abs(l1h - l1) < 5e-2 * max(1.0, abs(l1))
    return _res(f"CLI gritlm.training.run native [{arch}] (loss decreases over 8 steps; --pass1_precision f16_operands)", ok, loss_step1=float(l1),
                loss_step8=float(l8), loss_step1_f16_pass1=float(l1h), loss_step8_f16_pass1=float(l8h))


def check_cli_unified_native(arch="mistral"):
    """python -m gritlm.training.run --mode unified on the GPU: generative branch (causal kernels + lm_head + CE) and the GradCache
    embedding step both on the native engine; both losses fall over 8 steps."""
    import json
    import tempfile
    from gritlm.training import run
    W = synth.WORDS
    with tempfile.TemporaryDirectory() as td:
        if arch == "mixtral":
            d16 = synth.build_mixtral_dir(os.path.join(td, "m16"), "moe-tiny", 0, "bfloat16")
        else:
            d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        os.makedirs(os.path.join(td, "data"))
        rows = []
        for i in range(0, 64, 2):
            negs = [["w3", " ".join(W[j:j + 7])] for j in range(i + 20, i + 27)]
            rows.append({"query": ["w1 w2", " ".join(W[i:i + 5])], "pos": [["w3", " ".join(W[i + 1:i + 9])]], "neg": negs})
        open(os.path.join(td, "data", "emb.jsonl"), "w").write("\n".join(json.dumps(r) for r in rows))
        gen = [{"text": [" ".join(W[i:i + 4]), " ".join(W[(7 * i) % 300:(7 * i) % 300 + 12])]} for i in range(32)]
        open(os.path.join(td, "data", "gen.jsonl"), "w").write("\n".join(json.dumps(r) for r in gen))
        common = ["--model_name_or_path", d16, "--train_data", os.path.join(td, "data"), "--output_dir", os.path.join(td, "out"), "--bf16",
                  "--mode", "unified", "--per_device_train_batch_size", "2", "--gradient_accumulation_steps", "4", "--no_gen_gas",
                  "--no_emb_gas", "--per_device_generative_bs", "4", "--train_group_size", "8", "--pooling_method", "mean",
                  "--learning_rate", "3e-4", "--query_max_len", "24", "--passage_max_len", "40", "--generative_max_len", "48",
                  "--report_to", "none", "--logging_steps", "1"]
        l1 = run.main(common + ["--max_steps", "1"]); g1 = run.main.last_loss_gen
        l8 = run.main(common + ["--max_steps", "8"]); g8 = run.main.last_loss_gen
    ok = all(np.isfinite(v) for v in (l1, l8, g1, g8)) and l8 < l1 and g8 < g1
    return _res(f"[{arch}] CLI --mode unified native (emb + gen losses decrease over 8 steps)", bool(ok), loss_emb_1=float(l1), loss_emb_8=float(l8),
                loss_gen_1=float(g1), loss_gen_8=float(g8))


def check_packed_encode(cfg_name="gqa", B=5, S=150):
    """Un-padded (packed / varlen) encode == padded encode, bit for bit, and both match the oracle."""
    eng, cfg, w = build_engine(cfg_name, 3)
    ids, mask = synth.make_batch(cfg, B, S, seed=91, min_len=7)
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    instr = torch.tensor([0, 2, 5, 1, 0][:B], dtype=torch.int32, device=DEV)
    ok, out = True, {}
    ok &= MistralEncoderEngine.is_right_padded(tm)
    holes = tm.clone(); holes[1, 3] = 0
    ok &= not MistralEncoderEngine.is_right_padded(holes)
    for method in ("mean", "weightedmean", "cls", "lasttoken"):
        il = instr if "mean" in method else None
        a = f32(eng.encode_pooled(tid, tm, method, True, il, packed=False))
        b = f32(eng.encode_pooled(tid, tm, method, True, il, packed=True))
        out[f"{method}_maxdiff"] = float(np.max(np.abs(a - b)))
        ok &= np.array_equal(a, b)
        ref = O.encode_core(w, cfg, ids, mask, method, True, None if il is None else instr.cpu().numpy())
        c = float(np.max(1 - np.sum(b * ref, axis=1)))
        out[f"{method}_1-cos_oracle"] = c
        ok &= c < 1e-4
    return _res(f"packed (un-padded) encode == padded encode [{cfg_name},B={B},S={S}]", ok, **out)


def check_causal_encode(cfg_name="gqa", B=4, S=150):
    """attn='cc..' embedding models (causal attention + lasttoken / weightedmean pooling): engine.causal vs the oracle, packed == padded."""
    eng, cfg, w = build_engine(cfg_name, 4)
    eng.causal = True
    ids, mask = synth.make_batch(cfg, B, S, seed=93, min_len=17)
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    ok, out = True, {}
    h_ref = O.mistral_encode(w, cfg, ids, mask, causal=True)
    for method in ("lasttoken", "weightedmean", "mean"):
        a = f32(eng.encode_pooled(tid, tm, method, True, packed=False))
        b = f32(eng.encode_pooled(tid, tm, method, True, packed=True))
        ref = O.l2_normalize(O.pooling(h_ref, mask, method))
        c = float(np.max(1 - np.sum(b * ref, axis=1)))
        out[f"{method}_1-cos_oracle"] = c
        ok &= np.array_equal(a, b) and c < 1e-4
    eng.causal = False
    e_bi = f32(eng.encode_pooled(tid, tm, "mean", True))
    ok &= float(np.max(1 - np.sum(e_bi * ref, axis=1))) > 1e-3          # and it differs from the bidirectional embedding
    return _res(f"causal ('cc') embedding encode [{cfg_name}] vs oracle", bool(ok), **out)


def check_sliding_window_encode():
    """Causal attention with Mistral's sliding window, engine vs the REFERENCE's eager path (tests/golden/sliding_window_gqa.npz: window 16
    on sequences of up to 200 tokens; the fixture records how many keys a query saw in the generating run) -- padded and packed."""
    g = np.load(os.path.join(GOLDEN, "sliding_window_gqa.npz"))
    eng, cfg, w = build_engine(str(g["cfg_name"]), int(g["seed_w"]))
    eng.causal, eng.window_keys = True, int(g["window_keys"])
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    valid = mask.astype(bool)
    ref = g["last_hidden_state"]
    rel = lambda a: float(np.linalg.norm((a - ref)[valid]) / np.linalg.norm(ref[valid]))
    h_pad = f32(eng.forward(tid, tm))
    eng.window_keys = 0
    h_full = f32(eng.forward(tid, tm))
    eng.window_keys = int(g["window_keys"])
    a = f32(eng.encode_pooled(tid, tm, "mean", True, packed=False))
    b = f32(eng.encode_pooled(tid, tm, "mean", True, packed=True))
    emb_ref = O.l2_normalize(O.pooling(ref, mask, "mean"))
    cos = float(np.max(1 - np.sum(b * emb_ref, axis=1)))
    out = dict(rel_l2_vs_reference=rel(h_pad), rel_l2_without_window=rel(h_full), pooled_1_minus_cos=cos, window_keys=int(g["window_keys"]))
    ok = rel(h_pad) < 2e-2 and rel(h_full) > 10 * rel(h_pad) and np.array_equal(a, b) and cos < 1e-4
    return _res("sliding-window causal encode vs reference (eager path)", bool(ok), **out)


def check_generative_window(window=9):
    """The generative branch under a sliding window (train engine: causal attention fwd + bwd with ``window_keys``): loss vs the oracle
    with the same window, packed == padded, and it differs from the loss without a window."""
    import tempfile
    from gritlm_amd.training import GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, "generative_tiny.npz"))
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, 0)
    h = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], causal=True, window=window)
    ref_loss = O.next_token_loss(h @ g["lm_head"].T, g["labels"], "mixed", float(g["factor_mixed"]))
    out, ok, grads = dict(loss_oracle=ref_loss, loss_no_window_reference=float(g["loss_gen_mixed"])), True, {}
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLMTrainModel(model_name_or_path=d16, mode="unified", pooling_method="mean", normalized=True, attn="bbcc", temperature=0.02,
                             negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16, loss_gen_type="mixed",
                             loss_gen_factor=float(g["factor_mixed"]))
        m.enable_native()
        m.train_engine.window_keys = window
        for packed in (True, False):
            m.native_packed = packed
            m.model.zero_grad(set_to_none=True)
            gen = {"input_ids": torch.from_numpy(g["input_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["attention_mask"]).to(DEV),
                   "labels": torch.from_numpy(g["labels"]).to(DEV)}
            o = m(generative=gen)
            o.loss_gen.backward()
            tag = "packed" if packed else "padded"
            out[f"loss_{tag}"] = float(o.loss_gen.item())
            ok &= abs(out[f"loss_{tag}"] - ref_loss) < 1e-2 * max(1.0, abs(ref_loss))
            grads[tag] = {n: p.grad.detach().float().clone() for n, p in m.model.named_parameters() if p.grad is not None}
        worst = max(float((grads["packed"][n] - grads["padded"][n]).norm() / (grads["padded"][n].norm() + 1e-20)) for n in grads["padded"])
        out["worst_grad_rel_packed_vs_padded"] = worst
        ok &= worst < 1e-2 and abs(ref_loss - float(g["loss_gen_mixed"])) > 1e-3
        ok &= all(bool(torch.isfinite(t).all()) for t in grads["packed"].values())
    return _res(f"generative step under a sliding window of {window} keys", bool(ok), **out)


def check_generate_native_api():
    """GritLM.generate_native: the reference's RAG call (rag/eval.py:277-302: ``model.generate(**inputs, past_key_values=kv_cache,
    min_new_tokens=, max_new_tokens=, pad_token_id=)`` with a mask that spans cache + inputs) on the native decoder, returned like Hugging
    Face's generate: [inputs | new tokens]; greedy tokens equal Hugging Face's own generate() on the same cache wherever its fp32 margin is
    clear; EOS handling (tokens after a row's EOS are pad, the common tail is trimmed); unsupported options raise."""
    import tempfile
    from gritlm_amd import GritLM
    g = np.load(os.path.join(GOLDEN, "gritlm_encode.npz"))
    sents = [str(x) for x in g["sentences"]][:2]
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, mode="unified")
        _, cache = m.encode(sents[:1], max_length=48, get_cache=True)
        get = lambda c, li: (c.layers[li].keys, c.layers[li].values) if hasattr(c, "layers") else (c[li][0], c[li][1])
        n_layers = len(cache.layers) if hasattr(cache, "layers") else len(cache)
        kv = [(get(cache, li)[0].clone(), get(cache, li)[1].clone()) for li in range(n_layers)]
        S0 = kv[0][0].shape[2]
        q = m.tokenizer([" ".join(synth.WORDS[5:12])], return_tensors="pt", add_special_tokens=False)["input_ids"].to(DEV)
        P = q.shape[1]
        mask = torch.ones((1, S0 + P), dtype=torch.long, device=DEV)            # the reference's mask: ones over the cache, then the inputs' mask
        full = m.generate_native(input_ids=q, attention_mask=mask, past_key_values=kv, max_new_tokens=6, min_new_tokens=6, pad_token_id=0)
        ok &= tuple(full.shape) == (1, P + 6) and bool(torch.equal(full[:, :P], q))
        direct = m.native_decoder().generate(q, 6, past_key_values=kv)
        ok &= bool(torch.equal(full[:, P:], direct))
        # Hugging Face's generate() on the same cache (what the reference runs): same greedy tokens where the fp32 margin is clear
        from transformers import DynamicCache
        pc = DynamicCache()
        for li, (k_, v_) in enumerate(kv):
            pc.update(k_.clone(), v_.clone(), li)
        hf = m.generate(input_ids=torch.cat([torch.zeros((1, S0), dtype=torch.long, device=DEV), q], dim=1), attention_mask=mask, past_key_values=pc,
                        max_new_tokens=6, min_new_tokens=6, do_sample=False, pad_token_id=0)[0, S0 + P:]
        out["tokens_equal_hf_generate"] = int((hf == full[0, P:]).sum())
        ok &= out["tokens_equal_hf_generate"] >= 4                     # (bf16 near-ties on a random tiny model may flip a late token)
        # EOS: make the second generated token the EOS -> the rest of the row is pad and the tail is trimmed to the EOS
        eos = int(full[0, P + 1])
        first = int(full[0, P])
        cut = m.generate_native(input_ids=q, attention_mask=mask, past_key_values=kv, max_new_tokens=6, eos_token_id=eos, pad_token_id=0)
        want = [first, eos] if first != eos else [eos]
        ok &= cut[0, P:].tolist() == want
        out["eos_trimmed_len"] = int(cut.shape[1] - P)
        # a long budget: the decoder stops replaying once every row has emitted EOS (checked every 16 tokens); same result
        cut40 = m.generate_native(input_ids=q, attention_mask=mask, past_key_values=kv, max_new_tokens=40, eos_token_id=eos, pad_token_id=0)
        ok &= bool(torch.equal(cut40, cut))
        raw = m.native_decoder().generate(q, 40, past_key_values=kv, eos_token_id=eos)
        ok &= tuple(raw.shape) == (1, 40) and raw[0, :len(want)].tolist() == want and bool((raw[0, len(want):] == eos).all())
        # a plain prompt (no cache)
        plain = m.generate_native(input_ids=q, max_new_tokens=3, min_new_tokens=3)
        ok &= tuple(plain.shape) == (1, P + 3)
        for bad in (dict(do_sample=True), dict(num_beams=2), dict(min_new_tokens=2)):
            try:
                m.generate_native(input_ids=q, max_new_tokens=4, **bad); ok = False
            except NotImplementedError:
                pass
    return _res("GritLM.generate_native: the reference's cached-generation call on the native decoder", bool(ok), **out)


def check_edge_cases():
    """Empty / minimal / degenerate inputs the host can hand over (reference behaviour noted per case)."""
    ok, notes = True, {}
    eng, cfg, w = build_engine("tiny", 0)
    # one document, one token
    ids = np.array([[7]], dtype=np.int64); mask = np.ones((1, 1), dtype=np.int64)
    e = f32(eng.encode_pooled(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV), "mean", True))
    ref = O.encode_core(w, cfg, ids, mask, "mean", True)
    notes["1x1_1-cos"] = float(1 - np.sum(e * ref)); ok &= notes["1x1_1-cos"] < 1e-4
    # empty batch: zero rows in, zero rows out, no launch failure
    z = ops.rmsnorm(torch.empty((0, 256), dtype=torch.bfloat16, device=DEV), torch.ones(256, dtype=torch.bfloat16, device=DEV), 1e-5)
    ok &= z.shape == (0, 256)
    ok &= ops.gemm_nt(torch.empty((0, 64), dtype=torch.bfloat16, device=DEV), torch.zeros((16, 64), dtype=torch.bfloat16, device=DEV)).shape == (0, 16)
    ok &= ops.pool_norm(torch.empty((0, 4, 64), dtype=torch.bfloat16, device=DEV), torch.empty((0, 4), dtype=torch.int64, device=DEV), "mean", True).shape == (0, 64)
    # an all-masked pooling row divides by zero exactly like the unguarded reference (gritlm.py:213-214): NaN, other rows intact
    hid = rnd((2, 6, 64), 3); m = np.array([[1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0]], dtype=np.int64)
    out = f32(ops.pool_norm(bf(hid), torch.from_numpy(m).to(DEV), "mean", False))
    with np.errstate(all="ignore"):
        refp = O.pooling(hid, m, "mean")
    ok &= np.allclose(out[0], refp[0], atol=1e-6) and np.isnan(out[1]).all() and np.isnan(refp[1]).all()
    # unsupported shapes are refused loudly, never computed wrongly
    from gritlm_amd._lib import GritHipError
    for fn in (lambda: ops.gemm_nt(torch.zeros((4, 48), dtype=torch.bfloat16, device=DEV), torch.zeros((16, 48), dtype=torch.bfloat16, device=DEV)),
               lambda: ops.attn_bidir(torch.zeros((8, 3 * 64), dtype=torch.bfloat16, device=DEV), torch.ones((1, 1), dtype=torch.int64, device=DEV), 1, 8, 1, 1, 64)):
        try:
            fn(); ok = False
        except GritHipError:
            pass
    return _res("edge cases (1 token, empty batch, all-masked row, unsupported shapes)", ok, **notes)


def check_long_sequence(S=4096):
    """Maximum sequence length the reference evaluates with (rag/eval.py:283 tokenises up to 4096): flash attention vs oracle."""
    return check_attention(B=1, S=S, nq=2, nkv=1, mask_kind="ragged_long", seed=5)


def check_full_shape_properties(B=24, S=512):
    """Size-independent properties at the true GritLM-7B LAYER shape (H 4096, 32/8 heads, I 14336; 2 layers) where the oracle is
    too slow for a direct comparison: (1) documents are independent -- permuting the batch permutes the embeddings bit for bit;
    (2) padding invariance -- appending pad tokens never changes an embedding (packed == padded, bitwise);
    (3) unit norm; (4) a document's embedding does not depend on its batch mates."""
    cfg = EncoderConfig.from_dict(dict(synth.CONFIGS["7b"], num_hidden_layers=2))
    eng = MistralEncoderEngine.random_init(cfg, DEV, seed=11)
    g = torch.Generator(device=DEV).manual_seed(5)
    ids = torch.randint(3, cfg.vocab_size, (B, S), generator=g, device=DEV)
    lens = torch.randint(40, S + 1, (B,), generator=g, device=DEV); lens[0] = S
    mask = (torch.arange(S, device=DEV).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
    e = eng.encode_pooled(ids, mask, "mean", True, packed=True)
    perm = torch.randperm(B, generator=g, device=DEV)
    e_perm = eng.encode_pooled(ids[perm], mask[perm], "mean", True, packed=True)
    e_pad = eng.encode_pooled(ids, mask, "mean", True, packed=False)
    e_sub = eng.encode_pooled(ids[:5], mask[:5], "mean", True, packed=True)
    norms = e.norm(dim=1)
    ok = torch.equal(e[perm], e_perm) and torch.equal(e, e_pad) and torch.equal(e[:5], e_sub)
    ok &= bool(((norms - 1).abs() < 1e-5).all()) and bool(torch.isfinite(e).all())
    return _res(f"full layer shape properties [B={B},S={S},H=4096,L=2]", ok, max_norm_dev=float((norms - 1).abs().max()))


def check_get_cache():
    """encode(get_cache=True): embeddings + per-layer KV of the bidirectional pass, vs the Hugging Face module on the same GPU
    (the reference path: gritlm/gritlm.py:131-140 hands back outputs[1]).
    The comparison target is the stock Hugging Face module run on THIS GPU, not a CPU fixture: stock transformers.MistralModel with a
    bidirectional mask is bit-equal to the reference on CPU (tests/test_oracle_golden.py::test_torch_reference_equals_reference_fixture),
    and the cache it returns for use_cache=True is by construction what the reference returns."""
    import tempfile
    from gritlm_amd import GritLM
    g = np.load(os.path.join(GOLDEN, "gritlm_encode.npz"))
    sents = [str(x) for x in g["sentences"]][:5]
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16)
        emb, cache = m.encode(sents, max_length=48, get_cache=True)
        hf = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, native=False)
        emb_hf, cache_hf = hf.encode(sents, max_length=48, get_cache=True)
    ok = hf.engine is None and m.engine is not None
    one_minus_cos = float(np.max(1 - np.sum(emb * emb_hf, axis=1)))
    ok &= one_minus_cos < 1e-4
    worst = 0.0
    get = lambda c, li: (c.layers[li].keys, c.layers[li].values) if hasattr(c, "layers") else (c[li][0], c[li][1])
    for li in range(2):
        (k1, v1), (k2, v2) = get(cache, li), get(cache_hf, li)
        ok &= tuple(k1.shape) == tuple(k2.shape)
        for a, b in ((k1, k2), (v1, v2)):
            e = float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))
            worst = max(worst, e)
    ok &= worst < 3e-2
    # the RAG flow end to end on the drop-in: document cache -> native greedy continuation vs Hugging Face generate() on the same cache
    # (first-position logits; later tokens of a random-init model are decided by near-ties)
    with torch.no_grad():
        from transformers import DynamicCache
        q = m.tokenizer([" ".join(synth.WORDS[5:12])], return_tensors="pt", add_special_tokens=False)["input_ids"].to(DEV)
        one = [(get(cache, li)[0][:1], get(cache, li)[1][:1]) for li in range(2)]
        plen = int(m.tokenizer([sents[0]], return_tensors="pt", truncation=True, max_length=48)["input_ids"].shape[1])
        one = [(k[:, :, :plen].contiguous(), v[:, :, :plen].contiguous()) for k, v in one]
        toks, lg = m.native_decoder().generate(q, 3, past_key_values=one, return_logits=True)
        pc = DynamicCache()
        for li, (k, v) in enumerate(one):
            pc.update(k.clone(), v.clone(), li)
        hf_lg = m.model(input_ids=q, past_key_values=pc, attention_mask=torch.ones((1, plen + q.shape[1]), dtype=torch.long, device=DEV),
                        position_ids=torch.arange(plen, plen + q.shape[1], device=DEV).unsqueeze(0)).logits[0, -1].float()
    dl = float((lg[0, 0].float() - hf_lg).abs().max())
    ok &= dl < 0.05 * float(hf_lg.std()) + 2e-2 and tuple(toks.shape) == (1, 3)
    return _res("encode(get_cache=True): native KV vs Hugging Face KV; native decode from the cache vs HF", ok, emb_1_minus_cos=one_minus_cos,
                kv_max_rel=worst, decode_logit_abs_vs_hf=dl, hf_logit_std=float(hf_lg.std()))


def _overlap_worker(rank, world, port, model_dir, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # gloo all_reduce accepts device tensors: enough for this path
    try:
        torch.cuda.set_device(0)
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        bq, G = 2, int(g["group"])
        sq, sp = slice(rank * bq, (rank + 1) * bq), slice(rank * bq * G, (rank + 1) * bq * G)
        q = {"input_ids": torch.from_numpy(g["q_ids"][sq]).cuda(), "attention_mask": torch.from_numpy(g["q_mask"][sq]).cuda()}
        p = {"input_ids": torch.from_numpy(g["p_ids"][sp]).cuda(), "attention_mask": torch.from_numpy(g["p_mask"][sp]).cuda()}
        GradCacheStep(m, chunk_size=2)(q, p, sync=(world > 1))
        sd = dict(m._backbone().named_parameters())
        ret[(world, rank)] = {n: sd[n].grad.float().cpu().numpy() for n in ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight",
                                                                            "norm.weight", "embed_tokens.weight")}
    finally:
        dist.destroy_process_group()


def check_overlapped_grad_sync():
    """Data-parallel gradient averaging overlapped with the last chunk's backward (OverlappedGradSync + engine.backward callback):
    2 processes on this GPU, each with half of the batch (local negatives), must end with the mean of the two single-process grads."""
    import socket
    import tempfile
    import torch.multiprocessing as mp
    def port():
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); p_ = s_.getsockname()[1]; s_.close(); return p_
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_overlap_worker, args=(2, port(), d16, ret), nprocs=2, join=True)
        solo = mgr.dict()
        for r in range(2):        # the same two half-batches, no averaging: world "1" runs that reuse the rank's slice
            mp.spawn(_solo_worker, args=(r, port(), d16, solo), nprocs=1, join=True)
    worst, ok = 0.0, True
    for n in ret[(2, 0)]:
        mean = 0.5 * (solo[0][n] + solo[1][n])
        for r in range(2):
            e = float(np.linalg.norm(ret[(2, r)][n] - mean) / (np.linalg.norm(mean) + 1e-20))
            worst = max(worst, e)
        ok &= np.array_equal(ret[(2, 0)][n], ret[(2, 1)][n])
    return _res("overlapped data-parallel gradient sync (2 processes)", ok and worst < 2e-2, worst_rel=worst)


def _solo_worker(_, rank, port, model_dir, ret):
    tmp = {}
    _overlap_worker_solo(rank, port, model_dir, tmp)
    ret[rank] = tmp["g"]


def _overlap_worker_solo(rank, port, model_dir, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        torch.cuda.set_device(0)
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        bq, G = 2, int(g["group"])
        sq, sp = slice(rank * bq, (rank + 1) * bq), slice(rank * bq * G, (rank + 1) * bq * G)
        q = {"input_ids": torch.from_numpy(g["q_ids"][sq]).cuda(), "attention_mask": torch.from_numpy(g["q_mask"][sq]).cuda()}
        p = {"input_ids": torch.from_numpy(g["p_ids"][sp]).cuda(), "attention_mask": torch.from_numpy(g["p_mask"][sp]).cuda()}
        GradCacheStep(m, chunk_size=2)(q, p, sync=False)
        sd = dict(m._backbone().named_parameters())
        out["g"] = {n: sd[n].grad.float().cpu().numpy() for n in ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight",
                                                                  "norm.weight", "embed_tokens.weight")}
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the fp16-operand policies beyond dense inference (sparse-MoE engine, GradCache pass 1, get_cache, the "auto" ladder)
def check_gemm_f16_stacked(M=300, I=512, K=256):
    """grit_gemm_f16_nt with GRIT_EPI_SWIGLU_STACKED ([gate; up] weights: the training engine's layout, used by GradCache pass 1 under an
    fp16 policy) must be bit-identical to GRIT_EPI_SWIGLU on the pre-interleaved copy of the same fp16 weights."""
    from gritlm_amd._lib import EPI_SWIGLU_STACKED
    a, wg, wu = fh(f16r(rnd((M, K), 3))), fh(f16r(rnd((I, K), 4, 0.05))), fh(f16r(rnd((I, K), 5, 0.05)))
    _f16_flag()
    ref = ops.gemm_nt(a, swiglu_interleave(wg, wu), epilogue=EPI_SWIGLU)
    got = ops.gemm_nt(a, torch.cat([wg, wu], dim=0).contiguous(), epilogue=EPI_SWIGLU_STACKED)
    ok = bool(torch.equal(ref, got)) and ref.dtype == torch.float16 and not _f16_flag()
    return _res(f"f16 swiglu stacked == interleaved [M={M},I={I},K={K}]", ok, max_abs=float((ref.float() - got.float()).abs().max()))


def check_gemm_f16_grouped(counts=(300, 0, 17, 256, 513, 1, 0, 64), N=384, K=256, epi=EPI_STORE, seed=165):
    """grit_gemm_f16_nt_grouped (device-side counts, gathered A rows) vs fp64 products of the same fp16 operands per group: ONE fp16
    rounding of the accumulator (SWIGLU / SWIGLU_STACKED: of silu(gate) * up evaluated in fp32); empty and tiny groups included; the
    stacked form equals the interleaved form bit for bit; un-gathered == gathered on pre-permuted rows."""
    from gritlm_amd._lib import EPI_SWIGLU_STACKED
    E, M = len(counts), int(sum(counts))
    Tsrc = M // 2 + 5
    a = f16r(rnd((Tsrc, K), seed))
    w = f16r(rnd((E, N, K), seed + 1, 0.05))
    rng = np.random.default_rng(seed + 2)
    a_rows = rng.integers(0, Tsrc, size=M).astype(np.int32)
    tcounts = torch.tensor(counts, dtype=torch.int32, device=DEV)
    ta, trows = fh(a), torch.from_numpy(a_rows).to(DEV)
    _f16_flag()
    ok = True
    if epi == EPI_SWIGLU:
        I = N // 2
        wi = torch.stack([swiglu_interleave(fh(w[e, :I]), fh(w[e, I:])) for e in range(E)]).contiguous()
        out_t = ops.gemm_nt_grouped(ta, wi, tcounts, M, epilogue=epi, a_rows=trows)
        out_s = ops.gemm_nt_grouped(ta, fh(w), tcounts, M, epilogue=EPI_SWIGLU_STACKED, a_rows=trows)
        ok &= bool(torch.equal(out_t, out_s))
        wop = wi
    else:
        wop = fh(w)
        out_t = ops.gemm_nt_grouped(ta, wop, tcounts, M, a_rows=trows)
    ok &= out_t.dtype == torch.float16
    out = out_t.float().cpu().numpy().astype(np.float64)
    ref = np.zeros(out.shape, dtype=np.float64)
    off = 0
    for e, c in enumerate(counts):
        full = a[a_rows[off:off + c]].astype(np.float64) @ w[e].astype(np.float64).T
        if epi == EPI_SWIGLU:
            g, u = full[:, :N // 2], full[:, N // 2:]
            full = g / (1.0 + np.exp(-g)) * u
        ref[off:off + c] = full
        off += c
    scale = float(np.sqrt(np.mean(ref ** 2))) + 1e-12
    err = float(np.max(np.abs(out - ref) / (2.0 ** -11 * np.abs(ref) + 6.0e-8 + 2e-5 * scale)))
    out2 = ops.gemm_nt_grouped(fh(a[a_rows]), wop, tcounts, M, epilogue=epi)
    flag = _f16_flag()
    ok &= err < 1.0 and bool(torch.equal(out_t, out2)) and not flag
    return _res(f"gemm_f16_grouped[counts={list(counts)},N={N},K={K},epi={epi}]", bool(ok), max_err_over_tol=err, overflow_flag=flag)


def check_moe_router_f32(T=777, H=512, E=8, eps=1e-5):
    """grit_moe_router_top2_f32 (the routing of the f16_operands policy: fp32 residual stream in, the post-attention RMSNorm folded in,
    nothing rounded) against the reference's arithmetic in fp64 (scripts/modeling_mixtral_gritlm.py:843-849 on RMSNorm(h)): the same two
    experts for every token whose 2nd and 3rd logits are further apart than fp32 summation noise, routing weights within 2e-6 + 2.5e-7 x the
    largest logit (fp32 summation noise of the H-term dot products); index
    (counts, stable sort, inverse map) exact."""
    rng = np.random.default_rng(261)
    h = (rng.standard_normal((T, H)) * 3.0).astype(np.float32)
    h[5] *= 1e3; h[6] *= 1e-3                                             # rows of very different scale: the norm is folded in
    lnw = O.bf16_round(1.0 + 0.1 * rng.standard_normal(H).astype(np.float32))
    gw = O.bf16_round(rng.standard_normal((E, H)).astype(np.float32) * 0.5)
    th = torch.from_numpy(h).to(DEV)
    experts, weights, counts, row_token, rows = ops.moe_route_f32(th, bf(lnw), eps, bf(gw))
    h64 = h.astype(np.float64)
    xn = lnw.astype(np.float64) * (h64 / np.sqrt(np.mean(h64 ** 2, axis=1, keepdims=True) + eps))
    logits = xn @ gw.astype(np.float64).T
    pr = np.exp(logits - logits.max(1, keepdims=True)); pr /= pr.sum(1, keepdims=True)
    order = np.argsort(-pr, axis=1, kind="stable")
    sel_ref = order[:, :2]
    w_ref = np.take_along_axis(pr, sel_ref, axis=1); w_ref /= w_ref.sum(1, keepdims=True)
    srt = np.sort(logits, axis=1)[:, ::-1]
    clear = (srt[:, 1] - srt[:, 2] > 1e-3) & (srt[:, 0] - srt[:, 1] > 1e-3)
    sel, wt = experts.cpu().numpy(), weights.cpu().numpy().astype(np.float64)
    same = (sel == sel_ref).all(1)
    werr = float(np.max(np.abs(wt - w_ref)[same])) if same.any() else 1.0
    # fp32 dot products of H terms carry ~1e-7 * |logit| of summation noise (the reference's fp32 matmul does too), which the softmax
    # passes on to the weights scaled by p (1 - p) <= 1/4
    wtol = 2e-6 + 2.5e-7 * float(np.abs(logits).max())
    ok = bool(same[clear].all()) and clear.mean() > 0.99 and werr < wtol
    flat = sel.reshape(-1)
    ok &= np.array_equal(counts.cpu().numpy(), np.bincount(flat, minlength=E))
    o2 = np.argsort(flat, kind="stable")
    ok &= np.array_equal(row_token.cpu().numpy(), (o2 // 2).astype(np.int32))
    inv = np.empty_like(o2); inv[o2] = np.arange(o2.size)
    ok &= np.array_equal(rows.cpu().numpy().reshape(-1), inv.astype(np.int32))
    return _res(f"moe router fp32 stream [T={T},H={H},E={E}]", bool(ok), clear_frac=float(clear.mean()), agree_all=float(same.mean()), max_weight_err=werr)


def check_moe_combine_f32(T=333, H=512):
    """grit_moe_combine_f32: out = residual + w0 y[r0] + w1 y[r1] in fp32 (y fp16) vs fp64; in place on the residual; residual = NULL."""
    rng = np.random.default_rng(271)
    y = f16r(rng.standard_normal((2 * T, H)).astype(np.float32))
    res = rng.standard_normal((T, H)).astype(np.float32) * 5.0
    perm = rng.permutation(2 * T).astype(np.int32).reshape(T, 2)
    wts = rng.random((T, 2)).astype(np.float32)
    ty, tr, tw = fh(y), torch.from_numpy(perm).to(DEV), torch.from_numpy(wts).to(DEV)
    th = torch.from_numpy(res).to(DEV)
    ops.moe_combine(ty, tr, tw, th, out=th)
    ref = res.astype(np.float64) + wts[:, :1].astype(np.float64) * y[perm[:, 0]] + wts[:, 1:].astype(np.float64) * y[perm[:, 1]]
    e1 = float(np.max(np.abs(th.cpu().numpy() - ref)) / np.sqrt(np.mean(ref ** 2)))
    o2 = ops.moe_combine(ty, tr, tw, None)
    ref2 = ref - res
    e2 = float(np.max(np.abs(o2.cpu().numpy() - ref2)) / np.sqrt(np.mean(ref2 ** 2)))
    return _res(f"moe combine fp32 [T={T},H={H}]", e1 < 2e-6 and e2 < 2e-6 and o2.dtype == torch.float32, rel_err=e1, rel_err_no_residual=e2)


def _final_norm64(stream: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """final RMSNorm (scripts/modeling_mistral_gritlm.py:84-89) of a residual stream in fp64: takes the bf16 rounding of last_hidden_state
    (the pooling kernels' input format, one 2^-9 rounding per element = ~2e-3 of a row's norm) out of a hidden-state comparison"""
    x = stream.astype(np.float64)
    return w.astype(np.float64) * (x / np.sqrt(np.mean(x * x, axis=-1, keepdims=True) + eps))


def check_mixtral_f16_operands(cfg_name="moe-tiny"):
    """The sparse-MoE engine under precision='f16_operands' (fp32 stream, fp32 routing on the stream, fp16 expert GEMMs, fp32 combine)
    against the fixture the REFERENCE's modeling_mixtral_gritlm.py produced in fp32: EVERY token takes the fp32 reference's experts in
    every layer wherever the reference's own 2nd-vs-3rd margin is not a tie, hidden states at least 10x closer than the bf16 policy,
    embeddings within 1e-5 (north-star 1e-4), packed == padded bit for bit, no overflow; 'f16_stream' / 'fp32_residual' are refused."""
    from gritlm_amd._lib import GritHipError
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    cfg = cfg if isinstance(cfg, dict) else synth.CONFIGS[cfg_name]
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    valid = mask.astype(bool)
    ref32 = g["last_hidden_state"]
    rel = lambda a, b: float(np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid]))
    r_b = rel(f32(eng.forward(tid, tm)), ref32)
    eng.set_precision("f16_operands")
    _f16_flag()
    eng.record_routing = []
    h = f32(eng.forward(tid, tm))
    routing = np.sort(np.stack([r.cpu().numpy() for r in eng.record_routing]).reshape(len(eng.layers), *ids.shape, 2), axis=-1)
    eng.record_routing = None
    agree = float((routing == np.sort(g["routing"], axis=-1)).all(-1)[:, valid].mean())
    r_f = rel(h, ref32)
    # the same comparison without last_hidden_state's bf16 rounding: the engine's fp32 stream through the final norm in fp64
    hs = eng.forward(tid, tm, final_norm=False).cpu().numpy()
    r_s = rel(_final_norm64(hs, w["norm.weight"], cfg["rms_norm_eps"]).astype(np.float32), ref32)
    out = dict(rel_hidden_bf16=r_b, rel_hidden_f16_operands=r_f, rel_hidden_f16_operands_fp32_stream=r_s, routing_agree_with_fp32_ref=agree,
               routing_agree_of_bf16_ref=_yard()[f"encoder_{cfg_name}/routing_agree_of_bf16_ref"])
    # (last_hidden_state is bf16 in every policy: its own rounding is ~2e-3 of a row; the stream itself must be >= 10x closer than bf16's)
    ok = agree >= 0.999 and r_f < max(0.3 * r_b, 3e-3) and r_s < 0.1 * r_b and not np.isnan(h).any()
    for method in ("mean", "weightedmean"):
        e = eng.encode_pooled(tid, tm, method, True, packed=False)
        ep = eng.encode_pooled(tid, tm, method, True, packed=True)
        c32 = float(np.max(1 - np.sum(f32(e).astype(np.float64) * g[f"emb_{method}"].astype(np.float64), axis=1)))
        out[f"{method}_1-cos"] = c32
        ok &= c32 < 1e-5 and bool(torch.equal(e, ep))
    out["overflow_flag"] = _f16_flag()
    ok &= not out["overflow_flag"] and eng.f16_weight_stats["overflow"] == 0
    for pol in ("f16_stream", "fp32_residual"):
        try:
            eng.set_precision(pol); out[pol + "_refused"] = False
        except GritHipError:
            out[pol + "_refused"] = True
        ok &= out[pol + "_refused"]
    ok &= eng.supported_precisions() == ("f16_operands", "bf16")
    return _res(f"mixtral encoder[{cfg_name}] f16_operands policy vs reference fp32", bool(ok), **out)


def check_mixtral_layer_true_shape_f16():
    """ONE layer at the TRUE Mixtral-8x7B layer shape under precision='f16_operands' vs the reference-generated fixture
    (tests/golden/encoder_8x7b-l1.npz), held to the criteria the bf16 engine's check was FIRST written with and
    to the north-star's embedding tolerance: (1) every valid token whose router margin in the fp32 run exceeds 1e-2 takes the fp32
    reference's two experts, overall agreement >= 0.999 (the reference's own bf16 run: 0.984); (2) probe rows that took the reference's
    experts: relative l2 error of all rows together < 3.5e-2 (the reference's own bf16 run: 2.4e-2), per-row median < 3e-3 -- this is the
    FORMAT of last_hidden_state (bf16 in every policy: one 2^-9 rounding per element = ~2e-3 of a row); the engine's fp32 stream pushed
    through the final norm in fp64 is held to a per-row median < 1.75e-3 = 1/8 of the reference's own bf16 run (1.39e-2: three more
    mantissa bits; measured 1.15e-3) and all rows together < 1e-2 (measured 5.2e-3; bf16 reference 2.4e-2); (3) pooled embeddings within
    1e-4 of the fp32 reference's (measured 3.7e-6 / 5.8e-6; the reference's bf16 run: 6.0e-4 / 7.7e-4);
    (4) packed == padded bit for bit; no overflow."""
    g = np.load(os.path.join(GOLDEN, "encoder_8x7b-l1.npz"))
    eng, cfg, w = build_engine("8x7b-l1", int(g["seed_w"]))
    norm_w = w["norm.weight"].copy()
    del w
    eng.set_precision("f16_operands")
    _f16_flag()
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    eng.record_routing = []
    h = f32(eng.forward(tid, tm))
    routing = np.sort(eng.record_routing[0].cpu().numpy().reshape(-1, 2), axis=-1)
    eng.record_routing = None
    valid = mask.astype(bool).reshape(-1)
    r32 = np.sort(g["routing"], axis=-1).reshape(-1, 2)
    agree_tok = (routing == r32).all(-1)
    margin = g["router_margin_2nd_vs_3rd"].reshape(-1)
    clear = valid & (margin > 1e-2)
    out = dict(routing_agree=float(agree_tok[valid].mean()), routing_agree_of_bf16_ref=float((np.sort(g["routing_bf16"], -1).reshape(-1, 2) == r32).all(-1)[valid].mean()),
               clear_margin_tokens=int(clear.sum()), clear_margin_agree=float(agree_tok[clear].mean()),
               smallest_margin_of_a_disagreeing_token=float(margin[valid & ~agree_tok].max()) if (valid & ~agree_tok).any() else 0.0)
    ok = out["clear_margin_agree"] == 1.0 and out["routing_agree"] >= 0.999 and not np.isnan(h).any()
    probe = g["probe_rows"]
    pa = agree_tok[probe]
    hp = h.reshape(-1, h.shape[-1])[probe]
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    out["probe_rows_same_experts"] = int(pa.sum())
    out["rel_ours_vs_fp32"] = rel(hp[pa], g["probe_hidden"][pa]); out["rel_refbf16_vs_fp32"] = rel(g["probe_hidden_bf16"][pa], g["probe_hidden"][pa])
    pr = np.linalg.norm(hp[pa] - g["probe_hidden"][pa], axis=1) / np.linalg.norm(g["probe_hidden"][pa], axis=1)
    out["row_rel_median"], out["row_rel_max"] = float(np.median(pr)), float(pr.max())
    hs = eng.forward(tid, tm, final_norm=False).cpu().numpy()
    hs = _final_norm64(hs.reshape(-1, hs.shape[-1])[probe], norm_w, synth.CONFIGS["8x7b-l1"]["rms_norm_eps"])
    ps = np.linalg.norm(hs[pa] - g["probe_hidden"][pa], axis=1) / np.linalg.norm(g["probe_hidden"][pa], axis=1)
    out["fp32_stream_row_rel_median"], out["fp32_stream_row_rel_max"] = float(np.median(ps)), float(ps.max())
    out["fp32_stream_rel_all_rows"] = rel(hs[pa], g["probe_hidden"][pa])
    ok &= pa.sum() >= 60 and out["rel_ours_vs_fp32"] < 3.5e-2 and out["row_rel_median"] < 3e-3 and out["fp32_stream_row_rel_median"] < 1.75e-3 \
        and out["fp32_stream_rel_all_rows"] < 1e-2
    for method in ("mean", "weightedmean"):
        e = eng.encode_pooled(tid, tm, method, True, packed=False)
        ep = eng.encode_pooled(tid, tm, method, True, packed=True)
        c32 = float(np.max(1 - np.sum(f32(e).astype(np.float64) * g[f"emb_{method}"].astype(np.float64), axis=1)))
        out[f"{method}_1-cos"] = c32
        out[f"{method}_1-cos_of_bf16ref"] = float(np.max(1 - np.sum(g[f"emb_{method}_bf16"] * g[f"emb_{method}"], axis=1)))
        ok &= c32 < 1e-4 and bool(torch.equal(e, ep))
    out["overflow_flag"] = _f16_flag()
    ok &= not out["overflow_flag"]
    return _res("mixtral layer at the true 8x7B shape, f16_operands, ORIGINAL criteria (margin 1e-2, aggregate 3.5e-2) + embeddings 1e-4", bool(ok), **out)


def _train_model_on(bb, cfg, tau=0.02):
    from gritlm_amd.training.engine import MistralTrainEngine
    from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel
    m = GritLMTrainModel.__new__(GritLMTrainModel)
    torch.nn.Module.__init__(m)
    m.model, m.projection, m.pooling_method, m.normalized, m.attn, m.embedding_attr = bb, None, "mean", True, "bbcc", None
    m.emb_loss_fn = DistributedContrastiveLoss(tau, False)
    m.train_engine = MistralTrainEngine(bb, cfg, DEV)
    return m


def check_train_nograd_f16_equals_encoder(cfg_name="gqa", policy="f16_operands"):
    """The training engine's no-grad forward under an fp16 policy (GradCache pass 1; stacked [gate; up] weights, packed parameters)
    must produce the bits of the inference engine under the same policy on the same weights -- padded and packed -- and follow an
    in-place parameter update (the fp16 copies are keyed on the parameters' version counters)."""
    from gritlm_amd.training.engine import SyntheticBackbone
    cfg = EncoderConfig.from_dict(synth.CONFIGS[cfg_name])
    bb = SyntheticBackbone(cfg, DEV, seed=5)
    sd = {k: v.detach().clone() for k, v in bb.state_dict().items()}
    m = _train_model_on(bb, cfg)
    m.train_engine.set_nograd_precision(policy)
    ids, mask = synth.make_batch(synth.CONFIGS[cfg_name], 6, 150, seed=25, min_len=9)
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    eng = MistralEncoderEngine.from_state_dict(cfg, sd, DEV)
    eng.set_precision(policy)
    _f16_flag()
    out, ok = {}, True
    for packed in (False, True):
        m.native_packed = packed
        with torch.no_grad():
            r = m.encode({"input_ids": tid, "attention_mask": tm})
        e = eng.encode_pooled(tid, tm, "mean", True, packed=packed)
        out[f"equal_packed={packed}"] = bool(torch.equal(r, e))
        ok &= out[f"equal_packed={packed}"]
    # an in-place update of one parameter must reach the fp16 copy without weights_updated()
    with torch.no_grad():
        bb.layers[0].mlp.down_proj.weight.mul_(0.5)
        r2 = m.encode({"input_ids": tid, "attention_mask": tm})
    out["follows_in_place_update"] = not bool(torch.equal(r2, r))
    sd2 = {k: v.detach().clone() for k, v in bb.state_dict().items()}
    e2 = MistralEncoderEngine.from_state_dict(cfg, sd2, DEV).set_precision(policy).encode_pooled(tid, tm, "mean", True, packed=True)
    out["equal_after_update"] = bool(torch.equal(r2, e2))
    # with grad enabled the forward is the bf16 one (its saved activations feed the bf16 backward kernels)
    r3 = m.encode({"input_ids": tid, "attention_mask": tm})
    out["grad_forward_is_bf16"] = bool(r3.requires_grad) and not bool(torch.equal(r3.detach(), r2))
    out["overflow_flag"] = _f16_flag()
    ok &= out["follows_in_place_update"] and out["equal_after_update"] and out["grad_forward_is_bf16"] and not out["overflow_flag"]
    return _res(f"train engine no-grad forward [{policy}] == inference engine, bit for bit [{cfg_name}]", bool(ok), **out)


def check_gradcache_f16_pass1(fixture="train_7b-l1", cfg_name="7b-l1", chunk=2, loss_bounds=None, rep_bounds=None):
    """GradCacheStep(precision='f16_operands' / 'f16_stream') at the TRUE 7B layer shape vs the reference's fp32 step: pass 1 (the no-grad
    forward that defines the representations and the loss) runs on fp16 operands, pass 2 in the reference's bf16 arithmetic.
    `train_7b-l1` (one layer; 2 queries + 4 passages) and `train_7b-d8` (EIGHT distinct layers; 16 queries + 32 passages: round 6, the
    depth fixture VERDICT r05 #1b asks for).  Held to: pass-1 representations within 1e-5 of the reference's fp32 reps; |loss - fp32 loss|
    within the north-star's 1e-3 (on the 2-query fixture the fp16-stream policy gets 2e-3: with 1/tau = 50 a single score moves the loss of
    2 queries by ~1e-3 at 1 - cos = 8e-7; the 16-query depth fixture holds BOTH fp16 policies to 1e-3); every parameter's gradient probe
    and norm within the SAME bounds as the all-bf16 step (1.25x the reference's own bf16 run + floor) -- i.e. feeding pass 2's bf16 forward
    with rep gradients cached at pass 1's fp16 reps costs nothing measurable against bf16's own gradient error."""
    import tempfile
    from gritlm_amd.training import GradCacheStep, GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    loss_bounds = loss_bounds or {"bf16": LOSS_VS_F32_REF, "f16_operands": LOSS_VS_F32_REF, "f16_stream": 2e-3}
    rep_bounds = rep_bounds or {"bf16": 1e-4, "f16_operands": 1e-5, "f16_stream": 1e-5}
    out, ok = {}, True
    q = {"input_ids": torch.from_numpy(g["q_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["q_mask"]).to(DEV)}
    p = {"input_ids": torch.from_numpy(g["p_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["p_mask"]).to(DEV)}
    ref_loss = float(g["loss"])
    out["loss_gap_of_reference_bf16_run"] = abs(float(g["loss_bf16"]) - ref_loss)
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), cfg_name, 0, "bfloat16")
        for pol in ("bf16", "f16_operands", "f16_stream"):
            m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                                 temperature=float(g["tau"]), negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
            m.enable_native()
            _f16_flag()
            gc = GradCacheStep(m, chunk_size=chunk, precision=pol)
            loss = gc(dict(q), dict(p))
            m.train_engine.check_f16_overflow()
            lv = float(loss.item())
            out[f"loss_minus_f32ref[{pol}]"] = lv - ref_loss
            for nm, r in zip(("q", "p"), gc.last_reps):
                c = float(np.max(1 - np.sum(f32(r).astype(np.float64) * g[nm + "_reps"].astype(np.float64), axis=1)))
                out[f"{nm}_reps_1-cos[{pol}]"] = c
                ok &= c < rep_bounds[pol]
            ok &= abs(lv - ref_loss) < loss_bounds[pol]
            worst_ratio = worst_nratio = 0.0
            for n, t in m._backbone().named_parameters():
                got = t.grad
                ref_n = float(g["gnorm/" + n])
                en = abs(float(got.double().norm().item()) - ref_n) / (ref_n + 1e-20)
                worst_nratio = max(worst_nratio, en / (1.25 * _yard()[f"{fixture}/gnorm_rel_bf16/{n}"] + GRAD_FLOOR))
                ref = g["probe/" + n]
                if n == "embed_tokens.weight":
                    gp = f32(got[torch.from_numpy(g["probe_rows/" + n]).to(DEV)])
                elif got.dim() == 2:
                    gp = f32(got[:ref.shape[0]])
                else:
                    gp = f32(got)
                e = float(np.linalg.norm(gp - ref) / (np.linalg.norm(ref) + 1e-20))
                worst_ratio = max(worst_ratio, e / (1.25 * _yard()[f"{fixture}/probe_rel_bf16/{n}"] + GRAD_FLOOR))
            out[f"probe_err_over_bound[{pol}]"] = worst_ratio
            out[f"norm_err_over_bound[{pol}]"] = worst_nratio
            ok &= worst_ratio <= 1.0 and worst_nratio <= 1.0
            del m, gc
            torch.cuda.empty_cache()
    return _res(f"GradCache with an fp16 pass 1 [{fixture}] vs reference fp32 loss + grads", bool(ok), **out)


def check_train_nograd_f16_mixtral(cfg_name="moe-tiny"):
    """MixtralTrainEngine's no-grad forward under 'f16_operands' (fp32 routing on the stream, grouped fp16 expert GEMMs on the fused
    gate_up_proj / down_proj parameters, fp32 combine) == the inference MoE engine under the same policy, bit for bit, and within 1e-5 of
    the fixture the reference produced in fp32."""
    import tempfile
    from gritlm_amd.training import GritLMTrainModel
    g = np.load(os.path.join(GOLDEN, f"encoder_{cfg_name}.npz"))
    ids, mask = g["input_ids"], g["attention_mask"]
    tid, tm = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    eng, cfg, w = build_engine(cfg_name, int(g["seed_w"]))
    eng.set_precision("f16_operands")
    out, ok = {}, True
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mixtral_dir(os.path.join(td, "x16"), cfg_name, int(g["seed_w"]), "bfloat16")
        m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        m.train_engine.set_nograd_precision("f16_operands")
        _f16_flag()
        for packed in (False, True):
            m.native_packed = packed
            with torch.no_grad():
                r = m.encode({"input_ids": tid, "attention_mask": tm})
            e = eng.encode_pooled(tid, tm, "mean", True, packed=packed)
            out[f"equal_packed={packed}"] = bool(torch.equal(r, e))
            ok &= out[f"equal_packed={packed}"]
        c = float(np.max(1 - np.sum(f32(r).astype(np.float64) * g["emb_mean"].astype(np.float64), axis=1)))
        out["1-cos_vs_reference_fp32"] = c
        out["overflow_flag"] = _f16_flag()
        ok &= c < 1e-5 and not out["overflow_flag"]
        try:
            m.train_engine.set_nograd_precision("f16_stream"); out["f16_stream_refused"] = False
        except Exception:      # noqa: BLE001
            out["f16_stream_refused"] = True
        ok &= out["f16_stream_refused"]
    return _res(f"mixtral train engine no-grad forward [f16_operands] == inference engine [{cfg_name}]", bool(ok), **out)


def check_gritlm_f16_auto_ladder():
    """precision='auto' on checkpoint-like activations.  At the 7B LAYER shape (H 4096, I 14336, 2 layers), three models:
    (a) tame N(0, 0.02) weights: 'auto' stays on f16_stream; (b) two residual channels driven to ~1e5 (beyond fp16: 65504) by scaled
    down_proj rows in both layers: f16_stream flags from the RESIDUAL epilogue / norm, the ladder steps to f16_operands (fp32 stream) and
    stays; (c) gate / up scaled so the SwiGLU activation itself crosses 65504: f16_operands flags too (the SWIGLU epilogue),
    the ladder lands on fp32_residual.  In every case encode() RETURNS (never raises), the rows are finite and within 2e-3 of the bf16
    policy's rows for the same model (1e-5 of the fp16 policy's own rows where no rung was left), the rung is sticky for the next call,
    and an explicit f16 policy on the same model raises and names the devices.  Also: the flag fires from every kernel class that rounds
    to fp16 (checked per policy through the engine)."""
    import tempfile
    from gritlm_amd import GritLM
    from gritlm_amd._lib import GritHipError
    cfgd = dict(synth.CONFIGS["7b-l1"]); cfgd["num_hidden_layers"] = 2
    sents = [" ".join(synth.WORDS[(7 * i + j) % len(synth.WORDS)] for j in range(20 + 3 * i)) for i in range(6)]
    out, ok = {}, True
    # down_proj rows x 1e5: the MLP output of a N(0, 0.02) layer is ~1.2, so channels 7 / 11 of the stream reach ~1e5 (fp16: 65504) while
    # every MFMA operand stays small (RMSNorm output <= sqrt(H / 2) = 45).  gate / up x 100: pre-activations ~ N(0, 128^2), silu(g) * u up to
    # ~3e5 -> the SwiGLU epilogue's fp16 rounding overflows under both fp16 policies (and the stream, ~4e4 rms, leaves fp16 as well)
    cases = {"tame": (None, None, "f16_stream"), "massive_stream": (1.0e5, None, "f16_operands"), "massive_act": (None, 100.0, "fp32_residual")}
    with tempfile.TemporaryDirectory() as td:
        for name, (dscale, guscale, want) in cases.items():
            w = synth.make_weights(cfgd, 3)
            if dscale:
                for li in range(2):
                    k = f"layers.{li}.mlp.down_proj.weight"
                    d = w[k].copy(); d[7, :] *= dscale; d[11, :] *= dscale
                    w[k] = O.bf16_round(d)
            if guscale:
                for k in list(w):
                    if "gate_proj" in k or "up_proj" in k:
                        w[k] = O.bf16_round(w[k] * guscale)
            d16 = synth.build_mistral_dir(os.path.join(td, name), cfgd, 3, "bfloat16", weights=w)
            m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16, precision="auto")
            ok &= m.engine is not None and m.precision == "f16_stream"
            e = m.encode(sents, batch_size=4, max_length=64)
            out[f"{name}_rung"] = m.precision
            ok &= m.precision == want and bool(np.isfinite(e).all())
            e_again = m.encode(sents, batch_size=4, max_length=64)
            ok &= m.precision == want and np.array_equal(e, e_again)                    # sticky rung, deterministic rows
            m0 = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16)
            e0 = m0.encode(sents, batch_size=4, max_length=64)
            c = float(np.max(1 - np.sum(e.astype(np.float64) * e0.astype(np.float64), axis=1)))
            out[f"{name}_1-cos_vs_bf16_policy"] = c
            ok &= c < 2e-3
            if want != "f16_stream":                                                   # the explicit policy raises, names the device, clears
                mx = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda:0", torch_dtype=torch.bfloat16, precision="f16_stream")
                try:
                    mx.encode(sents, batch_size=4, max_length=64); out[f"{name}_explicit_raises"] = False
                except GritHipError as ex:
                    out[f"{name}_explicit_raises"] = "fp16 range" in str(ex) and "cuda:0" in str(ex)
                ok &= out[f"{name}_explicit_raises"] is True
                mx.set_precision("bf16")
                ok &= bool(np.isfinite(mx.encode(sents, batch_size=4, max_length=64)).all())       # no stale flag blamed on the next call
                del mx
            # stream magnitude actually reached (bf16 policy, residual stream before the final norm)
            ids = m0.tokenizer(sents, padding=True, truncation=True, return_tensors="pt", max_length=64)
            hs = m0.engine.forward(ids["input_ids"], ids["attention_mask"], final_norm=False)
            out[f"{name}_max_abs_residual_stream"] = float(hs.float().abs().max())
            del m, m0
            torch.cuda.empty_cache()
    ok &= out["massive_stream_max_abs_residual_stream"] > 65504.0 > out["tame_max_abs_residual_stream"]
    return _res("GritLM(precision='auto'): ladder on massive-activation models at the 7B layer shape", bool(ok), **out)


def check_get_cache_f16():
    """encode(get_cache=True) under the fp16 policies: the embeddings keep the policy's accuracy (within 1e-5 of the no-cache call), the
    cache comes back in the reference's format (bf16 [B,nkv,S,d] per layer) within one bf16 ulp of the fp32-accurate K / V -- closer to
    the fp32 Hugging Face module's cache than the bf16 policy's cache is --, engine.forward(kv_dtype=None) hands out the fp16 K / V the
    attention itself read (ONE rounding from the fp32 accumulator), and the native decoder continues from the cache."""
    import tempfile
    from gritlm_amd import GritLM
    g = np.load(os.path.join(GOLDEN, "gritlm_encode.npz"))
    sents = [str(x) for x in g["sentences"]][:5]
    out, ok = {}, True
    get = lambda c, li: (c.layers[li].keys, c.layers[li].values) if hasattr(c, "layers") else (c[li][0], c[li][1])
    with tempfile.TemporaryDirectory() as td:
        d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "tiny", 0, "bfloat16")
        d32 = synth.build_mistral_dir(os.path.join(td, "m32"), "tiny", 0, "float32")
        hf = GritLM(d32, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.float32, native=False)
        emb32, cache32 = hf.encode(sents, max_length=48, get_cache=True)
        mb = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16)
        _, cache_b = mb.encode(sents, max_length=48, get_cache=True)
        for pol in ("f16_operands", "f16_stream"):
            m = GritLM(d16, pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, precision=pol)
            emb, cache = m.encode(sents, max_length=48, get_cache=True)
            emb_nc = m.encode(sents, max_length=48)
            c = float(np.max(1 - np.sum(emb.astype(np.float64) * emb32.astype(np.float64), axis=1)))
            out[f"emb_1-cos_vs_fp32[{pol}]"] = c
            ok &= c < 1e-5 and float(np.max(1 - np.sum(emb * emb_nc, axis=1))) < 1e-6
            worst = worst_b = 0.0
            for li in range(2):
                for a, b, r in zip(get(cache, li), get(cache_b, li), get(cache32, li)):
                    ok &= a.dtype == torch.bfloat16 and tuple(a.shape) == tuple(r.shape)
                    sc = float(r.float().abs().max()) + 1e-9
                    worst = max(worst, float((a.float() - r.float()).abs().max()) / sc)
                    worst_b = max(worst_b, float((b.float() - r.float()).abs().max()) / sc)
            out[f"kv_max_rel_vs_fp32[{pol}]"], out["kv_max_rel_vs_fp32[bf16]"] = worst, worst_b
            ok &= worst <= 2.0 ** -8 and worst <= worst_b
            ids = m.tokenizer(sents, padding=True, truncation=True, return_tensors="pt", max_length=48)
            _, kv16 = m.engine.forward(ids["input_ids"], ids["attention_mask"], return_kv=True, kv_dtype=None)
            ok &= kv16[0][0].dtype == torch.float16
            k16 = kv16[1][0].float(); r = get(cache32, 1)[0].float()
            out[f"k_fp16_max_rel_vs_fp32[{pol}]"] = float((k16 - r).abs().max() / (r.abs().max() + 1e-9))
            ok &= out[f"k_fp16_max_rel_vs_fp32[{pol}]"] < 2.0 ** -10
            if hasattr(m.model, "lm_head"):
                with torch.no_grad():
                    q = m.tokenizer([" ".join(synth.WORDS[5:12])], return_tensors="pt", add_special_tokens=False)["input_ids"].to(DEV)
                    plen = int(m.tokenizer([sents[0]], return_tensors="pt", truncation=True, max_length=48)["input_ids"].shape[1])
                    one = [(get(cache, li)[0][:1, :, :plen].contiguous(), get(cache, li)[1][:1, :, :plen].contiguous()) for li in range(2)]
                    toks = m.native_decoder().generate(q, 3, past_key_values=one)
                    ok &= m.native_decoder().last_precision == "f16"                           # the decoder follows the engine's policy (round 6)
                    toks2 = m.native_decoder().generate(q, 3)                                  # prompt prefill under an fp16 policy: runs, policy restored
                    # native_kv_cache: the fp16 K/V the attention read, handed to the native decoder as they are
                    m.native_kv_cache = True
                    _, cache_n = m.encode(sents, max_length=48, get_cache=True)
                    m.native_kv_cache = False
                    ok &= get(cache_n, 0)[0].dtype == torch.float16 and bool(torch.equal(get(cache_n, 1)[0], kv16[1][0]))
                    one_n = [(get(cache_n, li)[0][:1, :, :plen].contiguous(), get(cache_n, li)[1][:1, :, :plen].contiguous()) for li in range(2)]
                    _, lg_n = m.native_decoder().generate(q, 3, past_key_values=one_n, return_logits=True)
                    _, lg_b = m.native_decoder().generate(q, 3, past_key_values=one, return_logits=True)
                    ok &= lg_n.dtype == torch.float32 and float((1 - torch.nn.functional.cosine_similarity(lg_n[0].double(), lg_b[0].double(), dim=1)).max()) < 1e-4
                ok &= tuple(toks.shape) == (1, 3) and tuple(toks2.shape) == (1, 3) and m.engine.precision == pol
            del m
    return _res("encode(get_cache=True) under the fp16 policies", bool(ok), **out)


ALL_CHECKS = [
    ("embed", check_embed, {}),
    ("rmsnorm_4096", check_rmsnorm, dict(T=37, H=4096)),
    ("rmsnorm_256", check_rmsnorm, dict(T=9, H=256)),
    ("rmsnorm_1024", check_rmsnorm, dict(T=130, H=1024)),
    ("rope", check_rope, {}),
    ("rope_inv", check_rope, dict(inverse=True)),
    ("gemm_256", check_gemm, dict(M=256, N=256, K=64)),
    ("gemm_edge", check_gemm, dict(M=300, N=272, K=128)),
    ("gemm_big", check_gemm, dict(M=1024, N=768, K=512)),
    ("gemm_small_m", check_gemm, dict(M=17, N=1536, K=256)),
    ("gemm_residual", check_gemm, dict(M=520, N=512, K=192, epi=EPI_RESIDUAL)),
    ("gemm_swiglu", check_gemm, dict(M=300, N=1024, K=256, epi=EPI_SWIGLU)),
    ("gemm_swiglu_edge", check_gemm, dict(M=70, N=576, K=64, epi=EPI_SWIGLU)),
    ("gemm_residual_f32", check_gemm_residual_f32, {}),
    ("gemm_residual_f32_edge", check_gemm_residual_f32, dict(M=300, N=272, K=128, in_place=False)),
    ("gemm_residual_f32_persistent", check_gemm_residual_f32, dict(M=8192, N=4096, K=512, seed=29)),      # 512 tiles >= 2 per CU: the persistent form
    ("f32_stream_ops", check_f32_stream_ops, {}),
    ("f32_stream_ops_768", check_f32_stream_ops, dict(T=9, H=768)),
    ("f32_stream_ops_264", check_f32_stream_ops, dict(T=5, H=264)),
    ("gemm_counter_rings_reclaimed", check_gemm_counter_rings_are_reclaimed, {}),
    ("gemm_pair", check_gemm_pair, {}),
    ("gemm_pair_store", check_gemm_pair, dict(M1=17, N1=1536, M2=1024, N2=256, K=256, accumulate=False)),
    ("gemm_pair_wgrad_shape", check_gemm_pair, dict(M1=4096, N1=14336, M2=6144, N2=4096, K=2048)),
    ("gemm_rope", check_gemm_rope, {}),
    ("gemm_rope_packed", check_gemm_rope, dict(M=513, nq=2, nkv=1, K=128, packed=True)),
    ("gemm_rope_7b", check_gemm_rope, dict(M=1024, nq=32, nkv=8, K=512, S=512)),
    ("mask_pack", check_mask_pack, {}),
    ("attn_ragged", check_attention, dict(mask_kind="ragged")),
    ("attn_holes", check_attention, dict(mask_kind="holes", S=257)),
    ("attn_left", check_attention, dict(mask_kind="left", S=192)),
    ("attn_full_512", check_attention, dict(B=1, S=512, nq=8, nkv=2, mask_kind="none")),
    ("attn_short", check_attention, dict(B=3, S=33, nq=2, nkv=1, mask_kind="ragged")),
    ("attn_causal_ragged", check_attention, dict(mask_kind="ragged", causal=True)),
    ("attn_causal_full_512", check_attention, dict(B=1, S=512, nq=8, nkv=2, mask_kind="none", causal=True)),
    ("attn_causal_short", check_attention, dict(B=3, S=33, nq=2, nkv=1, mask_kind="ragged", causal=True)),
    ("attn_causal_holes", check_attention, dict(mask_kind="holes", S=257, causal=True)),
    ("attn_short_rows", check_attention, dict(B=2, S=512, nq=8, nkv=2, mask_kind="short_rows", seed=24)),
    ("attn_seam_qpw2", check_attention_seam, dict(qpw=2)),
    ("attn_seam_qpw4", check_attention_seam, dict(qpw=4)),
    ("attn_production_shape", check_attention_production_shape, {}),
    ("attn_production_shape_causal", check_attention_production_shape, dict(causal=True, seed=32)),
    ("pool_mean", check_pool, dict(method="mean")),
    ("pool_weightedmean", check_pool, dict(method="weightedmean")),
    ("pool_cls", check_pool, dict(method="cls")),
    ("pool_lasttoken", check_pool, dict(method="lasttoken")),
    ("pool_mean_nonorm", check_pool, dict(method="mean", normalize=False)),
    ("pool_mean_4096", check_pool, dict(method="mean", B=2, S=19, H=4096)),
    ("pool_bwd_mean", check_pool_bwd, dict(method="mean")),
    ("pool_bwd_weightedmean", check_pool_bwd, dict(method="weightedmean")),
    ("pool_bwd_nonorm", check_pool_bwd, dict(method="mean", normalize=False)),
    ("infonce_a", check_infonce, dict(tag="a")),
    ("infonce_b", check_infonce, dict(tag="b")),
    ("infonce_c", check_infonce, dict(tag="c")),
    ("infonce_local", check_infonce_local_rows, {}),
    ("infonce_big", check_infonce_big, {}),
    ("infonce_reproducible", check_infonce_reproducible, {}),
    ("infonce_tiles128", check_infonce_shapes, dict(nq=1024, group=4, H=96)),                       # 8 x 32 = 256 tiles of 128 x 128
    ("infonce_tiles128_local", check_infonce_shapes, dict(nq=1024, group=4, H=64, q_off=384, nq_loc=128, p_off=1536, np_loc=512)),
    ("infonce_scalar_loads", check_infonce_shapes, dict(nq=75, group=3, H=50)),                     # H % 4 != 0, Np % 4 != 0
    ("infonce_odd_offsets", check_infonce_shapes, dict(nq=130, group=2, H=132, q_off=7, nq_loc=33, p_off=13, np_loc=71)),
    ("infonce_few_rows", check_infonce_shapes, dict(nq=8, group=8, H=260)),                          # M <= 32: the 32 x 128 tile
    ("infonce_ragged_k", check_infonce_shapes, dict(nq=96, group=5, H=1028)),                        # K = 1028 / 480 / 96: partial K-steps
    ("transpose", check_transpose, {}),
    ("rmsnorm_bwd", check_rmsnorm_bwd, {}),
    ("rmsnorm_bwd_4096", check_rmsnorm_bwd, dict(T=21, H=4096, with_res=False)),
    ("rmsnorm_bwd_4096_res", check_rmsnorm_bwd, dict(T=2077, H=4096, with_res=True)),     # > 4 x 512 rows: every workgroup loops
    ("rmsnorm_bwd_2048_res", check_rmsnorm_bwd, dict(T=300, H=2048, with_res=True)),
    ("swiglu", check_swiglu, {}),
    ("attn_bwd_ragged", check_attention_bwd, {}),
    ("attn_bwd_holes", check_attention_bwd, dict(mask_kind="holes", S=130, B=2, nq=2, nkv=1)),
    ("attn_bwd_full", check_attention_bwd, dict(mask_kind="none", S=256, B=1, nq=8, nkv=2)),
    ("attn_bwd_causal_ragged", check_attention_bwd, dict(causal=True)),
    ("attn_bwd_causal_full", check_attention_bwd, dict(mask_kind="none", S=256, B=1, nq=8, nkv=2, causal=True)),
    ("attn_bwd_causal_330", check_attention_bwd, dict(mask_kind="ragged", S=330, B=2, nq=2, nkv=1, causal=True)),
    ("attn_bwd_varlen_causal", check_attention_bwd_varlen, dict(causal=True)),
    ("attn_window_16", check_attention, dict(mask_kind="ragged", causal=True, window=16)),
    ("attn_window_1", check_attention, dict(B=1, S=130, nq=2, nkv=1, mask_kind="none", causal=True, window=1)),
    ("attn_window_64_513", check_attention, dict(B=3, S=513, nq=4, nkv=2, mask_kind="ragged", seed=26, causal=True, window=64)),
    ("attn_window_100_holes", check_attention, dict(B=2, S=330, nq=2, nkv=1, mask_kind="holes", seed=22, causal=True, window=100)),
    ("attn_window_129_1024", check_attention, dict(B=1, S=1024, nq=2, nkv=1, mask_kind="none", seed=25, causal=True, window=129)),
    ("attn_window_300_short_rows", check_attention, dict(B=2, S=512, nq=8, nkv=2, mask_kind="short_rows", seed=27, causal=True, window=300)),
    ("attn_window_ge_S", check_attention, dict(mask_kind="ragged", causal=True, window=200)),
    ("attn_bwd_window_16", check_attention_bwd, dict(causal=True, window=16)),
    ("attn_bwd_window_2", check_attention_bwd, dict(mask_kind="none", S=130, B=1, nq=2, nkv=1, causal=True, window=2)),   # (window 1: p = 1, dq = dk = 0 exactly)
    ("attn_bwd_window_100_330", check_attention_bwd, dict(mask_kind="ragged", S=330, B=2, nq=2, nkv=1, causal=True, window=100)),
    ("attn_bwd_window_129_holes", check_attention_bwd, dict(mask_kind="holes", S=400, B=2, nq=2, nkv=1, causal=True, window=129)),
    ("attn_bwd_varlen_window_64", check_attention_bwd_varlen, dict(lens=(385, 33, 512, 7), nq=4, nkv=2, causal=True, window=64)),
    ("attn_bwd_varlen_window_200", check_attention_bwd_varlen, dict(lens=(200, 71, 128, 1, 300, 513, 64, 40), nq=4, nkv=2, causal=True, window=200)),
    ("attn_bwd_varlen", check_attention_bwd_varlen, {}),
    ("attn_bwd_varlen_gqa4", check_attention_bwd_varlen, dict(lens=(129, 64, 257), nq=8, nkv=2)),
    ("pool_bwd_varlen_mean", check_pool_bwd_varlen, dict(method="mean")),
    ("pool_bwd_varlen_weightedmean", check_pool_bwd_varlen, dict(method="weightedmean")),
    ("pool_bwd_varlen_lasttoken", check_pool_bwd_varlen, dict(method="lasttoken", normalize=False)),
    ("embed_scatter", check_embed_scatter, {}),
    ("encoder_tiny", check_encoder_golden, dict(cfg_name="tiny")),
    ("encoder_gqa", check_encoder_golden, dict(cfg_name="gqa")),
    ("encoder_oracle", check_encoder_vs_oracle_bf16, {}),
    ("inputs_embeds_and_layer_range", check_inputs_embeds_and_layer_range, {}),
    ("encoder_7b_layer", check_encoder_7b_layer, {}),
    ("encoder_tiny_fp32_residual", check_encoder_fp32_residual, dict(cfg_name="tiny")),
    ("encoder_gqa_fp32_residual", check_encoder_fp32_residual, dict(cfg_name="gqa")),
    ("encoder_7b_layer_fp32_residual", check_encoder_fp32_residual, dict(cfg_name="7b-l1")),
    ("moe_router_bwd", check_moe_router_bwd, {}),
    ("moe_router_bwd_e4_no_aux", check_moe_router_bwd, dict(T=300, H=256, E=4, aux=False)),
    ("moe_router_bwd_e16_h4096", check_moe_router_bwd, dict(T=1100, H=4096, E=16, aux=True)),
    ("gritlm_multi_gpu_in_process", check_gritlm_multi_gpu_in_process, {}),
    ("full_depth_parity_32_layers", check_full_depth_parity, {}),
    ("full_depth_parity_32_layers_fp32_residual", check_full_depth_parity, dict(residual_fp32=True)),
    ("full_depth_parity_32_layers_f16_operands", check_full_depth_parity, dict(precision="f16_operands")),
    ("full_depth_parity_32_layers_f16_stream", check_full_depth_parity, dict(precision="f16_stream")),
    ("gemm_f16_256", check_gemm_f16, dict(M=256, N=256, K=64)),
    ("gemm_f16_edge", check_gemm_f16, dict(M=300, N=272, K=128)),
    ("gemm_f16_big", check_gemm_f16, dict(M=1024, N=768, K=512)),
    ("gemm_f16_subnormal_weights", check_gemm_f16, dict(M=520, N=512, K=256, subnormal_weights=True)),
    ("gemm_f16_swiglu", check_gemm_f16, dict(M=300, N=1024, K=256, epi=EPI_SWIGLU)),
    ("gemm_f16_swiglu_edge", check_gemm_f16, dict(M=70, N=576, K=64, epi=EPI_SWIGLU)),
    ("gemm_f16_residual_f32", check_gemm_f16, dict(M=520, N=512, K=192, epi=EPI_RESIDUAL_F32)),
    ("gemm_f16_residual_f32_persistent", check_gemm_f16, dict(M=8192, N=4096, K=512, epi=EPI_RESIDUAL_F32, seed=129)),
    ("gemm_f16_residual_f16_stream", check_gemm_f16, dict(M=520, N=512, K=192, epi=EPI_RESIDUAL, seed=131)),
    ("gemm_f16_residual_f16_stream_edge", check_gemm_f16, dict(M=300, N=272, K=128, epi=EPI_RESIDUAL, seed=133)),
    ("gemm_f16_residual_f16_stream_persistent", check_gemm_f16, dict(M=8192, N=4096, K=512, epi=EPI_RESIDUAL, seed=135)),
    ("gemm_f16_rope", check_gemm_f16_rope, {}),
    ("gemm_f16_rope_packed", check_gemm_f16_rope, dict(M=513, nq=2, nkv=1, K=128, packed=True)),
    ("gemm_f16_rope_7b", check_gemm_f16_rope, dict(M=1024, nq=32, nkv=8, K=512, S=512)),
    ("gemm_f16_full_swiglu_28672x4096", check_gemm_f16_fullshape, dict(M=4096, N=28672, K=4096, epi=EPI_SWIGLU)),
    ("gemm_f16_full_store_6144x4096", check_gemm_f16_fullshape, dict(M=4096, N=6144, K=4096)),
    ("gemm_f16_full_residual_m131072_k14336", check_gemm_f16_fullshape, dict(M=131072, N=4096, K=14336, epi=EPI_RESIDUAL_F32)),
    ("f16_overflow_flag", check_f16_overflow_flag, {}),
    ("f16_stream_ops", check_f16_stream_ops, {}),
    ("f16_stream_ops_264", check_f16_stream_ops, dict(T=5, H=264)),
    ("f16_stream_norm_and_gather", check_f16_stream_norm_and_gather, {}),
    ("f16_stream_norm_and_gather_264", check_f16_stream_norm_and_gather, dict(T=5, H=264)),
    ("attn_f16_ragged", check_attention_f16, dict(mask_kind="ragged")),
    ("attn_f16_holes", check_attention_f16, dict(mask_kind="holes", S=257)),
    ("attn_f16_full_512", check_attention_f16, dict(B=1, S=512, nq=8, nkv=2, mask_kind="none")),
    ("attn_f16_packed", check_attention_f16, dict(B=4, S=513, nq=4, nkv=2, packed_lens=(513, 1, 129, 300))),
    ("attn_f16_causal", check_attention_f16, dict(B=2, S=513, nq=4, nkv=2, mask_kind="ragged", seed=213, causal=True)),
    ("attn_f16_causal_window", check_attention_f16, dict(B=2, S=330, nq=4, nkv=2, mask_kind="ragged", seed=215, causal=True, window=100)),
    ("attn_f16_causal_packed", check_attention_f16, dict(B=4, S=513, nq=4, nkv=2, packed_lens=(513, 1, 129, 300), causal=True, seed=217)),
    ("encoder_causal_f16", check_encoder_causal_f16, {}),
    ("encoder_tiny_f16_operands", check_encoder_f16_operands, dict(cfg_name="tiny")),
    ("encoder_gqa_f16_operands", check_encoder_f16_operands, dict(cfg_name="gqa")),
    ("encoder_7b_layer_f16_operands", check_encoder_f16_operands, dict(cfg_name="7b-l1")),
    ("encoder_tiny_f16_stream", check_encoder_f16_operands, dict(cfg_name="tiny", policy="f16_stream")),
    ("encoder_gqa_f16_stream", check_encoder_f16_operands, dict(cfg_name="gqa", policy="f16_stream")),
    ("encoder_7b_layer_f16_stream", check_encoder_f16_operands, dict(cfg_name="7b-l1", policy="f16_stream")),
    ("f16_policy_raises_on_overflow", check_f16_policy_raises_on_overflow, {}),
    ("gemm_f16_swiglu_stacked", check_gemm_f16_stacked, {}),
    ("gemm_f16_grouped_store", check_gemm_f16_grouped, {}),
    ("gemm_f16_grouped_swiglu", check_gemm_f16_grouped, dict(N=512, epi=EPI_SWIGLU)),
    ("moe_router_f32", check_moe_router_f32, {}),
    ("moe_router_f32_16e", check_moe_router_f32, dict(T=300, H=4096, E=16)),
    ("moe_combine_f32", check_moe_combine_f32, {}),
    ("mixtral_f16_operands_moe-tiny", check_mixtral_f16_operands, {}),
    ("mixtral_layer_true_shape_8x7b_f16_original_criteria", check_mixtral_layer_true_shape_f16, {}),
    ("train_nograd_f16_operands_equals_encoder", check_train_nograd_f16_equals_encoder, {}),
    ("train_nograd_f16_stream_equals_encoder", check_train_nograd_f16_equals_encoder, dict(policy="f16_stream")),
    ("train_nograd_f16_mixtral", check_train_nograd_f16_mixtral, {}),
    ("gradcache_f16_pass1_7b_layer", check_gradcache_f16_pass1, {}),
    # eight layers, 16 queries: the reference's own bf16 run is 1.2e-4 off in the reps here, and both fp16 policies are held to the north-star's loss tolerance
    ("gradcache_f16_pass1_7b_depth8", check_gradcache_f16_pass1,
     dict(fixture="train_7b-d8", cfg_name="7b-d8", chunk=8, loss_bounds={"bf16": 1e-2, "f16_operands": 1e-3, "f16_stream": 1e-3},
          rep_bounds={"bf16": 5e-4, "f16_operands": 1e-5, "f16_stream": 1e-5})),      # (the reference's own bf16 run: 3.3e-4)
    ("gritlm_f16_auto_ladder", check_gritlm_f16_auto_ladder, {}),
    ("get_cache_f16", check_get_cache_f16, {}),
    ("gritlm_f16_operands", check_gritlm_f16_operands, {}),
    ("gemm_full_swiglu_28672x4096", check_gemm_fullshape, dict(M=4096, N=28672, K=4096, epi=EPI_SWIGLU)),
    ("gemm_full_residual_4096x14336", check_gemm_fullshape, dict(M=4096, N=4096, K=14336, epi=EPI_RESIDUAL)),
    ("gemm_full_store_6144x4096", check_gemm_fullshape, dict(M=4096, N=6144, K=4096)),
    ("gemm_full_m131072", check_gemm_fullshape, dict(M=131072, N=4096, K=4096, epi=EPI_RESIDUAL)),
    ("gemm_full_m131072_k14336", check_gemm_fullshape, dict(M=131072, N=4096, K=14336)),
    ("moe_router", check_moe_router, {}),
    ("moe_router_e4_4096", check_moe_router, dict(T=130, H=4096, E=4)),
    ("gemm_grouped", check_gemm_grouped, {}),
    ("gemm_grouped_swiglu", check_gemm_grouped, dict(counts=(70, 5, 0, 260), N=512, K=128, epi=EPI_SWIGLU)),
    ("moe_block", check_moe_block, {}),
    ("mixtral_tiny", check_mixtral_golden, dict(cfg_name="moe-tiny")),
    ("mixtral_gqa", check_mixtral_golden, dict(cfg_name="moe-gqa")),
    ("mixtral_layer_true_shape_8x7b", check_mixtral_layer_true_shape, {}),
    ("gemm_grouped_full_swiglu_28672x4096", check_gemm_grouped_fullshape, {}),
    ("gemm_grouped_full_store_4096x14336", check_gemm_grouped_fullshape, dict(N=4096, K=14336, epi=EPI_STORE, seed=303)),
    ("causal_encode", check_causal_encode, {}),
    ("sliding_window_encode", check_sliding_window_encode, {}),
    ("generative_window", check_generative_window, {}),
    ("edge_cases", check_edge_cases, {}),
    ("long_sequence_4096", check_long_sequence, {}),
    ("full_shape_properties", check_full_shape_properties, {}),
    ("packed_encode", check_packed_encode, {}),
    ("packed_encode_tiny", check_packed_encode, dict(cfg_name="tiny", B=3, S=260)),
    ("gritlm_native_encode", check_gritlm_native_encode, {}),
    ("gritlm_api_variants", check_gritlm_api_variants, {}),
    ("gritlm_native_mixtral", check_gritlm_native_mixtral, {}),
    ("get_cache", check_get_cache, {}),
    ("train_direct", check_train_step, dict(mode="direct")),
    ("train_gradcache", check_train_step, dict(mode="gradcache")),
    ("train_7b_layer", check_train_step_7b_layer, {}),
    ("grouped_training_epilogues", check_grouped_training_epilogues, {}),
    ("train_mixtral_direct", check_train_step_mixtral, dict(mode="direct")),
    ("train_mixtral_gradcache", check_train_step_mixtral, dict(mode="gradcache")),
    ("train_mixtral_recompute", check_train_step_mixtral, dict(mode="recompute")),
    ("generative_mixtral_aux", check_generative_mixtral, {}),
    ("train_packed_vs_padded", check_train_packed_vs_padded, {}),
    ("gradcache_pass1_superchunks", check_gradcache_pass1_superchunks, {}),
    ("train_recompute", check_train_recompute, {}),
    ("swiglu_stacked", check_swiglu_stacked, {}),
    ("swiglu_train_epilogues", check_swiglu_fused_train_epilogues, {}),
    ("swiglu_train_epilogues_big", check_swiglu_fused_train_epilogues, dict(M=4100, I=14336, K=4096)),
    ("swiglu_stacked_7b", check_swiglu_stacked, dict(M=512, I=14336, K=4096)),
    ("rccl_world1_step", check_rccl_world1_step, {}),
    ("native_comm_world1", check_native_comm, {}),
    ("ce", check_ce, {}),
    ("ce_vocab32000", check_ce, dict(T=40, V=32000)),
    ("generative_mixed", check_generative_step, dict(kind="mixed")),
    ("generative_token", check_generative_step, dict(kind="token")),
    ("gemv", check_gemv, {}),
    ("gemv_b1_7b", check_gemv, dict(B=1, N=6144, K=4096)),
    ("gemv_b8_residual", check_gemv, dict(B=8, N=515, K=1024, epi=EPI_RESIDUAL)),
    ("gemv_swiglu", check_gemv, dict(B=2, N=1024, K=256, epi=EPI_SWIGLU)),
    ("decode_fused_ops", check_decode_fused_ops, {}),
    ("decode_deferred_norm", check_decode_deferred_norm, {}),
    ("attn_decode", check_attn_decode, {}),
    ("attn_decode_gqa4_b1", check_attn_decode, dict(B=1, nq=32, nkv=8, Lmax=2304, lens=(2100,))),
    ("native_generate", check_native_generate, {}),
    ("native_generate_gqa", check_native_generate, dict(cfg_name="gqa", P=9, new=6, rows=3)),
    ("native_generate_exact_fused_norm", check_native_generate, dict(fuse_norm="all")),
    # bf16 model vs the FP32 oracle at H = 4096 / I = 14336 (K = 14336 bf16 activations): measured 6.6 % of the logit spread, against
    # 1.5 % per layer for the reference's own bf16 run at this shape (encoder_7b-l1 fixture); greedy tokens must still agree
    ("native_generate_7b_layer_shape", check_native_generate, dict(cfg_name="7b-l2s", P=12, new=6, rows=1, tol=0.10)),
    ("gemv_f16", check_gemv_f16, {}),
    ("gemv_f16_7b", check_gemv_f16, dict(N=6144, K=4096)),
    ("attn_decode_f16", check_attn_decode_f16, {}),
    ("attn_decode_f16_gqa4_b1", check_attn_decode_f16, dict(B=1, nq=32, nkv=8, Lmax=2304, lens=(2100,))),
    ("attn_decode_f16_mha", check_attn_decode_f16, dict(B=2, nq=4, nkv=4, Lmax=512, lens=(300, 64))),              # one query head per kv head: the non-per-head launch
    ("attn_decode_f16_gqa8", check_attn_decode_f16, dict(B=1, nq=16, nkv=2, Lmax=512, lens=(511,))),               # 8 query heads per kv head; the last slot of the cache
    ("attn_decode_f16_long", check_attn_decode_f16, dict(B=1, nq=8, nkv=2, Lmax=8192, lens=(8000,))),              # 126 splits: the combine's tail loop beyond 64 splits
    ("argmax_f32", check_argmax_f32, {}),
    ("argmax_f32_b8", check_argmax_f32, dict(B=8, V=1001)),
    ("native_generate_f16", check_native_generate_f16, {}),
    ("native_generate_f16_stream_gqa", check_native_generate_f16, dict(cfg_name="gqa", P=9, new=6, rows=3, policy="f16_stream")),
    ("native_generate_f16_7b_layer_shape", check_native_generate_f16, dict(cfg_name="7b-l2s", P=12, new=6, rows=1)),
    ("native_generate_f16_rows8", check_native_generate_f16, dict(cfg_name="tiny", P=5, new=4, rows=8)),          # the 8-row instantiations of every GEMV form
    ("gemv_expert", check_gemv_expert, {}),
    ("gemv_expert_8x7b_w13_strides", check_gemv_expert, dict(big=True)),
    ("native_generate_moe", check_native_generate, dict(cfg_name="moe-tiny", P=9, new=6, rows=2, tol=0.08)),
    ("native_generate_moe_f16", check_native_generate_f16, dict(cfg_name="moe-tiny", P=9, new=6, rows=2, policy="f16_operands", cos_bound=2e-5)),
    ("generate_native_api", check_generate_native_api, {}),
    ("native_generate_prompt_chunk", check_native_generate_prompt_chunk, {}),
    ("native_generate_prompt_chunk_f16", check_native_generate_prompt_chunk, dict(policy="f16_stream")),
    ("native_generate_prompt_chunk_b1_p37", check_native_generate_prompt_chunk, dict(cfg_name="tiny", B=1, P=37, new=3)),
    ("native_generate_prompt_chunk_7b_layer_shape_f16", check_native_generate_prompt_chunk, dict(cfg_name="7b-l2s", B=1, P=9, new=3, policy="f16_operands")),
    ("train_causal_embedding_weightedmean", check_train_causal_embedding, {}),
    ("train_causal_embedding_lasttoken", check_train_causal_embedding, dict(pooling="lasttoken")),
    ("wgrad_accumulation_drift", check_wgrad_accumulation_drift, {}),
    ("knn_topk", check_knn_topk, {}),
    ("knn_topk_transposed_big", check_knn_topk, dict(Q=3, N=300000, H=128, k=100, transposed=True)),
    ("knn_topk_small", check_knn_topk, dict(Q=2, N=37, H=64, k=37)),
    ("knn_topk_transposed_odd", check_knn_topk, dict(Q=33, N=4099, H=68, k=5, transposed=True)),    # [H,N] index, N % 4 != 0, 64 x 64 tiles
    ("knn_topk_q32", check_knn_topk, dict(Q=32, N=70000, H=128, k=10)),                              # the 32 x 128 tile on a long index
    ("rag_distributed_index", check_rag_distributed_index, {}),
    ("cli_native", check_cli_native, {}),
    ("cli_native_mixtral", check_cli_native, dict(arch="mixtral")),
    ("cli_unified_native", check_cli_unified_native, {}),
    ("cli_unified_native_mixtral", check_cli_unified_native, dict(arch="mixtral")),
    ("overlapped_grad_sync", check_overlapped_grad_sync, {}),
]
