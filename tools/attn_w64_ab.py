#!/usr/bin/env python3
"""A/B of the two bidirectional attention forwards behind grit_attn_bidir_fwd / grit_attn_bidir_varlen_fwd: the W64 kernel (round 4: 64
query rows per wave, one wave per SIMD, in-wave QK / softmax pipeline) against the round-3 kernel (GRIT_ATTN_FWD=v3), in ONE process.

  * equality: every per-row operation of the two kernels is the same arithmetic in the same order, so the outputs and the LSE rows must be
    BIT-IDENTICAL on every shape -- full tiles, ragged tails, masks with holes, 1 / 2 / 3-tile sequences, all-masked rows, packed rows;
  * timing: interleaved launches, HIP events, TFLOP/s = 4 B nq S^2 d / t (padded) or 4 nq d sum(len^2) / t (packed).

    python tools/attn_w64_ab.py [--quick] [--out gpurun_out/attn_w64_ab.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gritlm_amd import ops  # noqa: E402

DEV = "cuda"
NQ, NKV, D = 32, 8, 128


def use(which):
    if which == "w64":
        os.environ["GRIT_ATTN_FWD"] = "w64"
    else:
        os.environ.pop("GRIT_ATTN_FWD", None)              # the default kernel


def mk_qkv(T, g, nq=NQ, nkv=NKV):
    return (torch.randn((T, (nq + 2 * nkv) * D), generator=g, device=DEV, dtype=torch.float32)).to(torch.bfloat16)


def run_padded(qkv, bits, B, S, nq, nkv, which):
    use(which)
    out = torch.full((B * S, nq * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.full((B, nq, S), float("nan"), dtype=torch.float32, device=DEV)
    ops.attn_bidir(qkv, bits, B, S, nq, nkv, D, out=out, lse=lse)
    return out, lse


def run_varlen(qkv, cu, max_len, nq, nkv, which):
    use(which)
    T = qkv.shape[0]
    out = torch.full((T, nq * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.full((T, nq), float("nan"), dtype=torch.float32, device=DEV)
    ops.attn_bidir_varlen(qkv, cu, max_len, nq, nkv, D, out=out, lse=lse)
    return out, lse


def same(a, b):
    """bit equality, NaN == NaN (rows the kernels never write keep their NaN fill on both sides)"""
    return bool(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a.view(torch.int32),
                            b.view(torch.int16) if b.dtype == torch.bfloat16 else b.view(torch.int32)))


def time_pair(fn, flops, rounds=5, inner=3):
    res = {}
    for which in ("v3", "w64"):
        fn(which)
    torch.cuda.synchronize()
    acc = {"v3": [], "w64": []}
    for _ in range(rounds):
        for which in ("v3", "w64"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn(which)
            e1.record()
            torch.cuda.synchronize()
            acc[which].append(e0.elapsed_time(e1) / inner)
    for which in acc:
        ms = sorted(acc[which])[len(acc[which]) // 2]
        res[which] = {"ms": ms, "tflops": flops / (ms * 1e-3) / 1e12}
    res["w64_over_v3"] = res["v3"]["ms"] / res["w64"]["ms"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "attn_w64_ab.json"))
    a = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(5)
    report = {"equality": [], "timing": {}}
    ok = True

    # ---- equality, padded layout with key bitmasks
    def bits_of(mask):
        return ops.mask_pack(mask.to(torch.int64).contiguous())

    cases = []
    for (B, S, nq, nkv, kind) in [(2, 64, 8, 2, "full"), (2, 128, 8, 2, "full"), (3, 130, 8, 2, "ragged"), (2, 37, 4, 4, "ragged"),
                                   (4, 200, 8, 2, "holes"), (3, 512, 32, 8, "ragged"), (2, 700, 8, 8, "holes"), (2, 192, 8, 2, "allmasked_row"),
                                   (5, 257, 16, 4, "left_padding"), (2, 1024, 8, 2, "full")]:
        mask = torch.ones((B, S), dtype=torch.int64, device=DEV)
        if kind == "ragged":
            lens = torch.randint(1, S + 1, (B,), generator=g, device=DEV)
            lens[0] = S
            mask = (torch.arange(S, device=DEV).unsqueeze(0) < lens.unsqueeze(1)).to(torch.int64)
        elif kind == "holes":
            mask = (torch.rand((B, S), generator=g, device=DEV) > 0.3).to(torch.int64)
            mask[:, S - 50:] = 0
            mask[0, :] = 1
        elif kind == "allmasked_row":
            mask[1, :] = 0
        elif kind == "left_padding":
            lens = torch.randint(1, S + 1, (B,), generator=g, device=DEV)
            mask = (torch.arange(S, device=DEV).unsqueeze(0) >= (S - lens).unsqueeze(1)).to(torch.int64)
        qkv = mk_qkv(B * S, g, nq, nkv)
        # a spiked key: forces the (rare) rescale branch in some rows
        qkv[S // 2, nq * D:(nq + 1) * D] *= 6.0
        bits = bits_of(mask)
        o3, l3 = run_padded(qkv, bits, B, S, nq, nkv, "v3")
        o6, l6 = run_padded(qkv, bits, B, S, nq, nkv, "w64")
        torch.cuda.synchronize()
        eq_o, eq_l = same(o3, o6), same(l3, l6)
        md = float((o3.float() - o6.float()).nan_to_num().abs().max())
        report["equality"].append({"layout": "padded", "B": B, "S": S, "nq": nq, "nkv": nkv, "mask": kind, "out_equal": eq_o, "lse_equal": eq_l,
                                   "max_abs_diff": md, "finite": bool(torch.isfinite(o6.float()).all())})
        ok &= eq_o and eq_l
    # ---- equality, packed layout
    for lens_l, nq, nkv in [([1], 4, 2), ([64], 4, 2), ([65, 3], 8, 2), ([128, 129, 127], 8, 2), ([192, 200, 7, 512], 8, 2),
                            ([300, 511, 512, 64, 90, 17], 32, 8), ([1000, 30, 640], 8, 8)]:
        lens = torch.tensor(lens_l, dtype=torch.int32, device=DEV)
        cu = torch.zeros((len(lens_l) + 1,), dtype=torch.int32, device=DEV)
        cu[1:] = torch.cumsum(lens, 0)
        T = int(cu[-1])
        qkv = mk_qkv(T, g, nq, nkv)
        o3, l3 = run_varlen(qkv, cu, max(lens_l), nq, nkv, "v3")
        o6, l6 = run_varlen(qkv, cu, max(lens_l), nq, nkv, "w64")
        torch.cuda.synchronize()
        eq_o, eq_l = same(o3, o6), same(l3, l6)
        report["equality"].append({"layout": "packed", "lens": lens_l, "nq": nq, "nkv": nkv, "out_equal": eq_o, "lse_equal": eq_l,
                                   "max_abs_diff": float((o3.float() - o6.float()).nan_to_num().abs().max()),
                                   "finite": bool(torch.isfinite(o6.float()).all())})
        ok &= eq_o and eq_l
    report["all_equal"] = bool(ok)
    print(json.dumps(report["equality"], indent=0), flush=True)

    # ---- timing
    shapes = [("B256_S512", 256, 512), ("B64_S2048", 64, 2048)] + ([] if a.quick else [("B16_S8192", 16, 8192), ("B32_S512", 32, 512)])
    for name, B, S in shapes:
        qkv = mk_qkv(B * S, g)
        bits = bits_of(torch.ones((B, S), dtype=torch.int64, device=DEV))
        out = torch.empty((B * S, NQ * D), dtype=torch.bfloat16, device=DEV)

        def fn(which, qkv=qkv, bits=bits, B=B, S=S, out=out):
            use(which)
            ops.attn_bidir(qkv, bits, B, S, NQ, NKV, D, out=out)
        report["timing"][name] = time_pair(fn, 4.0 * B * NQ * S * S * D)
        print(name, json.dumps(report["timing"][name]), flush=True)
        del qkv, out
    # packed: full-length rows (what the training step's packed path runs) and ragged rows
    for name, lens_t in [("packed_256x512", torch.full((256,), 512, dtype=torch.int32)),
                         ("packed_ragged_U64_512_x256", torch.randint(64, 513, (256,), generator=torch.Generator().manual_seed(3), dtype=torch.int32))]:
        lens = lens_t.to(DEV)
        cu = torch.zeros((lens.numel() + 1,), dtype=torch.int32, device=DEV)
        cu[1:] = torch.cumsum(lens, 0)
        T = int(cu[-1])
        qkv = mk_qkv(T, g)
        out = torch.empty((T, NQ * D), dtype=torch.bfloat16, device=DEV)
        mx = int(lens.max())

        def fn(which, qkv=qkv, cu=cu, mx=mx, out=out):
            use(which)
            ops.attn_bidir_varlen(qkv, cu, mx, NQ, NKV, D, out=out)
        report["timing"][name] = time_pair(fn, 4.0 * NQ * D * float((lens.double() ** 2).sum()))
        print(name, json.dumps(report["timing"][name]), flush=True)
        del qkv, out
    use("v3")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(report, open(a.out, "w"), indent=1)
    print("ALL_EQUAL" if ok else "MISMATCH", "wrote", a.out)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
