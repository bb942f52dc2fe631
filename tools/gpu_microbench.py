#!/usr/bin/env python3
"""Per-kernel timings at GritLM-7B shapes (HIP events on torch's current stream). Prints TF/s and GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gritlm_amd import ops  # noqa: E402
from gritlm_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def gemm_case(M, N, K, epi, tag):
    a = torch.randn((M, K), device=DEV, dtype=torch.float32).to(BF)
    w = (torch.randn((N, K), device=DEV, dtype=torch.float32) * 0.02).to(BF)
    n_out = N // 2 if epi == EPI_SWIGLU else N
    out = torch.empty((M, n_out), device=DEV, dtype=BF)
    res = torch.randn((M, N), device=DEV, dtype=torch.float32).to(BF) if epi == EPI_RESIDUAL else None
    med, mn = timeit(lambda: ops.gemm_nt(a, w, out=out, epilogue=epi, residual=res))
    fl = 2.0 * M * N * K
    print(f"gemm {tag:<10s} M={M:<7d} N={N:<6d} K={K:<6d} epi={epi}  med {med:8.3f} ms  min {mn:8.3f} ms  {fl / med / 1e9:8.1f} TF/s (med)  {fl / mn / 1e9:8.1f} TF/s (best)", flush=True)
    # torch (hipBLASLt) comparator on the same data
    med2, mn2 = timeit(lambda: torch.matmul(a, w.t()))
    print(f"     torch.matmul comparator                              med {med2:8.3f} ms  {fl / med2 / 1e9:8.1f} TF/s", flush=True)


def attn_case(B, S, nq=32, nkv=8):
    d = 128
    qkv = torch.randn((B * S, (nq + 2 * nkv) * d), device=DEV, dtype=torch.float32).to(BF)
    mask = torch.ones((B, S), dtype=torch.int64, device=DEV)
    bits = ops.mask_pack(mask)
    out = torch.empty((B * S, nq * d), device=DEV, dtype=BF)
    med, mn = timeit(lambda: ops.attn_bidir(qkv, bits, B, S, nq, nkv, d, out=out))
    fl = 4.0 * B * nq * S * S * d
    print(f"attn B={B} S={S} nq={nq} nkv={nkv}  med {med:8.3f} ms  {fl / med / 1e9:8.1f} TF/s", flush=True)
    q = qkv.view(B, S, nq + 2 * nkv, d)[:, :, :nq].transpose(1, 2)
    k = qkv.view(B, S, nq + 2 * nkv, d)[:, :, nq:nq + nkv].transpose(1, 2)
    v = qkv.view(B, S, nq + 2 * nkv, d)[:, :, nq + nkv:].transpose(1, 2)
    try:
        med2, _ = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=True))
        print(f"     torch sdpa comparator                 med {med2:8.3f} ms  {fl / med2 / 1e9:8.1f} TF/s", flush=True)
    except Exception as e:  # noqa: BLE001
        print("     torch sdpa comparator failed:", repr(e)[:200])


def hbm_cases(T=131072, H=4096, B=256, S=512):
    x = torch.randn((T, H), device=DEV, dtype=torch.float32).to(BF)
    w = torch.ones((H,), device=DEV, dtype=BF)
    y = torch.empty_like(x)
    med, mn = timeit(lambda: ops.rmsnorm(x, w, 1e-5, out=y))
    print(f"rmsnorm T={T} H={H}  med {med:7.3f} ms  {2 * T * H * 2 / med / 1e6:8.1f} GB/s", flush=True)
    mask = torch.ones((B, S), dtype=torch.int64, device=DEV)
    med, mn = timeit(lambda: ops.pool_norm(x.view(B, S, H), mask, "mean", True))
    print(f"pool+norm B={B} S={S} H={H}  med {med:7.3f} ms  {(T * H * 2 + B * H * 4) / med / 1e6:8.1f} GB/s", flush=True)
    qkv = torch.randn((T, 6144), device=DEV, dtype=torch.float32).to(BF)
    from gritlm_amd.encoder import rope_tables
    cos, sin = rope_tables(S, 128, 1e4, True, DEV)
    med, mn = timeit(lambda: ops.rope_qk_(qkv, cos, sin, S, 32, 8, 128))
    print(f"rope T={T}  med {med:7.3f} ms  {2 * T * 5120 * 2 / med / 1e6:8.1f} GB/s", flush=True)
    ids = torch.randint(0, 32000, (T,), device=DEV)
    tab = torch.randn((32000, H), device=DEV, dtype=torch.float32).to(BF)
    med, mn = timeit(lambda: ops.embed_gather(tab, ids, out=y))
    print(f"embed T={T}  med {med:7.3f} ms  {2 * T * H * 2 / med / 1e6:8.1f} GB/s", flush=True)


def new_kernel_cases():
    """HBM-bound kernels of the round-1 'next' rows: vocabulary CE, MoE router / combine, decode GEMVs."""
    T, V = 16384, 32000
    logits = torch.randn((T, V), device=DEV, dtype=torch.float32).to(BF)
    labels = torch.randint(0, V, (T,), device=DEV)
    med, _ = timeit(lambda: ops.ce_fwd(logits, labels))
    print(f"ce_fwd T={T} V={V}  med {med:7.3f} ms  {T * V * 2 / med / 1e6:8.1f} GB/s", flush=True)
    lse, _ = ops.ce_fwd(logits, labels)
    med, _ = timeit(lambda: ops.ce_bwd_(logits, labels, lse, 1.0))
    print(f"ce_bwd T={T} V={V}  med {med:7.3f} ms  {2 * T * V * 2 / med / 1e6:8.1f} GB/s", flush=True)
    T, H, E = 131072, 4096, 8
    x = torch.randn((T, H), device=DEV, dtype=torch.float32).to(BF)
    gw = (torch.randn((E, H), device=DEV) * 0.5).to(BF)
    med, _ = timeit(lambda: ops.moe_route(x, gw))
    print(f"moe router+index T={T}  med {med:7.3f} ms  {T * H * 2 / med / 1e6:8.1f} GB/s (x read)", flush=True)
    _, wts, counts, row_token, rows = ops.moe_route(x, gw)
    y = torch.randn((2 * T, H), device=DEV, dtype=torch.float32).to(BF)
    out = torch.empty_like(x)
    med, _ = timeit(lambda: ops.moe_combine(y, rows, wts, x, out=out))
    print(f"moe combine T={T}  med {med:7.3f} ms  {4 * T * H * 2 / med / 1e6:8.1f} GB/s", flush=True)
    for (N, K, epi, tag) in ((6144, 4096, EPI_STORE, "qkv"), (28672, 4096, EPI_SWIGLU, "gate_up"), (4096, 14336, EPI_RESIDUAL, "down"), (32000, 4096, EPI_STORE, "lm_head")):
        xv = torch.randn((1, K), device=DEV, dtype=torch.float32).to(BF)
        w = (torch.randn((N, K), device=DEV, dtype=torch.float32) * 0.02).to(BF)
        res = torch.randn((1, N), device=DEV, dtype=torch.float32).to(BF) if epi == EPI_RESIDUAL else None
        o = torch.empty((1, N // 2 if epi == EPI_SWIGLU else N), device=DEV, dtype=BF)
        med, _ = timeit(lambda: ops.gemv(xv, w, out=o, epilogue=epi, residual=res))
        print(f"gemv {tag:<8s} N={N:<6d} K={K:<6d}  med {med * 1e3:7.1f} us  {N * K * 2 / med / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "new":
        new_kernel_cases()
        sys.exit(0)
    M = int(os.environ.get("MB_M", 131072))
    gemm_case(M, 6144, 4096, EPI_STORE, "qkv")
    gemm_case(M, 4096, 4096, EPI_RESIDUAL, "o_proj")
    gemm_case(M, 28672, 4096, EPI_SWIGLU, "gate_up")
    gemm_case(M, 4096, 14336, EPI_RESIDUAL, "down")
    gemm_case(8192, 8192, 8192, EPI_STORE, "square8k")
    attn_case(256, 512)
    attn_case(16, 2048)
    attn_case(4, 8192)
    hbm_cases()
