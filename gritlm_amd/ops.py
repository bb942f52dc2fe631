"""Thin torch-tensor -> C-ABI adapters.  PyTorch is plumbing here: it owns device memory and streams;
all arithmetic happens in libgritlm_hip.so.  Every function launches on torch's current HIP stream."""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import (EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_STORE, EPI_SWIGLU, EPI_SWIGLU_BWD, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE,
                   POOL_MODES, check)

BF16, F16, F32, I64, I32 = torch.bfloat16, torch.float16, torch.float32, torch.int64, torch.int32


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-kernel-class HIP-event timing (bench.py's live roofline measurement).

    Events are recorded on torch's current stream -- the stream every kernel here is launched on."""

    def __init__(self):
        self.pairs: dict[str, list] = {}
        self.work: dict[str, float] = {}
        self.tags: dict[str, list] = {}
        self.deferred: dict[str, list] = {}

    def span(self, name: str, work=0.0, tag: str | None = None):
        """``work``: FLOPs (or bytes) of the launch -- a float, or a 0-dim DEVICE tensor when the amount depends on device data (the packed
        attention's sum of len^2): it is read at summary(), after the timed region, never inside it."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.pairs.setdefault(name, []).append((a, b))
        if torch.is_tensor(work):
            self.deferred.setdefault(name, []).append(work)
            work = 0.0
        self.tags.setdefault(name, []).append((tag, work))
        self.work[name] = self.work.get(name, 0.0) + work
        return a, b

    def summary(self) -> dict:
        torch.cuda.synchronize()
        for name, ws in self.deferred.items():
            self.work[name] = self.work.get(name, 0.0) + float(sum(float(w) for w in ws))
        self.deferred = {}
        out = {}
        for name, pairs in self.pairs.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = dict(launches=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms), work=self.work[name])
            by_tag = {}
            for t, (tag, work) in zip(ms, self.tags[name]):
                if tag is not None:
                    d = by_tag.setdefault(tag, dict(launches=0, total_ms=0.0, work=0.0))
                    d["launches"] += 1; d["total_ms"] += t; d["work"] += work
            if by_tag:
                out[name]["by_tag"] = by_tag
        return out


_timer: KernelTimer | None = None
_TAG_M = bool(os.environ.get("GRIT_TIMER_TAG_M"))          # tools/contrastive_shapes.py: the GEMM tags carry M as well (forward / dgrad / wgrad shapes apart)


def set_timer(t: KernelTimer | None):
    global _timer
    _timer = t


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.GritHipError(f"{name}: tensor must live on the GPU (got {t.device}); the native path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _chk2d(t: torch.Tensor, dtype, name: str):
    """Row-strided 2-D operand (unit column stride, leading dimension = stride(0)): what the GEMM entry point takes."""
    if not t.is_cuda:
        raise _lib.GritHipError(f"{name}: tensor must live on the GPU (got {t.device}); the native path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: 2-D tensor with unit column stride expected")
    return t.data_ptr()


def _chk3d(t: torch.Tensor, dtype, name: str):
    """[E,N,K] stack of row-strided matrices (unit column stride; leading dimension stride(1), matrix stride stride(0))."""
    if not t.is_cuda:
        raise _lib.GritHipError(f"{name}: tensor must live on the GPU (got {t.device}); the native path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError(f"{name}: 3-D tensor with unit column stride expected")
    return t.data_ptr()


def embed_gather(table: torch.Tensor, ids: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    V, H = table.shape
    T = ids.numel()
    if out is None:
        out = torch.empty((T, H), dtype=BF16, device=table.device)
    if table.dtype == F16:      # fp16 residual stream: an fp16 copy of the table, rows copied (the gather moves 16-bit words)
        check(_lib.load().grit_embed_gather(_chk(table, F16, "table"), _chk(ids, I64, "ids"), _chk(out, F16, "out"), T, H, V, _stream()),
              "grit_embed_gather")
        return out
    if out.dtype == F32:        # fp32 residual stream: rows widened (exact)
        check(_lib.load().grit_embed_gather_f32(_chk(table, BF16, "table"), _chk(ids, I64, "ids"), _chk(out, F32, "out"), T, H, V, _stream()),
              "grit_embed_gather_f32")
        return out
    check(_lib.load().grit_embed_gather(_chk(table, BF16, "table"), _chk(ids, I64, "ids"), _chk(out, BF16, "out"), T, H, V, _stream()),
          "grit_embed_gather")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """bf16 rows: the reference's bf16 arithmetic.  fp32 rows (the encoder's fp32 residual stream): one rounding, bf16 out -- or fp16 out
    when ``out`` is an fp16 tensor (the f16_operands policy; w stays bf16)."""
    H = x.shape[-1]
    T = x.numel() // H
    if out is None:
        out = torch.empty(x.shape, dtype=BF16, device=x.device)
    if x.dtype == F16:          # fp16 residual stream: one rounding, to fp16 (the next operand) or to bf16 (last_hidden_state)
        check(_lib.load().grit_rmsnorm_fwd_f16in(_chk(x, F16, "x"), _chk(w, BF16, "w"), _chk(out, out.dtype if out.dtype in (F16, BF16) else F16, "out"),
                                                 int(out.dtype == F16), T, H, float(eps), _stream()), "grit_rmsnorm_fwd_f16in")
    elif x.dtype == F32 and out.dtype == F16:
        check(_lib.load().grit_rmsnorm_fwd_f32in_f16(_chk(x, F32, "x"), _chk(w, BF16, "w"), _chk(out, F16, "out"), T, H, float(eps), _stream()),
              "grit_rmsnorm_fwd_f32in_f16")
    elif x.dtype == F32:
        check(_lib.load().grit_rmsnorm_fwd_f32in(_chk(x, F32, "x"), _chk(w, BF16, "w"), _chk(out, BF16, "out"), T, H, float(eps), _stream()),
              "grit_rmsnorm_fwd_f32in")
    else:
        check(_lib.load().grit_rmsnorm_fwd(_chk(x, BF16, "x"), _chk(w, BF16, "w"), _chk(out, BF16, "out"), T, H, float(eps), _stream()),
              "grit_rmsnorm_fwd")
    return out


def rope_qk_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, S: int, nq: int, nkv: int, d: int, inverse: bool = False):
    T, stride = qkv.shape
    check(_lib.load().grit_rope_qk_inplace(_chk(qkv, BF16, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"), T, S, nq, nkv, d,
                                           stride, int(inverse), _stream()), "grit_rope_qk_inplace")
    return qkv


def rope_qk_pos_(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, positions: torch.Tensor, nq: int, nkv: int, d: int,
                 inverse: bool = False):
    """RoPE for packed rows: positions[t] (int32) = index of token t inside its own sequence."""
    T, stride = qkv.shape
    check(_lib.load().grit_rope_qk_inplace_pos(_chk(qkv, BF16, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"),
                                               _chk(positions, I32, "positions"), T, cos.shape[0], nq, nkv, d, stride, int(inverse),
                                               _stream()),
          "grit_rope_qk_inplace_pos")
    return qkv


def _attn_entry(kind: str, causal: bool, window: int):
    """(entry point, extra arguments) of an attention call: bidirectional, causal, or causal with a sliding window of ``window`` keys."""
    lib = _lib.load()
    if window and window > 0:
        if not causal:
            raise ValueError("a sliding window applies to causal attention only (the bidirectional embedding path ignores it, as the reference)")
        return getattr(lib, "grit_attn_causal_window_" + kind), (int(window),)
    return getattr(lib, ("grit_attn_causal_" if causal else "grit_attn_bidir_") + kind), ()


def attn_bidir_varlen(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, nq: int, nkv: int, d: int,
                      out: torch.Tensor | None = None, scale: float | None = None, lse: torch.Tensor | None = None,
                      causal: bool = False, window: int = 0) -> torch.Tensor:
    """lse (optional, fp32 [T, nq]) receives the log-sum-exp rows for grit_attn_bidir_varlen_bwd.  ``window`` > 0 (causal only): sliding
    window, a query sees keys q - window + 1 .. q."""
    T, stride = qkv.shape
    B = cu_seqlens.numel() - 1
    if out is None:
        out = torch.empty((T, nq * d), dtype=qkv.dtype, device=qkv.device)
    if scale is None:
        scale = d ** -0.5
    ev = None
    if _timer is not None:           # algorithmic FLOPs of the packed launch: 4 d nq sum(len^2) (half of it when causal), summed on the device
        lens = (cu_seqlens[1:] - cu_seqlens[:-1]).double()
        ev = _timer.span("attn_bidir_fwd", (lens * lens).sum() * (4.0 * nq * d * (0.5 if causal else 1.0)))
    if ev:
        ev[0].record()
    fn, wa = _attn_entry("varlen_fwd", causal, window)
    odt = BF16
    if qkv.dtype == F16:            # the fp16-operand policies
        if window and window > 0 and not causal:
            raise ValueError("a sliding window applies to causal attention only")
        fn, wa, odt = (_lib.load().grit_attn_causal_varlen_f16_fwd, (int(window or 0),), F16) if causal else (_lib.load().grit_attn_bidir_varlen_f16_fwd, (), F16)
    check(fn(_chk(qkv, odt, "qkv"), _chk(cu_seqlens, I32, "cu_seqlens"), _chk(out, odt, "out"),
             0 if lse is None else _chk(lse, F32, "lse"), B, int(max_len), nq, nkv, d, stride, out.stride(0), float(scale), *wa, _stream()),
          "grit_attn_bidir_varlen_fwd")
    if ev:
        ev[1].record()
    return out


def pool_norm_varlen(hidden: torch.Tensor, cu_seqlens: torch.Tensor, method: str, normalize: bool,
                     instr_len: torch.Tensor | None = None, inv_norm: torch.Tensor | None = None) -> torch.Tensor:
    if method not in POOL_MODES:
        raise NotImplementedError(f"Unknown pooling method: {method}")
    T, H = hidden.shape
    B = cu_seqlens.numel() - 1
    out = torch.empty((B, H), dtype=F32, device=hidden.device)
    check(_lib.load().grit_pool_norm_varlen_fwd(_chk(hidden, BF16, "hidden"), _chk(cu_seqlens, I32, "cu_seqlens"),
                                                0 if instr_len is None else _chk(instr_len, I32, "instr_len"), _chk(out, F32, "out"),
                                                0 if inv_norm is None else _chk(inv_norm, F32, "inv_norm"), B, H,
                                                POOL_MODES[method], int(normalize), _stream()), "grit_pool_norm_varlen_fwd")
    return out


def pool_norm_varlen_bwd(y: torch.Tensor, dy: torch.Tensor, inv_norm: torch.Tensor | None, cu_seqlens: torch.Tensor, T: int, method: str,
                         normalize: bool, instr_len: torch.Tensor | None = None) -> torch.Tensor:
    B, H = y.shape
    dh = torch.empty((T, H), dtype=BF16, device=y.device)
    check(_lib.load().grit_pool_norm_varlen_bwd(_chk(y, F32, "y"), _chk(dy, F32, "dy"),
                                                0 if inv_norm is None else _chk(inv_norm, F32, "inv_norm"),
                                                _chk(cu_seqlens, I32, "cu_seqlens"),
                                                0 if instr_len is None else _chk(instr_len, I32, "instr_len"), _chk(dh, BF16, "dhidden"),
                                                B, H, POOL_MODES[method], int(normalize), _stream()), "grit_pool_norm_varlen_bwd")
    return dh


def gemm_nt(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor | None = None, epilogue: int = EPI_STORE,
            residual: torch.Tensor | None = None) -> torch.Tensor:
    """out[M,N] = a[M,K] @ w[N,K]^T (+ epilogue).  SWIGLU: w holds interleaved gate/up rows (SWIGLU_STACKED: [gate; up]), out is [M, N/2].
    SWIGLU_STACKED_SAVE: additionally writes the bf16 [gate | up] pre-activations into `residual` ([M, N]).
    SWIGLU_BWD: a @ w^T is d_act [M, N]; `residual` holds the saved [gate | up] ([M, 2N]); out = [d_gate | d_up] ([M, 2N])."""
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    n_out = N // 2 if epilogue in (EPI_SWIGLU, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE) else (2 * N if epilogue == EPI_SWIGLU_BWD else N)
    opd = a.dtype if a.dtype == F16 else BF16                 # fp16 operands: the f16_operands policy (STORE, SWIGLU, RESIDUAL_F32)
    odt = F32 if epilogue == EPI_RESIDUAL_F32 else opd        # RESIDUAL_F32: out and residual are the fp32 residual stream
    if opd == F16 and epilogue not in (EPI_STORE, EPI_SWIGLU, EPI_SWIGLU_STACKED, EPI_RESIDUAL, EPI_RESIDUAL_F32):
        raise _lib.GritHipError(f"gemm_nt: epilogue {epilogue} is not built for fp16 operands (STORE, SWIGLU, SWIGLU_STACKED, RESIDUAL, RESIDUAL_F32)")
    if out is None:
        out = torch.empty((M, n_out), dtype=odt, device=a.device)
    assert out.shape == (M, n_out)
    rp, ldr = 0, 0
    if epilogue in (EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_SWIGLU_STACKED_SAVE, EPI_SWIGLU_BWD):
        assert residual is not None and residual.shape == (M, 2 * N if epilogue == EPI_SWIGLU_BWD else N)
        rp, ldr = _chk2d(residual, odt, "residual"), residual.stride(0)
    name = "gemm_f16_nt" if opd == F16 else "gemm_bf16_nt"
    ev = _timer.span(name, 2.0 * M * N * K, tag=(f"M={M}," if _TAG_M else "") + f"N={N},K={K},epi={epilogue}") if _timer is not None else None
    if ev:
        ev[0].record()
    check(getattr(_lib.load(), "grit_" + name)(_chk2d(a, opd, "a"), _chk2d(w, opd, "w"), _chk2d(out, odt, "out"), M, N, K, a.stride(0),
                                               w.stride(0), out.stride(0), epilogue, rp, ldr, _stream()), "grit_" + name)
    if ev:
        ev[1].record()
    return out


def gemm_nt_pair(a1: torch.Tensor, w1: torch.Tensor, out1: torch.Tensor, a2: torch.Tensor, w2: torch.Tensor, out2: torch.Tensor,
                 accumulate: bool = True) -> None:
    """Two dense GEMMs with the same K in ONE launch: out_i (+)= a_i[M_i,K] @ w_i[N_i,K]^T, bit-identical to two gemm_nt calls.  Used for
    weight-gradient pairs whose tile counts do not fill whole waves of CUs on their own (``accumulate``: RESIDUAL epilogue onto out_i)."""
    (M1, K), (M2, K2) = a1.shape, a2.shape
    N1, N2 = w1.shape[0], w2.shape[0]
    assert K == K2 == w1.shape[1] == w2.shape[1] and out1.shape == (M1, N1) and out2.shape == (M2, N2)
    epi = EPI_RESIDUAL if accumulate else EPI_STORE
    ev = _timer.span("gemm_bf16_nt", 2.0 * (M1 * N1 + M2 * N2) * K, tag=(f"M={M1}+{M2}," if _TAG_M else "") + f"pair N={N1}+{N2},K={K},epi={epi}") if _timer is not None else None
    if ev:
        ev[0].record()
    o1, o2 = _chk2d(out1, BF16, "out1"), _chk2d(out2, BF16, "out2")
    check(_lib.load().grit_gemm_bf16_nt_pair(_chk2d(a1, BF16, "a1"), _chk2d(w1, BF16, "w1"), o1, o1 if accumulate else 0, M1, N1, a1.stride(0),
                                             w1.stride(0), out1.stride(0), out1.stride(0) if accumulate else 0,
                                             _chk2d(a2, BF16, "a2"), _chk2d(w2, BF16, "w2"), o2, o2 if accumulate else 0, M2, N2, a2.stride(0),
                                             w2.stride(0), out2.stride(0), out2.stride(0) if accumulate else 0, K, epi, _stream()),
          "grit_gemm_bf16_nt_pair")
    if ev:
        ev[1].record()


def gemm_nt_rope(a: torch.Tensor, w: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, rope_cols: int, S: int = 0,
                 positions: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """Fused QKV projection + RoPE on the leading ``rope_cols`` columns (q, k heads; head_dim 128): gemm_nt followed by rope_qk_[pos_],
    in one launch.  Positions: ``positions`` int32 [M] (packed rows) or row % S."""
    M, K = a.shape
    N = w.shape[0]
    opd = a.dtype if a.dtype == F16 else BF16                 # fp16 operands: rotation on the fp32 accumulators, one rounding
    if out is None:
        out = torch.empty((M, N), dtype=opd, device=a.device)
    name = "gemm_f16_nt" if opd == F16 else "gemm_bf16_nt"
    ev = _timer.span(name, 2.0 * M * N * K, tag=(f"M={M}," if _TAG_M else "") + f"N={N},K={K},epi=3") if _timer is not None else None
    if ev:
        ev[0].record()
    check(getattr(_lib.load(), "grit_" + name + "_rope")(_chk2d(a, opd, "a"), _chk2d(w, opd, "w"), _chk2d(out, opd, "out"), M, N, K, a.stride(0),
                                                         w.stride(0), out.stride(0), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"),
                                                         0 if positions is None else _chk(positions, I32, "positions"), int(S), cos.shape[0],
                                                         int(rope_cols), _stream()), "grit_" + name + "_rope")
    if ev:
        ev[1].record()
    return out


def gemm_nt_grouped(a: torch.Tensor, w: torch.Tensor, counts: torch.Tensor, m_total: int, out: torch.Tensor | None = None,
                    epilogue: int = EPI_STORE, a_rows: torch.Tensor | None = None) -> torch.Tensor:
    """Grouped GEMM over ``w [E,N,K]``: sorted row r (group by group, ``counts`` int32 [E] on the device) is
    a[a_rows[r]] @ w[g]^T.  ``m_total`` = number of sorted rows (2T for top-2 routing).  fp16 ``a`` / ``w`` / ``out``: the fp16
    instantiation (the "f16_operands" policy of the MoE engine; STORE, SWIGLU, SWIGLU_STACKED)."""
    E, N, K = w.shape
    assert a.shape[1] == K
    opd = F16 if a.dtype == F16 else BF16
    n_out = N // 2 if epilogue in (EPI_SWIGLU, EPI_SWIGLU_STACKED) else N
    if out is None:
        out = torch.empty((m_total, n_out), dtype=opd, device=a.device)
    assert out.shape == (m_total, n_out)
    name = "gemm_f16_nt_grouped" if opd == F16 else "gemm_bf16_nt_grouped"
    ev = _timer.span(name, 2.0 * m_total * N * K) if _timer is not None else None
    if ev:
        ev[0].record()
    check(getattr(_lib.load(), "grit_" + name)(_chk2d(a, opd, "a"), 0 if a_rows is None else _chk(a_rows, I32, "a_rows"),
                                               _chk3d(w, opd, "w"), _chk2d(out, opd, "out"), _chk(counts, I32, "counts"), E, m_total, N, K,
                                               a.stride(0), w.stride(1), w.stride(0), out.stride(0), epilogue, _stream()),
          "grit_" + name)
    if ev:
        ev[1].record()
    return out


def gemm_nt_grouped_epi(a: torch.Tensor, w: torch.Tensor, counts: torch.Tensor, m_total: int, epilogue: int, out: torch.Tensor | None = None,
                        residual: torch.Tensor | None = None, a_rows: torch.Tensor | None = None) -> torch.Tensor:
    """Grouped GEMM over ``w [E,N,K]`` with the training epilogues (see grit_gemm_bf16_nt_grouped_epi): STORE / SWIGLU* give
    ``out [m_total, N or N/2]``; SWIGLU_STACKED_SAVE also fills ``residual [m_total, N]`` with [gate | up]; SWIGLU_BWD reads the saved
    ``residual [m_total, 2N]`` and gives ``out [m_total, 2N]`` = [d_gate | d_up]."""
    E, N, K = w.shape
    assert a.shape[1] == K
    if epilogue in (EPI_SWIGLU, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE):
        n_out = N // 2
    elif epilogue == EPI_SWIGLU_BWD:
        n_out = 2 * N
    else:
        n_out = N
    if out is None:
        out = torch.empty((m_total, n_out), dtype=BF16, device=a.device)
    assert out.shape == (m_total, n_out)
    if epilogue in (EPI_SWIGLU_STACKED_SAVE, EPI_SWIGLU_BWD):
        need = N if epilogue == EPI_SWIGLU_STACKED_SAVE else 2 * N
        assert residual is not None and tuple(residual.shape) == (m_total, need), "residual: [m_total, N] (SAVE) / [m_total, 2N] (BWD)"
    ev = _timer.span("gemm_bf16_nt_grouped", 2.0 * m_total * N * K) if _timer is not None else None
    if ev:
        ev[0].record()
    check(_lib.load().grit_gemm_bf16_nt_grouped_epi(_chk2d(a, BF16, "a"), 0 if a_rows is None else _chk(a_rows, I32, "a_rows"),
                                                    _chk3d(w, BF16, "w"), _chk2d(out, BF16, "out"),
                                                    0 if residual is None else _chk2d(residual, BF16, "residual"), _chk(counts, I32, "counts"),
                                                    E, m_total, N, K, a.stride(0), w.stride(1), w.stride(0), out.stride(0),
                                                    0 if residual is None else residual.stride(0), epilogue, _stream()),
          "grit_gemm_bf16_nt_grouped_epi")
    if ev:
        ev[1].record()
    return out


def moe_combine_bwd(dout: torch.Tensor, y: torch.Tensor, row_token: torch.Tensor, rows: torch.Tensor, weights: torch.Tensor):
    """Backward of moe_combine: (dy [2T,H] bf16 = w * dout[token], dw [T,2] fp32 = <y, dout[token]>)."""
    T, H = dout.shape
    dy = torch.empty((2 * T, H), dtype=BF16, device=dout.device)
    dw = torch.empty((T, 2), dtype=F32, device=dout.device)
    check(_lib.load().grit_moe_combine_bwd(_chk(dout, BF16, "dout"), _chk(y, BF16, "y"), _chk(row_token, I32, "row_token"), _chk(rows, I32, "rows"),
                                           _chk(weights, F32, "weights"), dy.data_ptr(), dw.data_ptr(), T, H, _stream()), "grit_moe_combine_bwd")
    return dy, dw


def moe_router_bwd(x: torch.Tensor, gate_w: torch.Tensor, experts: torch.Tensor, dw: torch.Tensor, dx_in: torch.Tensor | None,
                   gate_grad: torch.Tensor, aux_dlogits: torch.Tensor | None = None, return_dlogits: bool = False):
    """Backward of the router (softmax -> top-2 -> renormalise through the gate Linear): returns dx [T,H] bf16 = dx_in + dlogits @ gate_w and
    accumulates gate_grad [E,H] bf16 += bf16(dlogits^T @ x).  dw [T,2] fp32 comes from moe_combine_bwd; aux_dlogits [T,E] fp32 optional."""
    T, H = x.shape
    E = gate_w.shape[0]
    lib = _lib.load()
    dlogits = torch.empty((T, E), dtype=F32, device=x.device)
    dx = torch.empty((T, H), dtype=BF16, device=x.device)
    check(lib.grit_moe_router_bwd(_chk(x, BF16, "x"), _chk(gate_w, BF16, "gate_w"), _chk(experts, I32, "experts"), _chk(dw, F32, "dw"),
                                  0 if aux_dlogits is None else _chk(aux_dlogits, F32, "aux_dlogits"),
                                  0 if dx_in is None else _chk(dx_in, BF16, "dx_in"), dx.data_ptr(), dlogits.data_ptr(), T, H, E, _stream()),
          "grit_moe_router_bwd")
    ws = torch.empty((int(lib.grit_moe_router_wgrad_workspace_floats(T, H, E)),), dtype=F32, device=x.device)
    check(lib.grit_moe_router_wgrad(_chk(x, BF16, "x"), dlogits.data_ptr(), _chk(gate_grad, BF16, "gate_grad"), ws.data_ptr(), T, H, E, _stream()),
          "grit_moe_router_wgrad")
    return (dx, dlogits) if return_dlogits else dx


def moe_route(x: torch.Tensor, gate_w: torch.Tensor):
    """Top-2 routing of x [T,H] -> (experts [T,2] i32, weights [T,2] f32, counts [E] i32, row_token [2T] i32, rows [T,2] i32)."""
    T, H = x.shape
    E = gate_w.shape[0]
    dev = x.device
    experts = torch.empty((T, 2), dtype=I32, device=dev)
    weights = torch.empty((T, 2), dtype=F32, device=dev)
    counts = torch.empty((E,), dtype=I32, device=dev)
    row_token = torch.empty((2 * T,), dtype=I32, device=dev)
    rows = torch.empty((T, 2), dtype=I32, device=dev)
    check(_lib.load().grit_moe_router_top2(_chk(x, BF16, "x"), _chk(gate_w, BF16, "gate_w"), experts.data_ptr(), weights.data_ptr(), T, H, E,
                                           _stream()), "grit_moe_router_top2")
    ws = torch.empty((int(_lib.load().grit_moe_index_workspace_ints(T, E)),), dtype=I32, device=dev)
    check(_lib.load().grit_moe_index(experts.data_ptr(), T, E, counts.data_ptr(), row_token.data_ptr(), rows.data_ptr(), ws.data_ptr(), _stream()),
          "grit_moe_index")
    return experts, weights, counts, row_token, rows


def moe_combine(y: torch.Tensor, rows: torch.Tensor, weights: torch.Tensor, residual: torch.Tensor | None, out: torch.Tensor | None = None):
    """out[t] = residual[t] + w[t,0] y[rows[t,0]] + w[t,1] y[rows[t,1]].  bf16 y / residual / out: the reference's bf16 rounding points;
    fp16 y with an fp32 residual / out: the "f16_operands" policy (fp32 arithmetic, nothing rounded)."""
    T = rows.shape[0]
    H = y.shape[1]
    if y.dtype == F16:
        if out is None:
            out = torch.empty((T, H), dtype=F32, device=y.device)
        check(_lib.load().grit_moe_combine_f32(_chk(y, F16, "y"), _chk(rows, I32, "rows"), _chk(weights, F32, "weights"),
                                               0 if residual is None else _chk(residual, F32, "residual"), _chk(out, F32, "out"), T, H, _stream()),
              "grit_moe_combine_f32")
        return out
    if out is None:
        out = torch.empty((T, H), dtype=BF16, device=y.device)
    check(_lib.load().grit_moe_combine(_chk(y, BF16, "y"), _chk(rows, I32, "rows"), _chk(weights, F32, "weights"),
                                       0 if residual is None else _chk(residual, BF16, "residual"), _chk(out, BF16, "out"), T, H, _stream()),
          "grit_moe_combine")
    return out


def moe_route_f32(h: torch.Tensor, ln_w: torch.Tensor, eps: float, gate_w: torch.Tensor):
    """moe_route for the "f16_operands" policy: the routing decision is taken in fp32 on the residual stream h [T,H] itself (the
    post-attention RMSNorm folded in, nothing rounded: grit_moe_router_top2_f32).  Same return tuple as moe_route."""
    T, H = h.shape
    E = gate_w.shape[0]
    dev = h.device
    experts = torch.empty((T, 2), dtype=I32, device=dev)
    weights = torch.empty((T, 2), dtype=F32, device=dev)
    counts = torch.empty((E,), dtype=I32, device=dev)
    row_token = torch.empty((2 * T,), dtype=I32, device=dev)
    rows = torch.empty((T, 2), dtype=I32, device=dev)
    check(_lib.load().grit_moe_router_top2_f32(_chk(h, F32, "h"), _chk(ln_w, BF16, "ln_w"), float(eps), _chk(gate_w, BF16, "gate_w"),
                                               experts.data_ptr(), weights.data_ptr(), T, H, E, _stream()), "grit_moe_router_top2_f32")
    ws = torch.empty((int(_lib.load().grit_moe_index_workspace_ints(T, E)),), dtype=I32, device=dev)
    check(_lib.load().grit_moe_index(experts.data_ptr(), T, E, counts.data_ptr(), row_token.data_ptr(), rows.data_ptr(), ws.data_ptr(), _stream()),
          "grit_moe_index")
    return experts, weights, counts, row_token, rows


def f16_overflow_flag(device, clear: bool = True) -> bool:
    """True when a kernel of the f16_operands policy stored an inf / nan on ``device`` since the last clear.  Waits for the device's
    current stream (one 4-byte D2H copy): call it once per encode, next to the copy of the embeddings."""
    import ctypes
    flag = ctypes.c_int(0)
    with torch.cuda.device(device):
        check(_lib.load().grit_f16_overflow_flag(ctypes.byref(flag), int(clear), _stream()), "grit_f16_overflow_flag")
    return bool(flag.value)


def mask_pack(mask: torch.Tensor) -> torch.Tensor:
    B, S = mask.shape
    bits = torch.empty((B, (S + 63) // 64), dtype=I64, device=mask.device)   # uint64 payload in int64 storage
    check(_lib.load().grit_mask_pack(_chk(mask, I64, "mask"), _chk(bits, I64, "bits"), B, S, _stream()), "grit_mask_pack")
    return bits


def attn_bidir(qkv: torch.Tensor, key_bits: torch.Tensor, B: int, S: int, nq: int, nkv: int, d: int,
               out: torch.Tensor | None = None, lse: torch.Tensor | None = None, scale: float | None = None,
               causal: bool = False, window: int = 0) -> torch.Tensor:
    T, stride = qkv.shape
    assert T == B * S
    if out is None:
        out = torch.empty((T, nq * d), dtype=qkv.dtype, device=qkv.device)
    if scale is None:
        scale = d ** -0.5
    ev = _timer.span("attn_bidir_fwd", 4.0 * B * nq * S * S * d) if _timer is not None else None
    if ev:
        ev[0].record()
    fn, wa = _attn_entry("fwd", causal, window)
    odt = BF16
    if qkv.dtype == F16:            # the fp16-operand policies
        if window and window > 0 and not causal:
            raise ValueError("a sliding window applies to causal attention only")
        fn, wa, odt = (_lib.load().grit_attn_causal_f16_fwd, (int(window or 0),), F16) if causal else (_lib.load().grit_attn_bidir_f16_fwd, (), F16)
    check(fn(_chk(qkv, odt, "qkv"), _chk(key_bits, I64, "key_bits"), _chk(out, odt, "out"),
             0 if lse is None else _chk(lse, F32, "lse"), B, S, nq, nkv, d, stride, out.stride(0), float(scale), *wa, _stream()), "grit_attn_bidir_fwd")
    if ev:
        ev[1].record()
    return out


def pool_norm(hidden: torch.Tensor, mask: torch.Tensor, method: str, normalize: bool, instr_len: torch.Tensor | None = None,
              inv_norm: torch.Tensor | None = None) -> torch.Tensor:
    if method not in POOL_MODES:
        raise NotImplementedError(f"Unknown pooling method: {method}")     # gritlm/gritlm.py:215
    B, S, H = hidden.shape
    out = torch.empty((B, H), dtype=F32, device=hidden.device)
    check(_lib.load().grit_pool_norm_fwd(_chk(hidden, BF16, "hidden"), _chk(mask, I64, "mask"),
                                         0 if instr_len is None else _chk(instr_len, I32, "instr_len"), _chk(out, F32, "out"),
                                         0 if inv_norm is None else _chk(inv_norm, F32, "inv_norm"), B, S, H, POOL_MODES[method],
                                         int(normalize), _stream()), "grit_pool_norm_fwd")
    return out


def pool_norm_bwd(y: torch.Tensor, dy: torch.Tensor, inv_norm: torch.Tensor | None, mask: torch.Tensor, method: str, normalize: bool,
                  S: int, instr_len: torch.Tensor | None = None) -> torch.Tensor:
    B, H = y.shape
    dh = torch.empty((B, S, H), dtype=BF16, device=y.device)
    check(_lib.load().grit_pool_norm_bwd(_chk(y, F32, "y"), _chk(dy, F32, "dy"), 0 if inv_norm is None else _chk(inv_norm, F32, "inv_norm"),
                                         _chk(mask, I64, "mask"), 0 if instr_len is None else _chk(instr_len, I32, "instr_len"),
                                         _chk(dh, BF16, "dhidden"), B, S, H, POOL_MODES[method], int(normalize), _stream()),
          "grit_pool_norm_bwd")
    return dh


def infonce(q: torch.Tensor, p: torch.Tensor, temperature: float, q_off: int = 0, nq_loc: int | None = None, p_off: int = 0,
            np_loc: int | None = None, want_grad: bool = True):
    """Returns (loss[1] fp32, dq [nq_loc,H] | None, dp [np_loc,H] | None)."""
    Nq, H = q.shape
    Np = p.shape[0]
    nq_loc = Nq if nq_loc is None else nq_loc
    np_loc = Np if np_loc is None else np_loc
    scores = torch.empty((Nq, Np), dtype=F32, device=q.device)
    loss_buf = torch.empty((1 + Nq,), dtype=F32, device=q.device)        # [0] mean loss, [1 + i] the term of row i
    loss, loss_rows = loss_buf[:1], loss_buf[1:]
    dq = torch.empty((nq_loc, H), dtype=F32, device=q.device) if want_grad else None
    dp = torch.empty((np_loc, H), dtype=F32, device=q.device) if want_grad else None
    check(_lib.load().grit_infonce_rows_fwd_bwd(_chk(q, F32, "q"), _chk(p, F32, "p"), 1.0 / float(temperature), _chk(scores, F32, "scores"),
                                                _chk(loss, F32, "loss"), _chk(loss_rows, F32, "loss_rows"),
                                                0 if dq is None else dq.data_ptr(), 0 if dp is None else dp.data_ptr(),
                                                Nq, Np, H, q_off, nq_loc, p_off, np_loc, _stream()), "grit_infonce_rows_fwd_bwd")
    return loss, dq, dp


def transpose(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """[R,C] -> [C,R]; ``out`` may be a wider buffer [C, >=R] (zero-padded K for the wgrad GEMM)."""
    R, Cc = x.shape
    ret_view = None
    if out is None:
        out = torch.empty((Cc, (R + 7) // 8 * 8), dtype=BF16, device=x.device)
        ret_view = out if out.shape[1] == R else out[:, :R]
    assert out.shape[0] == Cc and out.shape[1] >= R and x.stride(1) == 1 and out.stride(1) == 1
    if x.dtype != BF16 or out.dtype != BF16 or not x.is_cuda:
        raise TypeError("transpose: bf16 CUDA tensors expected")
    check(_lib.load().grit_transpose_bf16(x.data_ptr(), out.data_ptr(), R, Cc, x.stride(0), out.stride(0), _stream()),
          "grit_transpose_bf16")
    return out if ret_view is None else ret_view


def rmsnorm_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float, dw: torch.Tensor, dres: torch.Tensor | None = None,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """dx (+ dres) -> out; dw [H] fp32 += sum_t dy * xhat."""
    H = x.shape[-1]
    T = x.numel() // H
    if out is None:
        out = torch.empty_like(x)
    rows = _lib.load().grit_rmsnorm_bwd_workspace_rows(T)
    part = torch.empty((rows, H), dtype=F32, device=x.device)
    check(_lib.load().grit_rmsnorm_bwd(_chk(dy, BF16, "dy"), _chk(x, BF16, "x"), _chk(w, BF16, "w"),
                                       0 if dres is None else _chk(dres, BF16, "dres"), _chk(out, BF16, "dx"), part.data_ptr(),
                                       _chk(dw, F32, "dw"), T, H, float(eps), _stream()), "grit_rmsnorm_bwd")
    return out


def swiglu(gu: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    T, I2 = gu.shape
    I = I2 // 2
    if out is None:
        out = torch.empty((T, I), dtype=BF16, device=gu.device)
    check(_lib.load().grit_swiglu_fwd(_chk(gu, BF16, "gu"), _chk(out, BF16, "act"), T, I, _stream()), "grit_swiglu_fwd")
    return out


def swiglu_bwd(gu: torch.Tensor, dact: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    T, I2 = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    check(_lib.load().grit_swiglu_bwd(_chk(gu, BF16, "gu"), _chk(dact, BF16, "dact"), _chk(out, BF16, "dgu"), T, I2 // 2, _stream()),
          "grit_swiglu_bwd")
    return out


def attn_bidir_bwd(qkv: torch.Tensor, key_bits: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, B: int, S: int,
                   nq: int, nkv: int, d: int, dqkv: torch.Tensor | None = None, scale: float | None = None,
                   causal: bool = False, window: int = 0) -> torch.Tensor:
    T, stride = qkv.shape
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if scale is None:
        scale = d ** -0.5
    delta = torch.empty((B, nq, S), dtype=F32, device=qkv.device)
    fn, wa = _attn_entry("bwd", causal, window)
    check(fn(_chk(qkv, BF16, "qkv"), _chk(key_bits, I64, "key_bits"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse"),
             delta.data_ptr(), _chk(dqkv, BF16, "dqkv"), B, S, nq, nkv, d, stride, out.stride(0), float(scale), *wa, _stream()),
          "grit_attn_bidir_bwd")
    return dqkv


def attn_bidir_varlen_bwd(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, out: torch.Tensor, dout: torch.Tensor,
                          lse: torch.Tensor, nq: int, nkv: int, d: int, dqkv: torch.Tensor | None = None,
                          scale: float | None = None, causal: bool = False, window: int = 0) -> torch.Tensor:
    T, stride = qkv.shape
    B = cu_seqlens.numel() - 1
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if scale is None:
        scale = d ** -0.5
    delta = torch.empty((T, nq), dtype=F32, device=qkv.device)
    fn, wa = _attn_entry("varlen_bwd", causal, window)
    check(fn(_chk(qkv, BF16, "qkv"), _chk(cu_seqlens, I32, "cu_seqlens"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse"),
             delta.data_ptr(), _chk(dqkv, BF16, "dqkv"), B, int(max_len), T, nq, nkv, d, stride, out.stride(0), float(scale), *wa, _stream()),
          "grit_attn_bidir_varlen_bwd")
    return dqkv


def gemv(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor | None = None, epilogue: int = EPI_STORE,
         residual: torch.Tensor | None = None, out16: torch.Tensor | None = None) -> torch.Tensor:
    """out[B,N] = x[B,K] @ w[N,K]^T for B <= 8 rows (decode); same epilogues as gemm_nt.  fp16 ``w`` (and ``x``): the fp16-operand form --
    STORE / RESIDUAL write fp32 (``residual`` is the fp32 stream; ``out16``: additionally its fp16 rounding, the next GEMV's operand),
    SWIGLU writes the fp16 activation."""
    B, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if w.dtype == F16:
        odt = F16 if epilogue == EPI_SWIGLU else F32
        if out is None:
            out = torch.empty((B, n_out), dtype=odt, device=x.device)
        check(_lib.load().grit_gemv_f16(_chk2d(x, F16, "x"), _chk2d(w, F16, "w"), _chk2d(out, odt, "out"), B, N, K, x.stride(0), w.stride(0),
                                        out.stride(0), epilogue, 0 if residual is None else _chk2d(residual, F32, "residual"),
                                        0 if residual is None else residual.stride(0), 0 if out16 is None else _chk2d(out16, F16, "out16"),
                                        0 if out16 is None else out16.stride(0), _stream()), "grit_gemv_f16")
        return out
    if out is None:
        out = torch.empty((B, n_out), dtype=BF16, device=x.device)
    check(_lib.load().grit_gemv_bf16(_chk2d(x, BF16, "x"), _chk2d(w, BF16, "w"), _chk2d(out, BF16, "out"), B, N, K, x.stride(0), w.stride(0),
                                     out.stride(0), epilogue, 0 if residual is None else _chk2d(residual, BF16, "residual"),
                                     0 if residual is None else residual.stride(0), _stream()), "grit_gemv_bf16")
    return out


def gemv_expert(x: torch.Tensor, w_stack: torch.Tensor, expert: torch.Tensor, out: torch.Tensor | None = None, epilogue: int = EPI_STORE) -> torch.Tensor:
    """x [B,K] @ w_stack[expert[0]]^T for a stack [E,N,K]: the expert index is read on the DEVICE (``expert``: an int32 tensor or view whose
    first element is the index), so a sparse-MoE decode step can be captured in a graph.  STORE / SWIGLU; fp16 stack: the fp16 formats."""
    B, K = x.shape
    E, N, _ = w_stack.shape
    f16 = w_stack.dtype == F16
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    odt = (F16 if epilogue == EPI_SWIGLU else F32) if f16 else BF16
    if out is None:
        out = torch.empty((B, n_out), dtype=odt, device=x.device)
    if expert.dtype != I32 or not expert.is_cuda:
        raise TypeError("gemv_expert: expert must be an int32 CUDA tensor")
    fn = _lib.load().grit_gemv_f16_expert if f16 else _lib.load().grit_gemv_bf16_expert
    check(fn(_chk2d(x, F16 if f16 else BF16, "x"), _chk3d(w_stack, F16 if f16 else BF16, "w_stack"), _chk2d(out, odt, "out"), expert.data_ptr(),
             w_stack.stride(0), B, N, K, x.stride(0), w_stack.stride(1), out.stride(0), epilogue, _stream()), "grit_gemv_expert")
    return out


def moe_decode_combine_f32(h: torch.Tensor, h16: torch.Tensor, y: torch.Tensor, weights: torch.Tensor):
    """h[b] += weights[b,0] y[2b] + weights[b,1] y[2b+1] (fp32, in place) and h16 = fp16(h): the sparse-MoE decode step on fp16 operands."""
    B, H = h.shape
    check(_lib.load().grit_moe_decode_combine_f32(_chk(h, F32, "h"), _chk(h16, F16, "h16"), _chk(y, F32, "y"), _chk(weights, F32, "weights"), B, H,
                                                  _stream()), "grit_moe_decode_combine_f32")


def moe_router_top2(x: torch.Tensor, gate_w: torch.Tensor, experts: torch.Tensor, weights: torch.Tensor, ln_w: torch.Tensor | None = None,
                    eps: float = 0.0):
    """The routing decision alone into caller-owned buffers (experts [T,2] int32, weights [T,2] fp32) -- no index building, no host
    round trip: the sparse-MoE decode step.  bf16 x: the reference's router on the normalised bf16 rows; fp32 x with ``ln_w``: the
    fp16-operand policy's router on the residual stream itself (norm folded in, nothing rounded)."""
    T, H = x.shape
    E = gate_w.shape[0]
    if x.dtype == F32:
        check(_lib.load().grit_moe_router_top2_f32(_chk(x, F32, "h"), _chk(ln_w, BF16, "ln_w"), float(eps), _chk(gate_w, BF16, "gate_w"),
                                                   _chk(experts, I32, "experts"), _chk(weights, F32, "weights"), T, H, E, _stream()),
              "grit_moe_router_top2_f32")
    else:
        check(_lib.load().grit_moe_router_top2(_chk(x, BF16, "x"), _chk(gate_w, BF16, "gate_w"), _chk(experts, I32, "experts"),
                                               _chk(weights, F32, "weights"), T, H, E, _stream()), "grit_moe_router_top2")


def rmsnorm_gemv(x: torch.Tensor, ln_w: torch.Tensor, eps: float, w: torch.Tensor, out: torch.Tensor | None = None,
                 epilogue: int = EPI_STORE, deferred: bool = False) -> torch.Tensor:
    """gemv(rmsnorm(x, ln_w, eps), w) in one launch (decode step).  ``deferred``: the row scale multiplies the finished dot products
    (no pass over x in front of the weight stream; x_n is not rounded to bf16: not the bits of the two launches).  fp16 ``w``: x is the
    fp16 copy of the residual stream, the deferred form only; STORE writes fp32, SWIGLU the fp16 activation."""
    B, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if w.dtype == F16:
        if not deferred:
            raise ValueError("rmsnorm_gemv on fp16 weights: the deferred form only")
        odt = F16 if epilogue == EPI_SWIGLU else F32
        if out is None:
            out = torch.empty((B, n_out), dtype=odt, device=x.device)
        check(_lib.load().grit_rmsnorm_gemv_f16_deferred(_chk2d(x, F16, "x"), _chk(ln_w, BF16, "ln_w"), float(eps), _chk2d(w, F16, "w"),
                                                         _chk2d(out, odt, "out"), B, N, K, x.stride(0), w.stride(0), out.stride(0), epilogue,
                                                         _stream()), "grit_rmsnorm_gemv_f16_deferred")
        return out
    if out is None:
        out = torch.empty((B, n_out), dtype=BF16, device=x.device)
    fn = _lib.load().grit_rmsnorm_gemv_bf16_deferred if deferred else _lib.load().grit_rmsnorm_gemv_bf16
    check(fn(_chk2d(x, BF16, "x"), _chk(ln_w, BF16, "ln_w"), float(eps), _chk2d(w, BF16, "w"), _chk2d(out, BF16, "out"),
             B, N, K, x.stride(0), w.stride(0), out.stride(0), epilogue, _stream()), "grit_rmsnorm_gemv_bf16")
    return out


def rope_kv_append(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor,
                   nq: int, nkv: int, d: int):
    B, _, Lmax, _ = cache_k.shape
    assert cos.shape[0] >= Lmax
    check(_lib.load().grit_rope_kv_append(_chk2d(qkv, BF16, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"), _chk(cache_k, BF16, "cache_k"),
                                          _chk(cache_v, BF16, "cache_v"), _chk(lens, I32, "lens"), B, nq, nkv, d, Lmax, qkv.stride(0), _stream()),
          "grit_rope_kv_append")


def kv_append(qkv: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor, nq: int, nkv: int, d: int):
    B, _, Lmax, _ = cache_k.shape
    check(_lib.load().grit_kv_append(_chk2d(qkv, BF16, "qkv"), _chk(cache_k, BF16, "cache_k"), _chk(cache_v, BF16, "cache_v"),
                                     _chk(lens, I32, "lens"), B, nq, nkv, d, Lmax, qkv.stride(0), _stream()), "grit_kv_append")


def attn_decode_workspace(B: int, nq: int, nkv: int, Lmax: int, device) -> torch.Tensor:
    return torch.empty((int(_lib.load().grit_attn_decode_workspace_floats(B, nq, nkv, Lmax)),), dtype=F32, device=device)


def attn_decode(q: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor,
                nq: int, nkv: int, d: int, scale: float | None = None):
    B, _, Lmax, _ = cache_k.shape
    check(_lib.load().grit_attn_decode(_chk2d(q, BF16, "q"), _chk(cache_k, BF16, "cache_k"), _chk(cache_v, BF16, "cache_v"), _chk(lens, I32, "lens"),
                                       _chk2d(out, BF16, "out"), _chk(workspace, F32, "workspace"), B, nq, nkv, d, Lmax, q.stride(0), out.stride(0),
                                       float(d ** -0.5 if scale is None else scale), _stream()), "grit_attn_decode")
    return out


def attn_decode_rope(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor,
                     out: torch.Tensor, workspace: torch.Tensor, nq: int, nkv: int, d: int, scale: float | None = None):
    """rope_kv_append + attn_decode in one launch (qkv: the raw fused projection of the new token; the caches are appended to)."""
    B, _, Lmax, _ = cache_k.shape
    assert cos.shape[0] >= Lmax
    if cache_k.dtype == F16:    # fp16-operand form: fp32 q|k|v row, fp16 caches and ctx
        check(_lib.load().grit_attn_decode_rope_f16(_chk2d(qkv, F32, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"), _chk(cache_k, F16, "cache_k"),
                                                    _chk(cache_v, F16, "cache_v"), _chk(lens, I32, "lens"), _chk2d(out, F16, "out"),
                                                    _chk(workspace, F32, "workspace"), B, nq, nkv, d, Lmax, qkv.stride(0), out.stride(0),
                                                    float(d ** -0.5 if scale is None else scale), _stream()), "grit_attn_decode_rope_f16")
        return out
    check(_lib.load().grit_attn_decode_rope(_chk2d(qkv, BF16, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"), _chk(cache_k, BF16, "cache_k"),
                                            _chk(cache_v, BF16, "cache_v"), _chk(lens, I32, "lens"), _chk2d(out, BF16, "out"),
                                            _chk(workspace, F32, "workspace"), B, nq, nkv, d, Lmax, qkv.stride(0), out.stride(0),
                                            float(d ** -0.5 if scale is None else scale), _stream()), "grit_attn_decode_rope")
    return out


def rope_kv_append_rows(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor,
                        cache_row: torch.Tensor, nq: int, nkv: int, d: int):
    """rope_kv_append for V rows that are tokens of the caches' B <= V sequences (row v: sequence cache_row[v], position lens[v]); fp16
    caches: the fp16-operand formats (fp32 rows)."""
    V = qkv.shape[0]
    _, _, Lmax, _ = cache_k.shape
    f16 = cache_k.dtype == F16
    assert cos.shape[0] >= Lmax and lens.numel() == V and cache_row.numel() == V
    check(_lib.load().grit_rope_kv_append_rows(_chk2d(qkv, F32 if f16 else BF16, "qkv"), _chk(cos, F32, "cos"), _chk(sin, F32, "sin"),
                                               _chk(cache_k, cache_k.dtype, "cache_k"), _chk(cache_v, cache_k.dtype, "cache_v"), _chk(lens, I32, "lens"),
                                               _chk(cache_row, I32, "cache_row"), V, nq, nkv, d, Lmax, qkv.stride(0), int(f16), _stream()),
          "grit_rope_kv_append_rows")


def attn_decode_rows(q: torch.Tensor, cache_k: torch.Tensor, cache_v: torch.Tensor, lens: torch.Tensor, cache_row: torch.Tensor, out: torch.Tensor,
                     workspace: torch.Tensor, nq: int, nkv: int, d: int, scale: float | None = None):
    """attn_decode for V rows that are tokens of the caches' sequences: row v attends to keys 0 .. lens[v] of sequence cache_row[v]."""
    V = q.shape[0]
    _, _, Lmax, _ = cache_k.shape
    f16 = cache_k.dtype == F16
    check(_lib.load().grit_attn_decode_rows(_chk2d(q, F32 if f16 else BF16, "q"), _chk(cache_k, cache_k.dtype, "cache_k"), _chk(cache_v, cache_k.dtype, "cache_v"),
                                            _chk(lens, I32, "lens"), _chk(cache_row, I32, "cache_row"), _chk2d(out, F16 if f16 else BF16, "out"),
                                            _chk(workspace, F32, "workspace"), V, nq, nkv, d, Lmax, q.stride(0), out.stride(0),
                                            float(d ** -0.5 if scale is None else scale), int(f16), _stream()), "grit_attn_decode_rows")
    return out


def argmax_advance(logits: torch.Tensor, next_ids: torch.Tensor, lens: torch.Tensor | None = None, history: torch.Tensor | None = None,
                   step: torch.Tensor | None = None):
    B, V = logits.shape
    if logits.dtype == F32:
        check(_lib.load().grit_argmax_advance_f32(_chk2d(logits, F32, "logits"), logits.stride(0), V, _chk(next_ids, I64, "next"),
                                                  0 if lens is None else _chk(lens, I32, "lens"), 0 if history is None else _chk(history, I64, "history"),
                                                  0 if history is None else history.stride(0), 0 if step is None else _chk(step, I32, "step"), B,
                                                  _stream()), "grit_argmax_advance_f32")
        return next_ids
    check(_lib.load().grit_argmax_advance(_chk2d(logits, BF16, "logits"), logits.stride(0), V, _chk(next_ids, I64, "next"),
                                          0 if lens is None else _chk(lens, I32, "lens"), 0 if history is None else _chk(history, I64, "history"),
                                          0 if history is None else history.stride(0), 0 if step is None else _chk(step, I32, "step"), B,
                                          _stream()), "grit_argmax_advance")
    return next_ids


def knn_topk(queries: torch.Tensor, embeddings: torch.Tensor, k: int, transposed: bool = False):
    """Brute-force inner-product search: (scores [Q,k] fp32 descending, indices [Q,k] int64).  ``embeddings`` is [N,H], or with
    ``transposed=True`` the reference index layout [H,N] (rag/index.py:141); any strides."""
    if queries.dtype != F32 or embeddings.dtype != F32:
        raise TypeError("knn_topk: fp32 tensors expected")
    Q, H = queries.shape
    N = embeddings.shape[1] if transposed else embeddings.shape[0]
    assert (embeddings.shape[0] if transposed else embeddings.shape[1]) == H
    sn, sh = (embeddings.stride(1), embeddings.stride(0)) if transposed else (embeddings.stride(0), embeddings.stride(1))
    ws = torch.empty((int(_lib.load().grit_knn_workspace_bytes(Q, N, k)),), dtype=torch.uint8, device=queries.device)
    scores = torch.empty((Q, k), dtype=F32, device=queries.device)
    index = torch.empty((Q, k), dtype=I64, device=queries.device)
    if not embeddings.is_cuda:
        raise _lib.GritHipError("knn_topk: tensors must live on the GPU; the native path has no CPU fallback")
    check(_lib.load().grit_knn_topk(_chk(queries, F32, "queries"), embeddings.data_ptr(), Q, N, H, sn, sh, int(k), ws.data_ptr(), scores.data_ptr(),
                                    index.data_ptr(), _stream()), "grit_knn_topk")
    return scores, index


def ce_fwd(logits: torch.Tensor, labels: torch.Tensor):
    """(lse [T], loss_row [T]) fp32 of bf16 logits [T,V] against int64 labels (-100 ignored)."""
    T, V = logits.shape
    lse = torch.empty((T,), dtype=F32, device=logits.device)
    loss_row = torch.empty((T,), dtype=F32, device=logits.device)
    check(_lib.load().grit_ce_fwd(_chk2d(logits, BF16, "logits"), logits.stride(0), _chk(labels, I64, "labels"), lse.data_ptr(),
                                  loss_row.data_ptr(), T, V, _stream()), "grit_ce_fwd")
    return lse, loss_row


def ce_bwd_(logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, scale: float, dev_scale: torch.Tensor | None = None):
    """In place: logits <- (softmax - onehot) * scale * dev_scale (bf16)."""
    T, V = logits.shape
    check(_lib.load().grit_ce_bwd(_chk2d(logits, BF16, "logits"), logits.stride(0), _chk(labels, I64, "labels"), _chk(lse, F32, "lse"),
                                  0 if dev_scale is None else _chk(dev_scale, F32, "dev_scale"), float(scale), T, V, _stream()),
          "grit_ce_bwd")
    return logits


def embed_scatter_add(dh: torch.Tensor, ids: torch.Tensor, grad: torch.Tensor):
    """Embedding weight gradient: grad[ids[t]] += dh[t] (bf16 table, fp32 sums in token order; deterministic, no atomics).  The stable
    sort of the ids is index plumbing and stays in torch."""
    V, H = grad.shape
    T = ids.numel()
    sorted_ids, order = torch.sort(ids.reshape(-1).clamp(0, V - 1), stable=True)
    check(_lib.load().grit_embed_scatter_add_sorted(_chk(dh, BF16, "dh"), _chk(sorted_ids, I64, "sorted_ids"), _chk(order, I64, "order"),
                                                    _chk(grad, BF16, "grad"), T, H, V, _stream()), "grit_embed_scatter_add_sorted")
    return grad


def accum_bf16_from_f32(acc: torch.Tensor, x: torch.Tensor):
    assert acc.numel() == x.numel()
    check(_lib.load().grit_accum_bf16_from_f32(_chk(acc, BF16, "acc"), _chk(x, F32, "x"), acc.numel(), _stream()), "grit_accum_bf16_from_f32")
    return acc


# ---------------------------------------------------------------------------------------------------------------------
# Device guard.  A HIP launch goes to the calling thread's CURRENT device and `_stream()` is that device's current stream, while the
# reference API lets the caller put the model anywhere (GritLM(..., device="cuda:1"), gritlm/gritlm.py:24,57): every public op therefore
# runs with the device of its first tensor argument made current, so kernels, the stream they are ordered on and torch's own ops on
# that tensor all agree.  (One `current_device()` read per call when the devices already match.)
def _on_tensor_device(fn):
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        for a in (*args, *kwargs.values()):          # first tensor, positional or keyword
            if torch.is_tensor(a):
                if a.is_cuda and a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return guarded


# Every public function of this module that launches kernels is wrapped; host-only helpers are listed, so a new op is guarded by default
# and tests/test_abi.py::test_every_public_op_is_device_guarded fails if a launcher slips through.
_HOST_ONLY = ("set_timer", "check", "attn_decode_workspace", "f16_overflow_flag")
for _name, _fn in list(globals().items()):
    if callable(_fn) and getattr(_fn, "__module__", None) == __name__ and not _name.startswith("_") and not isinstance(_fn, type) \
            and _name not in _HOST_ONLY:
        globals()[_name] = _on_tensor_device(_fn)
        globals()[_name]._device_guarded = True
del _name, _fn
