"""Native greedy generation on the encoder engine's weights: prompt prefill + token-by-token decode on HIP kernels.

Replaces, for the RAG doc-caching flow of the reference (rag/eval.py:237-302: ``model.generate(inputs, past_key_values=kv_cache)`` where
``kv_cache`` came from ``encode(..., get_cache=True)``, gritlm/gritlm.py:131-140), the Hugging Face decode loop: every projection of a
decode step is an HBM-bound GEMV (``grit_gemv_bf16``), attention reads the sequence's KV once (``grit_attn_decode``), the new token's
K/V are appended in place, sampling is a device-side argmax -- and the whole step (5 launches per layer) is captured in ONE HIP
graph, so the host only replays it.  Greedy decoding only (``do_sample=False``), batch <= 8, head_dim 128.

Precision (round 6): the decoder follows its engine's policy.  Under ``"bf16"`` / ``"fp32_residual"`` the step is the reference's bf16
arithmetic (teacher-forced logits within ~1e-3 (1 - cos) of the fp32 module: the bf16 level).  Under ``"f16_operands"`` / ``"f16_stream"`` it
runs on the engine's fp16 weight copies around an fp32 residual stream (``grit_gemv_f16`` ...; formats in include/gritlm_hip.h): the
continuation of ``encode(get_cache=True)`` at the north-star's own level (fp16 K/V: same bytes per token as bf16).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import EPI_RESIDUAL, EPI_SWIGLU
from ._lib import GritHipError
from .encoder import F16_POLICIES, MistralEncoderEngine, rope_tables

BF16, F16, F32, I32, I64 = torch.bfloat16, torch.float16, torch.float32, torch.int32, torch.int64


def _layers_of(cache):
    """per-layer (k, v) [B, nkv, S, d] from the installed transformers' cache object or a legacy tuple of tuples."""
    if hasattr(cache, "layers"):
        return [(l.keys, l.values) for l in cache.layers]
    return [(l[0], l[1]) for l in cache]


class MistralDecoder:
    def __init__(self, engine: MistralEncoderEngine, lm_head: torch.Tensor):
        self.eng, self.cfg, self.device = engine, engine.cfg, engine.device
        self.moe = bool(engine.cfg.num_local_experts)      # sparse-MoE (Mixtral) layers: router + the two chosen experts per row (_step_moe)
        self.lm_head = lm_head.detach().to(device=self.device, dtype=BF16).contiguous()
        self.use_graph = True
        # which RMSNorms ride inside the following GEMV (grit_rmsnorm_gemv_bf16): "all" (input_layernorm -> q|k|v, post_attention_layernorm
        # -> gate|up, final norm -> lm_head), "qkv" (the MLP's and the final norm get their own launches) or "none"; GRIT_DECODE_FUSE_NORM for A/B runs
        # (tools/decode_variants.sh).  Same bits in these three forms: the fused kernel applies the reference's two roundings.
        # "deferred" / "deferred_mlp": all three norms / the MLP's and the final one ride in the DEFERRED form (grit_rmsnorm_gemv_bf16_deferred:
        # the row scale multiplies the finished dot products, no second pass over the row; x_n is not rounded to bf16 -- one rounding fewer
        # than the reference's arithmetic).
        import os
        # The exact forms (round 5, tools/decode_variants.sh on one box: all 3.49, qkv 3.25, none 3.32 ms per token): the 7168 workgroups of
        # the gate|up GEMV -- and the 8000 of lm_head -- each re-derive the row's RMS when the norm is fused, which costs more than the
        # one-row launch it saves; the q|k|v GEMV is a single workgroup wave deep and keeps the fusion.
        # Default "deferred" (tools/decode_norm_ab.sh, one box: qkv 3.149, deferred_mlp 3.014, deferred 2.904, all 3.410 ms per token).
        # It is NOT the reference's bf16 arithmetic bit for bit (one rounding fewer per norm; parity is held where it is bounded: greedy
        # tokens of the reference fixtures, teacher-forced logits vs fp32) -- GRIT_DECODE_FUSE_NORM=qkv restores the exact bits; both are
        # the same arithmetic at every batch size 1..8.
        self.fuse_norm = os.environ.get("GRIT_DECODE_FUSE_NORM", "deferred")
        # None: follow the engine's policy (fp16 operands under its fp16 policies); "bf16" / "f16" pin the decode arithmetic
        self.precision = os.environ.get("GRIT_DECODE_PRECISION") or None
        self.on_overflow = "raise"                # fp16 operands, a value beyond the range: "raise" | "bf16" (repeat the call in bf16)
        # prompt tokens on top of past_key_values: all at once through the step's kernels (same bits as one token per step; False or
        # GRIT_DECODE_PROMPT_CHUNK=0: the token-by-token loop).  Above `prompt_chunk_max_rows` rows (the attention workspace grows with
        # rows x splits) the loop is used.
        self.prompt_chunk = os.environ.get("GRIT_DECODE_PROMPT_CHUNK", "1") != "0" and not self.moe
        self.prompt_chunk_max_rows = 512
        self._lm_head16 = None
        self.last_precision = None                # what the last generate() call ran in ("bf16" | "f16"; "bf16 (f16 overflow)" after a fallback)

    def _f16(self) -> bool:
        if self.precision not in (None, "bf16", "f16"):
            raise ValueError(f"MistralDecoder.precision={self.precision!r}: None (follow the engine), 'bf16' or 'f16'")
        return self.precision == "f16" or (self.precision is None and self.eng.precision in F16_POLICIES)

    def _lm_head_f16(self):
        key = (self.lm_head.data_ptr(), self.lm_head._version)
        if self._lm_head16 is None or self._lm_head16[0] != key:
            w = self.lm_head.to(F16)
            if bool(torch.isinf(w).any()):
                raise GritHipError("native decode on fp16 operands: lm_head weights exceed the fp16 range")
            self._lm_head16 = (key, w)
        return self._lm_head16[1]

    def _step_f16(self, st):
        """One decode step on fp16 operands: fp32 stream / q|k|v row / logits; fp16 weights, K/V and GEMV operand rows (h16 = the stream's
        rounding, written by the residual epilogues; ctx; act); every norm in the deferred form (the row scale multiplies fp32 dot products)."""
        c, e = self.cfg, self.eng
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        h, h16, qkv, ctx, act = st["h"], st["h16"], st["qkv"], st["ctx"], st["act"]
        ops.embed_gather(e.embed, st["next"], out=h)
        h16.copy_(h)                                   # (bf16 embedding rows: exact in fp16 inside its range)
        for li, L in enumerate(e.layers):
            wqkv, wo, wgu, wdown = e._f16_weights(L)
            ck, cv = st["cache"][li]
            ops.rmsnorm_gemv(h16, L.ln1, eps, wqkv, out=qkv, deferred=True)
            ops.attn_decode_rope(qkv, st["cos"], st["sin"], ck, cv, st["lens"], ctx, st["ws"], nq, nkv, d)
            ops.gemv(ctx, wo, out=h, epilogue=EPI_RESIDUAL, residual=h, out16=h16)
            ops.rmsnorm_gemv(h16, L.ln2, eps, wgu, out=act, epilogue=EPI_SWIGLU, deferred=True)
            ops.gemv(act, wdown, out=h, epilogue=EPI_RESIDUAL, residual=h, out16=h16)
        ops.rmsnorm_gemv(h16, e.norm, eps, self._lm_head_f16(), out=st["logits"], deferred=True)

    def _step_moe(self, st):
        """One decode step of a sparse-MoE (Mixtral) model, MixtralSparseMoeBlock.forward (scripts/modeling_mixtral_gritlm.py:839-882) at 1..8
        rows: the attention half as in the dense step; then the router's top-2 decision ON THE DEVICE (experts [B,2], weights [B,2]) and,
        per row and choice, the two GEMVs of the chosen expert -- ``grit_gemv_*_expert`` reads the expert index from device memory, so the
        step is still one HIP graph and streams 2 of the 8 experts' weights per row -- and the weighted combine.
        bf16: the reference's arithmetic (exact RMSNorm for the block input, router on those bf16 rows, ``grit_moe_combine`` rounding
        points).  fp16 operands: the router reads the fp32 stream (norm folded in, nothing rounded), the experts run on fp16 copies, the
        combine adds fp32 expert outputs into the fp32 stream."""
        c, e = self.cfg, self.eng
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        f16 = st["f16"]
        h, h16, qkv, ctx, x, act2, y2 = st["h"], st["h16"], st["qkv"], st["ctx"], st["x"], st["act2"], st["y2"]
        experts, weights, rows_c = st["experts"], st["weights"], st["rows_c"]
        B = h.shape[0]
        ops.embed_gather(e.embed, st["next"], out=h)
        if f16:
            h16.copy_(h)
        xin = h16 if f16 else h
        for li, L in enumerate(e.layers):
            wqkv, wo, w13, w2 = e._f16_weights(L) if f16 else (L.wqkv, L.wo, L.w13, L.w2)
            ck, cv = st["cache"][li]
            ops.rmsnorm_gemv(xin, L.ln1, eps, wqkv, out=qkv, deferred=True)
            ops.attn_decode_rope(qkv, st["cos"], st["sin"], ck, cv, st["lens"], ctx, st["ws"], nq, nkv, d)
            if f16:
                ops.gemv(ctx, wo, out=h, epilogue=EPI_RESIDUAL, residual=h)
                ops.moe_router_top2(h, L.wgate, experts, weights, ln_w=L.ln2, eps=eps)
            else:
                ops.gemv(ctx, wo, out=h, epilogue=EPI_RESIDUAL, residual=h)
            ops.rmsnorm(h, L.ln2, eps, out=x)              # (bf16 rows: the reference's x_n; fp16 rows under fp16 operands)
            if not f16:
                ops.moe_router_top2(x, L.wgate, experts, weights)
            for b in range(B):
                for k in (0, 1):
                    r = 2 * b + k
                    ops.gemv_expert(x[b:b + 1], w13, experts[b, k:k + 1], out=act2[r:r + 1], epilogue=EPI_SWIGLU)
                    ops.gemv_expert(act2[r:r + 1], w2, experts[b, k:k + 1], out=y2[r:r + 1])
            if f16:
                ops.moe_decode_combine_f32(h, h16, y2, weights)
            else:
                ops.moe_combine(y2, rows_c, weights, h, out=h)
        ops.rmsnorm_gemv(xin, e.norm, eps, self._lm_head_f16() if f16 else self.lm_head, out=st["logits"], deferred=True)

    # ------------------------------------------------------------------ one decode step (all sizes static, lengths on the device)
    def _step(self, st):
        if self.moe:
            return self._step_moe(st)
        if st["f16"]:
            return self._step_f16(st)
        c, e = self.cfg, self.eng
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        h, qkv, ctx, act = st["h"], st["qkv"], st["ctx"], st["act"]
        ops.embed_gather(e.embed, st["next"], out=h)
        # beyond 2 rows the EXACT fused forms give the norm its own launch (they re-derive every row's RMS in every workgroup; same bits
        # either way).  The DEFERRED forms keep their arithmetic at every batch size (round 6): x_n is not rounded to bf16 there,
        # so switching to the un-fused norm at 3+ rows would make a row's logits -- and near-tie argmaxes -- depend on how many rows share
        # the step; the price is the deferred form's per-row work in the GEMV at 3..8 rows (batch 4: 4.93 against 4.69 ms per step, one box).
        if self.fuse_norm == "none" or (h.shape[0] > 2 and not self.fuse_norm.startswith("deferred")):
            return self._step_unfused_norm(st)
        dm = self.fuse_norm in ("deferred", "deferred_mlp")           # MLP / final norm in the deferred form
        dq = self.fuse_norm == "deferred"                              # q|k|v norm as well
        fuse_mlp = dm or self.fuse_norm == "all"
        for li, L in enumerate(e.layers):
            ck, cv = st["cache"][li]
            ops.rmsnorm_gemv(h, L.ln1, eps, L.wqkv, out=qkv, deferred=dq)         # input_layernorm + q/k/v projections
            ops.attn_decode_rope(qkv, st["cos"], st["sin"], ck, cv, st["lens"], ctx, st["ws"], nq, nkv, d)   # RoPE + KV append + attention
            ops.gemv(ctx, L.wo, out=h, epilogue=EPI_RESIDUAL, residual=h)
            if fuse_mlp:
                ops.rmsnorm_gemv(h, L.ln2, eps, L.wgu, out=act, epilogue=EPI_SWIGLU, deferred=dm)   # post_attention_layernorm + gate/up + SwiGLU
            else:
                ops.rmsnorm(h, L.ln2, eps, out=st["x"])
                ops.gemv(st["x"], L.wgu, out=act, epilogue=EPI_SWIGLU)
            ops.gemv(act, L.wdown, out=h, epilogue=EPI_RESIDUAL, residual=h)
        if fuse_mlp:
            ops.rmsnorm_gemv(h, e.norm, eps, self.lm_head, out=st["logits"], deferred=dm)       # final norm + lm_head
        else:
            ops.rmsnorm(h, e.norm, eps, out=st["x"])
            ops.gemv(st["x"], self.lm_head, out=st["logits"])

    def _step_unfused_norm(self, st):
        """More than 2 rows: every workgroup of the fused kernel would re-derive each row's RMS, so the norm gets its own launch."""
        c, e = self.cfg, self.eng
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        h, x, qkv, ctx, act = st["h"], st["x"], st["qkv"], st["ctx"], st["act"]
        for li, L in enumerate(e.layers):
            ck, cv = st["cache"][li]
            ops.rmsnorm(h, L.ln1, eps, out=x)
            ops.gemv(x, L.wqkv, out=qkv)
            ops.rope_kv_append(qkv, st["cos"], st["sin"], ck, cv, st["lens"], nq, nkv, d)
            ops.attn_decode(qkv, ck, cv, st["lens"], ctx, st["ws"], nq, nkv, d)
            ops.gemv(ctx, L.wo, out=h, epilogue=EPI_RESIDUAL, residual=h)
            ops.rmsnorm(h, L.ln2, eps, out=x)
            ops.gemv(x, L.wgu, out=act, epilogue=EPI_SWIGLU)
            ops.gemv(act, L.wdown, out=h, epilogue=EPI_RESIDUAL, residual=h)
        ops.rmsnorm(h, e.norm, eps, out=x)
        ops.gemv(x, self.lm_head, out=st["logits"])

    # ------------------------------------------------------------------ a prompt chunk on top of the cache, all its tokens at once
    def _prompt_rows(self, st, ids: torch.Tensor):
        """The P prompt tokens of every sequence in ONE pass over the weights per 8 rows instead of P decode steps (round 6): the V = B P
        rows go through the decode step's own kernels -- the GEMVs in groups of 8 rows, then every row's k / v appended at position
        lens + i (grit_rope_kv_append_rows), then every row attending to keys 0 .. lens + i of its sequence (grit_attn_decode_rows) --
        so the arithmetic per row is the token-by-token path's, bit for bit (tests: native_generate_prompt_chunk).  Leaves the logits after
        the last prompt token in st["logits"] and advances lens by P.  Built for the default (deferred-norm) step and the fp16 one."""
        c, e, dev = self.cfg, self.eng, self.device
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        f16 = st["f16"]
        B, P = ids.shape
        V = B * P
        op, wide = (F16, F32) if f16 else (BF16, BF16)
        mk = lambda n, dt: torch.empty((V, n), dtype=dt, device=dev)
        h, qkv, ctx, act = mk(c.hidden_size, wide), mk((nq + 2 * nkv) * d, wide), mk(nq * d, op), mk(c.intermediate_size, op)
        h16 = mk(c.hidden_size, F16) if f16 else None
        lens_v = (st["lens"].view(B, 1) + torch.arange(P, dtype=I32, device=dev).view(1, P)).reshape(V).contiguous()
        rows_v = torch.arange(B, dtype=I32, device=dev).repeat_interleave(P).contiguous()
        Lmax = st["cache"][0][0].shape[2]
        ws = ops.attn_decode_workspace(V, nq, nkv, Lmax, dev)
        ops.embed_gather(e.embed, ids.reshape(-1).contiguous(), out=h)
        if f16:
            h16.copy_(h)
        groups = [(a, min(a + 8, V)) for a in range(0, V, 8)]
        x_in = h16 if f16 else h                       # what a norm + GEMV launch reads
        for li, L in enumerate(e.layers):
            wqkv, wo, wgu, wdown = e._f16_weights(L) if f16 else (L.wqkv, L.wo, L.wgu, L.wdown)
            ck, cv = st["cache"][li]
            for a, b in groups:
                ops.rmsnorm_gemv(x_in[a:b], L.ln1, eps, wqkv, out=qkv[a:b], deferred=True)
            ops.rope_kv_append_rows(qkv, st["cos"], st["sin"], ck, cv, lens_v, rows_v, nq, nkv, d)
            ops.attn_decode_rows(qkv, ck, cv, lens_v, rows_v, ctx, ws, nq, nkv, d)
            for a, b in groups:
                if f16:
                    ops.gemv(ctx[a:b], wo, out=h[a:b], epilogue=EPI_RESIDUAL, residual=h[a:b], out16=h16[a:b])
                else:
                    ops.gemv(ctx[a:b], wo, out=h[a:b], epilogue=EPI_RESIDUAL, residual=h[a:b])
            for a, b in groups:
                ops.rmsnorm_gemv(x_in[a:b], L.ln2, eps, wgu, out=act[a:b], epilogue=EPI_SWIGLU, deferred=True)
            for a, b in groups:
                if f16:
                    ops.gemv(act[a:b], wdown, out=h[a:b], epilogue=EPI_RESIDUAL, residual=h[a:b], out16=h16[a:b])
                else:
                    ops.gemv(act[a:b], wdown, out=h[a:b], epilogue=EPI_RESIDUAL, residual=h[a:b])
        last = x_in.view(B, P, -1)[:, P - 1].contiguous()                  # the stream of every sequence's last prompt token
        ops.rmsnorm_gemv(last, e.norm, eps, self._lm_head_f16() if f16 else self.lm_head, out=st["logits"], deferred=True)
        st["lens"] += P

    def _sample(self, st):
        ops.argmax_advance(st["logits"], st["next"], st["lens"], st["history"], st["step"])

    def _state(self, B: int, Lmax: int, f16: bool = False):
        c, dev = self.cfg, self.device
        nq, nkv, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        op = F16 if f16 else BF16                      # operand rows and the K/V cache
        wide = F32 if f16 else BF16                    # the stream, the q|k|v row, the logits
        mk = lambda n, dt: torch.empty((B, n), dtype=dt, device=dev)
        # (the fp16 policies rotate with the unrounded fp32 tables, like the encoder's fp16 forward)
        cos, sin = rope_tables(Lmax, d, c.rope_theta, self.eng.rope_bf16 and not f16, dev)
        moe = {}
        if self.moe:
            moe = dict(act2=torch.empty((2 * B, c.intermediate_size), dtype=op, device=dev), y2=torch.empty((2 * B, c.hidden_size), dtype=wide, device=dev),
                       experts=torch.zeros((B, 2), dtype=I32, device=dev), weights=torch.zeros((B, 2), dtype=F32, device=dev),
                       rows_c=torch.arange(2 * B, dtype=I32, device=dev).view(B, 2).contiguous())
        return dict(**moe, f16=f16, h=mk(c.hidden_size, wide), h16=mk(c.hidden_size, F16) if f16 else None, x=mk(c.hidden_size, op), qkv=mk((nq + 2 * nkv) * d, wide), ctx=mk(nq * d, op),
                    act=mk(c.intermediate_size, op), logits=mk(self.lm_head.shape[0], wide), next=torch.zeros((B,), dtype=I64, device=dev),
                    lens=torch.zeros((B,), dtype=I32, device=dev), step=torch.zeros((1,), dtype=I32, device=dev), cos=cos, sin=sin,
                    ws=ops.attn_decode_workspace(B, nq, nkv, Lmax, dev),
                    cache=[(torch.zeros((B, nkv, Lmax, d), dtype=op, device=dev), torch.zeros((B, nkv, Lmax, d), dtype=op, device=dev))
                           for _ in range(c.num_hidden_layers)])

    # ------------------------------------------------------------------ API
    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, max_new_tokens: int, attention_mask: torch.Tensor | None = None, past_key_values=None,
                 past_lens: torch.Tensor | None = None, eos_token_id: int | None = None, return_logits: bool = False,
                 on_overflow: str | None = None, _force_bf16: bool = False):
        """Greedy continuation.  ``input_ids`` [B,P]: the NEW prompt tokens (right-padded rows need ``attention_mask``; with
        ``past_key_values`` all rows must be full length).  ``past_key_values``: per-layer K/V [B,nkv,S,d] of an already encoded
        prefix, e.g. the bidirectional document pass of ``encode(get_cache=True)``; ``past_lens`` [B] = valid prefix tokens per row
        (default S).  Returns the generated ids [B, max_new_tokens] (int64; positions after ``eos_token_id`` keep that id) and, with
        ``return_logits``, the logits of every generated position [B, max_new_tokens, V] (bf16; fp32 on fp16 operands).

        On fp16 operands (module docstring) a value beyond the fp16 range invalidates the call: ``on_overflow="raise"`` raises
        ``GritHipError``, ``"bf16"`` repeats the whole call in the bf16 arithmetic; default ``self.on_overflow`` ("raise";
        ``GritLM(precision="auto").native_decoder()`` sets "bf16": the ladder's last rung)."""
        dev, c = self.device, self.cfg
        f16 = self._f16() and not _force_bf16
        on_overflow = self.on_overflow if on_overflow is None else on_overflow
        if on_overflow not in ("raise", "bf16"):
            raise ValueError(f"on_overflow={on_overflow!r}: 'raise' or 'bf16'")
        if f16:
            ops.f16_overflow_flag(dev, clear=True)        # a stale flag of an earlier call is not this call's
            self.eng._f16_weights(self.eng.layers[0])     # (refuses weights beyond the fp16 range before anything runs)
        ids = input_ids.to(device=dev, dtype=I64)
        B, P = ids.shape
        if B > 8:
            raise NotImplementedError("native decode: batch <= 8")
        nq, nkv, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        past = _layers_of(past_key_values) if past_key_values is not None else None
        S0 = past[0][0].shape[2] if past is not None else 0
        if 0 < self.eng.window_keys < S0 + P + max_new_tokens:
            # the reference's sdpa path (window_keys = 0) attends to the whole cache; its eager / flash paths window the cache (and the
            # flash path slices it, modeling_mistral_gritlm.py:394-415) -- the decode kernels attend to the whole cache only
            raise NotImplementedError(f"native decode past the sliding window ({self.eng.window_keys} keys) is built for the sdpa semantics "
                                      "only (window_keys = 0: full causal attention over the cache)")
        Lmax = (S0 + P + max_new_tokens + 255) // 256 * 256
        st = self._state(B, Lmax, f16)
        st["history"] = torch.zeros((B, max_new_tokens), dtype=I64, device=dev)
        logits_all = torch.empty((B, max_new_tokens, self.lm_head.shape[0]), dtype=st["logits"].dtype, device=dev) if return_logits else None
        if past is None:
            # prefill: one causal pass over the prompt that also emits the post-RoPE K/V (the encoder engine's forward)
            mask = torch.ones((B, P), dtype=I64, device=dev) if attention_mask is None else attention_mask.to(device=dev, dtype=I64)
            # fp16 operands: the prompt pass runs under the engine's fp16 policy as well (causal fp16 attention, grit_attn_causal_f16_fwd); its
            # fp16 K/V go into the cache as they are and the first token's logits come from the un-normalised stream of the last prompt
            # token through the same deferred norm + lm_head launch the decode steps use.  A decoder pinned to "f16" on a bf16 engine
            # (or the reverse) switches the engine's policy for the pass and restores it.
            was, pol = self.eng.causal, self.eng.precision
            if f16 and pol not in F16_POLICIES:
                self.eng.precision = "f16_operands"
            elif not f16 and pol in F16_POLICIES:
                self.eng.precision = "bf16"
            self.eng.causal = True
            try:
                hidden, kv = self.eng.forward(ids, mask, borrow=True, return_kv=True, final_norm=not f16, kv_dtype=None if f16 else BF16)
            finally:
                self.eng.causal, self.eng.precision = was, pol
            for li, (k, v) in enumerate(kv):
                st["cache"][li][0][:, :, :P].copy_(k); st["cache"][li][1][:, :, :P].copy_(v)
            plen = mask.sum(dim=1).to(I32)
            st["lens"].copy_(plen)
            last = hidden[torch.arange(B, device=dev), (plen - 1).long()].contiguous()              # [B,H] of the last prompt token
            if f16:
                ops.rmsnorm_gemv(last.to(F16), self.eng.norm, c.rms_norm_eps, self._lm_head_f16(), out=st["logits"], deferred=True)
            else:
                ops.gemv(last, self.lm_head, out=st["logits"])                                       # (final-norm output, bf16)
        else:
            if attention_mask is not None and not bool((attention_mask != 0).all()):
                raise NotImplementedError("native decode: padded prompt rows on top of past_key_values")
            for li, (k, v) in enumerate(past):
                st["cache"][li][0][:, :, :S0].copy_(k.to(dev)); st["cache"][li][1][:, :, :S0].copy_(v.to(dev))
            if f16 and any(k.dtype != F16 or v.dtype != F16 for k, v in past):
                self._flag_nonfinite_cache(st, S0)        # (a bf16 / fp32 cache narrowed to fp16: values beyond 65504 became inf)
            st["lens"].copy_(torch.full((B,), S0, dtype=I32, device=dev) if past_lens is None else past_lens.to(device=dev, dtype=I32))
            # the prompt rides on the decode path: all its tokens at once (_prompt_rows: the step's own kernels over B P rows), or, for the
            # exact-norm forms and very long prompts, one token per step (teacher forced); its K/V land behind the cached prefix
            if P > 0 and self.prompt_chunk and (f16 or self.fuse_norm == "deferred") and B * P <= self.prompt_chunk_max_rows:
                self._prompt_rows(st, ids)
            else:
                for t in range(P):
                    st["next"].copy_(ids[:, t])
                    self._step(st)
                    st["lens"] += 1
        # first generated token from the prefill logits, then decode steps (graph replay)
        if return_logits:
            logits_all[:, 0].copy_(st["logits"])
        ops.argmax_advance(st["logits"], st["next"], None, st["history"], st["step"])
        graph = None
        for t in range(1, max_new_tokens):
            # Hugging Face's generate() stops once every row has emitted EOS; here the host only replays a graph, so it looks every 16
            # tokens (one small D2H) instead of after each one: at most 15 tokens of wasted steps, none of them returned
            if eos_token_id is not None and t % 16 == 0 and bool((st["history"][:, :t] == eos_token_id).any(dim=1).all()):
                break
            if graph is not None:
                graph.replay()
            else:
                self._step(st)
                self._sample_step(st, logits_all, t)
                if self.use_graph and logits_all is None and t + 1 < max_new_tokens:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        self._step(st)
                        self._sample(st)
        out = st["history"]
        self.last_precision = "f16" if f16 else ("bf16 (f16 overflow)" if _force_bf16 else "bf16")
        if f16 and (ops.f16_overflow_flag(dev, clear=True) or bool(st.get("cache_bad", False))):
            if on_overflow == "bf16":
                return self.generate(input_ids, max_new_tokens, attention_mask=attention_mask, past_key_values=past_key_values, past_lens=past_lens,
                                     eos_token_id=eos_token_id, return_logits=return_logits, _force_bf16=True)
            raise GritHipError("native decode on fp16 operands: a value exceeded the fp16 range (activation, K/V or a cached K/V that was "
                               "already inf); the tokens of this call are invalid -- decode with precision 'bf16' (MistralDecoder.precision, "
                               "or on_overflow='bf16')")
        if eos_token_id is not None:
            hit = (out == eos_token_id).long().cumsum(dim=1) > 0
            out = torch.where(hit, torch.full_like(out, eos_token_id), out)
        return (out, logits_all) if return_logits else out

    def _flag_nonfinite_cache(self, st, n: int):
        """K/V narrowed to fp16 on their way into the cache: a value beyond the range would surface as nan logits without touching the
        device's overflow flag (the fp32 outputs are not range-checked) -- raise it here."""
        bad = torch.zeros((), dtype=torch.bool, device=self.device)
        for ck, cv in st["cache"]:
            bad |= ~torch.isfinite(ck[:, :, :n]).all()
            bad |= ~torch.isfinite(cv[:, :, :n]).all()
        st["cache_bad"] = bad

    def _sample_step(self, st, logits_all, t):
        if logits_all is not None:
            logits_all[:, t].copy_(st["logits"])
        self._sample(st)
