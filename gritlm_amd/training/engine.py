"""Native forward+backward engine for the contrastive training step (GradCache pass 1 / pass 2).

Binds to the parameters of a Hugging Face ``MistralModel`` WITHOUT copying them: the fused QKV and [gate | up]
weights live in packed storage and the module's ``q_proj/k_proj/v_proj`` and ``gate_proj/up_proj`` parameters are
re-pointed to row-slices of it (views), so ``state_dict()`` keeps the reference names (checkpoints stay
interchangeable), the optimizer updates the packed storage in place, and the kernels see ONE [N,K] weight.
Gradients are written by the wgrad GEMMs straight into packed ``.grad`` storage (bf16, accumulated across GradCache
chunks by the GEMM's residual epilogue), exactly what ``param.grad`` accumulation does in the reference.

Replaces, for the embedding tower: torch autograd through scripts/modeling_mistral_gritlm.py + HF gradient
checkpointing (gritlm/training/run.py:83-84).  Memory is laid out for 288 GB HBM: by default pass 2 keeps every intermediate
of the chunk (no recompute => 3x forward FLOPs per chunk instead of the reference's 4x); ``recompute = True``
(``--gradient_checkpointing``) keeps only the layer inputs and re-runs each layer's forward inside backward().
"""
from __future__ import annotations

import os

import torch

from .. import ops
from .._lib import EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_STORE, EPI_SWIGLU_BWD, EPI_SWIGLU_STACKED, EPI_SWIGLU_STACKED_SAVE, GritHipError
from ..encoder import F16_POLICIES, EncoderConfig, rope_tables, sliding_window_keys

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


def _pad_k(n: int) -> int:
    """Token count -> K extent of the weight-gradient GEMMs: a multiple of 128, i.e. an EVEN number of 64-wide K-tiles, which the
    persistent launch form of the GEMM needs (zero columns contribute exact zeros)."""
    return (n + 127) // 128 * 128


class _LayerParams:
    __slots__ = ("wqkv", "wo", "wgu", "wdown", "ln1", "ln2", "gqkv", "go", "ggu", "gdown", "mods", "wgate")


class SavedForward:
    """Everything pass 2 keeps from the forward of one chunk."""
    __slots__ = ("ids", "mask", "geom", "B", "S", "layers", "h_final", "x_final", "router")


class _Geometry:
    """Row layout of one chunk.  Padded: T = B*S rows, key-padding bitmask.  Packed: only the real tokens of a right-padded
    batch (rows of sequence b at [cu[b], cu[b+1])), per-row RoPE positions -- GEMMs, norms and attention never see padding
    (the reference computes every padded position)."""
    __slots__ = ("packed", "B", "S", "T", "bits", "cu", "pos", "max_len", "keep", "causal", "window")

    @staticmethod
    def padded(mask: torch.Tensor):
        g = _Geometry()
        g.packed, (g.B, g.S), g.causal, g.window = False, mask.shape, False, 0
        g.T = g.B * g.S
        g.bits = ops.mask_pack(mask)
        g.cu = g.pos = g.keep = None
        g.max_len = g.S
        return g

    @staticmethod
    def from_mask(mask: torch.Tensor):
        """Packed geometry when the mask is right-padded with no empty row, else None."""
        B, S = mask.shape
        keep = mask != 0
        lens = keep.sum(dim=1)
        ar = torch.arange(S, device=mask.device)
        # one host sync decides the layout: (right-padded?, no empty rows?, longest row, number of tokens)
        ok, max_len, T = torch.stack([(((ar.unsqueeze(0) < lens.unsqueeze(1)) == keep).all() & (lens > 0).all()).to(torch.int64),
                                      lens.max(), lens.sum()]).tolist()
        if not ok:
            return None
        g = _Geometry()
        g.packed, g.B, g.S, g.T, g.max_len, g.keep, g.bits, g.causal, g.window = True, B, S, int(T), int(max_len), keep, None, False, 0
        g.cu = torch.zeros((B + 1,), dtype=torch.int32, device=mask.device)
        g.cu[1:] = torch.cumsum(lens, dim=0)
        g.pos = ar.to(torch.int32).unsqueeze(0).expand(B, S)[keep].contiguous()
        return g


class MistralTrainEngine:
    def __init__(self, backbone: torch.nn.Module, hf_config, device, lm_head: torch.nn.Module | None = None):
        """``lm_head`` (optional, nn.Linear [V,H] without bias): enables the generative branch (forward_lm / backward_lm)."""
        self.lm_head = lm_head.weight if lm_head is not None else None
        if self.lm_head is not None and self.lm_head.dtype != BF16:
            raise RuntimeError("MistralTrainEngine: lm_head must be bfloat16")
        # causal (generative) attention: keys a query sees under config.sliding_window -- depends on the attention path, as in the reference
        self.window_keys = sliding_window_keys(getattr(hf_config, "sliding_window", None), getattr(hf_config, "_attn_implementation", None))
        self.cfg = hf_config if isinstance(hf_config, EncoderConfig) else EncoderConfig.from_hf(hf_config)
        self.cfg.check_supported()
        self.device = torch.device(device)
        self.backbone = backbone
        self.rope_bf16 = True
        self._rope = {}
        self._tbuf = {}
        self._wT = {}
        self.cache_transposed_weights = False
        self.pair_wgrads = os.environ.get("GRIT_NO_WGRAD_PAIR") != "1"     # down_proj + q|k|v_proj weight gradients in one launch
        self._deferred_wgrad = None
        self.recompute = False          # gradient checkpointing (per-layer recompute in backward)
        # Precision policy of the NO-GRAD forward (GradCache pass 1: the pass that defines the representations and the loss; round 6):
        # "bf16" = the reference's bf16 arithmetic (the arithmetic of pass 2, whose saved activations the bf16 backward kernels read), or
        # "f16_operands" / "f16_stream" = the encoder's fp16-operand policies (gritlm_amd/encoder.py) on fp16 copies of the packed weights,
        # refreshed when the optimizer has changed them: the loss then matches the reference's fp32 loss to the north-star's 1e-3 at depth 32.
        self.nograd_precision = "bf16"
        self._w16 = {}                  # layer index -> (owner versions, fp16 (wqkv, wo, wgu, wdown))
        self._embed16 = None
        self._ws16 = {}
        self._router_log = None         # Mixtral: list collecting (logits fp32 [T,E], experts [T,2]) per layer while it is a list
        self._aux_dlogits = None        # Mixtral: d aux_loss / d router logits per layer during backward_lm
        c = self.cfg
        if any(p.dtype != BF16 for p in backbone.parameters()):
            raise RuntimeError("MistralTrainEngine: parameters must be bfloat16 (load the model with torch_dtype=bfloat16)")
        self.embed = backbone.embed_tokens.weight
        self.norm = backbone.norm.weight
        self.layers: list[_LayerParams] = []
        for layer in backbone.layers:
            L = _LayerParams()
            if not hasattr(layer, "mlp"):       # e.g. the reference file's own Mixtral class: block_sparse_moe.experts[i].w1/w2/w3
                raise RuntimeError(f"{type(self).__name__}: decoder layers of {type(backbone).__name__} have no `.mlp` (found: "
                                   f"{[n for n, _ in layer.named_children()]}); the training engine binds the layout of the installed "
                                   "transformers (Mistral: mlp.gate_proj/up_proj/down_proj; Mixtral >= 5: mlp.gate + fused "
                                   "mlp.experts.gate_up_proj/down_proj).  Load the checkpoint with AutoModel of the installed transformers.")
            at, mlp = layer.self_attn, layer.mlp
            L.wqkv = self._pack([at.q_proj.weight, at.k_proj.weight, at.v_proj.weight])
            L.wo = at.o_proj.weight
            L.ln1, L.ln2 = layer.input_layernorm.weight, layer.post_attention_layernorm.weight
            L.mods = (at, mlp, layer)
            L.gqkv = L.ggu = L.go = L.gdown = None
            self._bind_mlp(L, mlp)
            self.layers.append(L)
        self._g_embed = None
        self._g_norm = None
        self._f32_norm_grads = None

    # ------------------------------------------------------------------ MLP hooks (dense SwiGLU here; MixtralTrainEngine overrides)
    def _bind_mlp(self, L, mlp):
        L.wgu = self._pack([mlp.gate_proj.weight, mlp.up_proj.weight])
        L.wdown = mlp.down_proj.weight
        L.wgate = None

    def _prepare_mlp_grads(self, L):
        _, mlp, _ = L.mods
        L.ggu = self._packed_grad([mlp.gate_proj.weight, mlp.up_proj.weight], L.ggu)
        L.gdown = self._packed_grad([mlp.down_proj.weight], L.gdown)

    def _mlp_owners(self, li: int, name: str):
        _, mlp, _ = self.layers[li].mods
        return {"gu": (mlp.gate_proj.weight, mlp.up_proj.weight), "down": (mlp.down_proj.weight,)}[name]

    def _layer_grads(self, L) -> list[torch.Tensor]:
        return [L.gqkv, L.go, L.ggu, L.gdown]

    def _mlp_fwd(self, L, h_mid, x2, buf, need_bwd: bool, h_out):
        """h_out = h_mid + down(silu(gate(x2)) * up(x2)); returns what _mlp_bwd reads."""
        gu, act = buf["gu"], buf["act"]
        if need_bwd:      # one launch: activation + the saved pre-activations [gate | up]
            ops.gemm_nt(x2, L.wgu, out=act, epilogue=EPI_SWIGLU_STACKED_SAVE, residual=gu)
        else:
            ops.gemm_nt(x2, L.wgu, out=act, epilogue=EPI_SWIGLU_STACKED)
        ops.gemm_nt(act, L.wdown.data, out=h_out, epilogue=EPI_RESIDUAL, residual=h_mid)
        return dict(gu=gu, act=act)

    def _mlp_bwd(self, li: int, L, sv, dh):
        """Accumulates the MLP's weight gradients; returns d loss / d x2 ([T,H], the input of the MLP = post-attention norm output)."""
        # d_act = dh @ Wdown with the SwiGLU backward in the epilogue: [T, 2I] = [d_gate | d_up] straight from the accumulators
        dgu = ops.gemm_nt(dh, self._wt(li, "down", L.wdown), epilogue=EPI_SWIGLU_BWD, residual=sv["gu"])
        if self.pair_wgrads:      # transposed now (dh is consumed by the next norm backward), multiplied together with the q|k|v wgrad
            self._deferred_wgrad = (self._transposed_act(dh, "dh_down"), self._transposed_act(sv["act"], "act"), L.gdown)
        else:
            self._wgrad(dh, sv["act"], L.gdown, ("dh", "act"))
        dx2 = ops.gemm_nt(dgu, self._wt(li, "gu", L.wgu))                           # [T,H]
        self._wgrad(dgu, sv["x2"], L.ggu, ("dgu", "x"))
        return dx2

    # ------------------------------------------------------------------ parameter packing
    def _pack(self, params):
        rows = sum(p.shape[0] for p in params)
        packed = torch.empty((rows, params[0].shape[1]), dtype=BF16, device=self.device)
        r = 0
        for p in params:
            n = p.shape[0]
            packed[r:r + n].copy_(p.data)
            p.data = packed[r:r + n]          # the module parameter is now a view of the packed storage
            r += n
        return packed

    def _packed_grad(self, params, current):
        """(Re)attach ``.grad`` views of one packed gradient buffer; zero it when the optimizer cleared the grads."""
        rows = sum(p.shape[0] for p in params)
        fresh = current is None or any(p.grad is None for p in params)
        if current is None:
            current = torch.zeros((rows, params[0].shape[1]), dtype=BF16, device=self.device)
        elif fresh:
            current.zero_()
        if fresh:
            r = 0
            for p in params:
                n = p.shape[0]
                p.grad = current[r:r + n]
                r += n
        return current

    def prepare_grads(self):
        c = self.cfg
        for L in self.layers:
            at, mlp, layer = L.mods
            L.gqkv = self._packed_grad([at.q_proj.weight, at.k_proj.weight, at.v_proj.weight], L.gqkv)
            L.go = self._packed_grad([at.o_proj.weight], L.go)
            self._prepare_mlp_grads(L)
        # 1-D parameters and the embedding: plain .grad tensors
        for p in [self.embed, self.norm] + [x for L in self.layers for x in (L.ln1, L.ln2)]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self._f32_norm_grads is None:
            self._f32_norm_grads = torch.zeros((2 * len(self.layers) + 1, c.hidden_size), dtype=F32, device=self.device)

    def weights_updated(self):
        """Call after optimizer.step(): the transposed-weight cache and the fp16 weight copies are stale (both are also keyed on the
        Parameters' version counters, so an in-place update invalidates them even if this is never called)."""
        self._wT.clear()
        self._w16.clear()
        self._embed16 = None

    def set_nograd_precision(self, precision: str):
        if precision not in ("bf16",) + F16_POLICIES:
            raise ValueError(f"nograd_precision={precision!r}: one of {('bf16',) + F16_POLICIES}")
        if precision == "f16_stream" and self.cfg.num_local_experts:
            raise GritHipError("training engine: the sparse-MoE block routes on the fp32 residual stream: use 'f16_operands'")
        self.nograd_precision = precision
        return self

    def check_f16_overflow(self, clear: bool = True) -> None:
        """Raise if a kernel of the fp16 no-grad forward produced a value beyond the fp16 range since the last check (synchronises)."""
        if self.nograd_precision in F16_POLICIES and ops.f16_overflow_flag(self.device, clear):
            raise GritHipError(f"nograd_precision='{self.nograd_precision}': an activation exceeded the fp16 range in the no-grad forward; "
                               "the representations (and the loss) of this step are invalid -- use 'f16_operands' (fp32 residual stream) or 'bf16'")

    def _all_owners(self, li: int):
        at, _, _ = self.layers[li].mods
        return (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, at.o_proj.weight) + tuple(self._mlp_owners(li, "gu")) \
            + tuple(self._mlp_owners(li, "down"))

    def _f16_weights(self, li: int):
        """fp16 copies of layer li's packed GEMM weights (wqkv, wo, [gate; up] stacked, down) -- exact for 6.1e-5 <= |w| < 65520, refused
        beyond -- converted when first needed after an optimizer update (14.5 GB for the 7B shape: ~10 ms of a 50 s step)."""
        ver = tuple(p._version for p in self._all_owners(li))
        hit = self._w16.get(li)
        if hit is not None and hit[0] == ver:
            return hit[1]
        L = self.layers[li]
        out = []
        for w in (L.wqkv, L.wo, L.wgu, L.wdown):
            w16 = w.data.to(F16)
            for part in (w16 if w16.dim() == 3 else (w16,)):            # (slice by slice: an expert stack is 12 GB)
                if bool(torch.isinf(part).any()):
                    raise GritHipError(f"nograd_precision='{self.nograd_precision}': weights of layer {li} exceed the fp16 range")
            out.append(w16)
        self._w16[li] = (ver, tuple(out))
        return self._w16[li][1]

    def _mlp_fwd_f16(self, li: int, L, x2, h, ws, w16):
        """h += down(silu(gate(x2)) * up(x2)) on fp16 operands (x2 fp16; h the fp32 / fp16 residual stream, updated in place)."""
        _, _, wgu, wdown = w16
        ops.gemm_nt(x2, wgu, out=ws["act"], epilogue=EPI_SWIGLU_STACKED)
        ops.gemm_nt(ws["act"], wdown, out=h, epilogue=EPI_RESIDUAL_F32 if h.dtype == F32 else EPI_RESIDUAL, residual=h)

    def _ws16_buffers(self, T: int, stream_dtype):
        c = self.cfg
        cap = self._ws16.get("cap", 0)
        if cap < T or self._ws16.get("sdt") != stream_dtype:
            self._ws16.clear()
            nq, nkv, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
            mk = lambda n, dt=F16, rows=T: torch.empty((rows, n), dtype=dt, device=self.device)
            r = 2 if c.num_local_experts else 1
            self._ws16.update(cap=T, sdt=stream_dtype, h=mk(c.hidden_size, stream_dtype), x=mk(c.hidden_size), qkv=mk((nq + 2 * nkv) * d),
                              ctx=mk(nq * d), act=mk(c.intermediate_size, F16, r * T))
            if c.num_local_experts:
                self._ws16.update(y=mk(c.hidden_size, F16, 2 * T))
        return {k: (v[:2 * T] if k == "y" or (k == "act" and c.num_local_experts) else v[:T]) for k, v in self._ws16.items() if k not in ("cap", "sdt")}

    def _forward_nograd_f16(self, ids, geom, B, S):
        """The no-grad forward under an fp16-operand policy: the encoder engine's data flow (gritlm_amd/encoder.py) on this engine's packed
        parameters.  Returns last_hidden_state rows [T,H] bf16 (the pooling kernels' input; one rounding of the final RMSNorm)."""
        c, pol = self.cfg, self.nograd_precision
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        T = geom.T
        t = self._rope.get((S, "f32"))
        if t is None:
            t = self._rope[(S, "f32")] = rope_tables(S, d, c.rope_theta, False, self.device)      # unrounded tables: the rotation runs in fp32
        cos, sin = t
        ws = self._ws16_buffers(T, F32 if pol == "f16_operands" else F16)
        h, x, qkv, ctx = ws["h"], ws["x"], ws["qkv"], ws["ctx"]
        if pol == "f16_stream":
            key = (self.embed.data_ptr(), self.embed._version)
            if self._embed16 is None or self._embed16[0] != key:
                self._embed16 = (key, self.embed.data.to(F16))
            ops.embed_gather(self._embed16[1], ids, out=h)
        else:
            ops.embed_gather(self.embed.data, ids, out=h)
        epi = EPI_RESIDUAL_F32 if h.dtype == F32 else EPI_RESIDUAL
        for li, L in enumerate(self.layers):
            w16 = self._f16_weights(li)
            ops.rmsnorm(h, L.ln1.data, eps, out=x)
            if geom.packed:
                ops.gemm_nt_rope(x, w16[0], cos, sin, (nq + nkv) * d, positions=geom.pos, out=qkv)
                ops.attn_bidir_varlen(qkv, geom.cu, geom.max_len, nq, nkv, d, out=ctx, causal=geom.causal, window=geom.window)
            else:
                ops.gemm_nt_rope(x, w16[0], cos, sin, (nq + nkv) * d, S=S, out=qkv)
                ops.attn_bidir(qkv, geom.bits, B, S, nq, nkv, d, out=ctx, causal=geom.causal, window=geom.window)
            ops.gemm_nt(ctx, w16[1], out=h, epilogue=epi, residual=h)
            ops.rmsnorm(h, L.ln2.data, eps, out=x)
            self._mlp_fwd_f16(li, L, x, h, ws, w16)
        return ops.rmsnorm(h, self.norm.data, eps, out=torch.empty((T, c.hidden_size), dtype=BF16, device=self.device))

    # ------------------------------------------------------------------ helpers
    def _rope_tables(self, S):
        t = self._rope.get(S)
        if t is None:
            t = rope_tables(S, self.cfg.head_dim, self.cfg.rope_theta, self.rope_bf16, self.device)
            self._rope[S] = t
        return t

    def _transposed_act(self, x: torch.Tensor, tag: str) -> torch.Tensor:
        """x [T,N] -> x^T in a zero-padded [N, T rounded up to 128] buffer (K operand of the wgrad GEMM)."""
        T, N = x.shape
        Tp = _pad_k(T)
        key = (tag, N)
        buf = self._tbuf.get(key)
        if buf is None or buf.shape[1] < Tp:               # grow-only: packed chunks have a different T every time
            buf = torch.zeros((N, Tp), dtype=BF16, device=self.device)
            self._tbuf[key] = buf
        view = buf if buf.shape[1] == Tp else buf[:, :Tp]
        if Tp != T:
            view[:, T:].zero_()                            # K padding of the wgrad GEMM
        ops.transpose(x, out=view)
        return view

    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, g: torch.Tensor, tags=("dy", "x")):
        """g[N_out, K_in] += dy[T, N_out]^T @ x[T, K_in]  (bf16 accumulation in the packed .grad storage through the residual epilogue):
        two zero-padded transposed copies + the NT GEMM.  (A TN form of the kernel that reads dy and x as they lie -- fragments through
        ds_read_b64_tr_b16 -- was built and measured in round 2: bit-identical, 25-35 % slower than NT + both transposes, because the
        transposing reads double the LDS instructions of the load segments; profiles/r02_gemm_tn_wgrad.log.)"""
        ops.gemm_nt(self._transposed_act(dy, tags[0]), self._transposed_act(x, tags[1]), out=g, epilogue=EPI_RESIDUAL, residual=g)

    def _wt(self, li: int, name: str, w: torch.Tensor) -> torch.Tensor:
        """W^T for the dgrad GEMM.  With ``cache_transposed_weights`` it is reused by every GradCache chunk of a step; the cache is
        keyed on the owning Parameters' version counters, so an in-place optimizer update invalidates it even
        if ``weights_updated()`` is never called."""
        key = (li, name)
        at, _, _ = self.layers[li].mods
        if name == "qkv":
            owners = (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight)
        elif name == "o":
            owners = (at.o_proj.weight,)
        else:
            owners = self._mlp_owners(li, name)
        ver = tuple(p._version for p in owners)       # the Parameters' counters (the packed base tensor's own counter does not see them)
        hit = self._wT.get(key) if self.cache_transposed_weights else None
        if hit is not None and hit[0] == ver:
            return hit[1]
        if w.dim() == 3:                              # stacked expert weights [E,N,K] -> [E,K,N]
            t = torch.stack([ops.transpose(w.data[e]) for e in range(w.shape[0])])
        else:
            t = ops.transpose(w.data)
        if self.cache_transposed_weights:
            self._wT[key] = (ver, t)
        return t

    # ------------------------------------------------------------------ one decoder layer
    def _layer_buffers(self, T: int, with_gu: bool):
        c = self.cfg
        mk = lambda n: torch.empty((T, n), dtype=BF16, device=self.device)
        nq, nkv, d = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        return dict(x1=mk(c.hidden_size), qkv=mk((nq + 2 * nkv) * d), ctx=mk(nq * d), x2=mk(c.hidden_size),
                    gu=mk(2 * c.intermediate_size) if with_gu else None, act=mk(c.intermediate_size),
                    h_mid=mk(c.hidden_size) if with_gu else None)

    def _layer_fwd(self, L, h, geom, B, S, cos, sin, buf, need_bwd: bool, h_out=None):
        """MistralDecoderLayer.forward (scripts/modeling_mistral_gritlm.py:738-790) on the kernels.  ``need_bwd``: also produce what
        backward() reads (log-sum-exp rows, the pre-activation gate|up, a separate h_mid); otherwise SwiGLU is fused into the gate|up
        GEMM's epilogue (stacked-weight form) and the residual stream is updated in place."""
        c = self.cfg
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        T, dev = geom.T, self.device
        x1, qkv, ctx, x2 = buf["x1"], buf["qkv"], buf["ctx"], buf["x2"]
        ops.rmsnorm(h, L.ln1.data, eps, out=x1)
        if geom.packed:
            lse = torch.empty((T, nq), dtype=F32, device=dev) if need_bwd else None
            ops.gemm_nt_rope(x1, L.wqkv, cos, sin, (nq + nkv) * d, positions=geom.pos, out=qkv)      # q/k/v projections + RoPE epilogue
            ops.attn_bidir_varlen(qkv, geom.cu, geom.max_len, nq, nkv, d, out=ctx, lse=lse, causal=geom.causal, window=geom.window)
        else:
            lse = torch.empty((B, nq, S), dtype=F32, device=dev) if need_bwd else None
            ops.gemm_nt_rope(x1, L.wqkv, cos, sin, (nq + nkv) * d, S=S, out=qkv)
            ops.attn_bidir(qkv, geom.bits, B, S, nq, nkv, d, out=ctx, lse=lse, causal=geom.causal, window=geom.window)
        if need_bwd:
            h_mid = buf["h_mid"]
        else:                            # h must survive when it is a saved layer input (recompute policy) -> h_out doubles as h_mid
            h_mid = h_out if h_out is not None else h
        ops.gemm_nt(ctx, L.wo.data, out=h_mid, epilogue=EPI_RESIDUAL, residual=h)
        ops.rmsnorm(h_mid, L.ln2.data, eps, out=x2)
        if h_out is None:
            h_out = h_mid
        sv = dict(h_in=h, x1=x1, qkv=qkv, ctx=ctx, lse=lse, h_mid=h_mid, x2=x2, h_out=h_out)
        sv.update(self._mlp_fwd(L, h_mid, x2, buf, need_bwd, h_out))
        return sv

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, save: bool, packed: bool = False, causal: bool = False):
        """Returns (last_hidden_state, SavedForward | None).  Padded layout: last_hidden_state is [B,S,H] bf16.
        ``causal``: causal attention (the generative branch, 'cc' in the reference's attn string) instead of bidirectional.
        ``packed=True`` (and a right-padded mask without empty rows): [T_real,H] rows of the real tokens only; the geometry is in
        ``SavedForward.geom`` (returned even with ``save=False`` so the caller can pool)."""
        c = self.cfg
        B, S = input_ids.shape
        dev = self.device
        mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        geom = _Geometry.from_mask(mask) if packed else None
        if geom is None:
            geom = _Geometry.padded(mask)
        geom.causal = bool(causal)
        geom.window = int(self.window_keys) if causal and 0 < self.window_keys < S else 0
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        ids = ids[geom.keep].contiguous() if geom.packed else ids.view(-1)
        T = geom.T
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        H, I = c.hidden_size, c.intermediate_size
        saved = SavedForward()
        saved.ids, saved.mask, saved.geom, saved.B, saved.S, saved.layers = ids, mask, geom, B, S, []
        if not save and self.nograd_precision in F16_POLICIES and self._router_log is None:
            xf = self._forward_nograd_f16(ids, geom, B, S)
            return (xf if geom.packed else xf.view(B, S, H)), saved
        cos, sin = self._rope_tables(S)
        mk = lambda n: torch.empty((T, n), dtype=BF16, device=dev)
        h = ops.embed_gather(self.embed.data, ids, out=mk(H))
        # activation policy: save and not recompute -> every intermediate of every layer stays in HBM (3x forward FLOPs per step);
        # save and recompute -> only each layer's INPUT is kept and backward() re-runs the layer's forward first (the reference's
        # gradient checkpointing, gritlm/training/run.py:83-84: 4x forward FLOPs, ~1/17 of the activation memory);
        # not save -> one scratch set, SwiGLU fused into the gate|up GEMM's epilogue
        keep_all = save and not self.recompute
        scratch = None
        saved.router = self._router_log
        for L in self.layers:
            if keep_all or scratch is None:
                scratch = self._layer_buffers(T, with_gu=keep_all)
            h_out = mk(H) if save else None
            sv = self._layer_fwd(L, h, geom, B, S, cos, sin, scratch, need_bwd=keep_all, h_out=h_out)
            if save:
                saved.layers.append(sv if keep_all else dict(h_in=h))
            h = sv["h_out"]
        xf = ops.rmsnorm(h, self.norm.data, eps, out=mk(H))
        if save:
            saved.h_final, saved.x_final = h, xf
        return (xf if geom.packed else xf.view(B, S, H)), saved

    def forward_pooled(self, input_ids, attention_mask, method: str, normalize: bool, instr_len=None, save: bool = False,
                       packed: bool = True, causal: bool = False):
        """reps [B,H] fp32 = normalise(pool(encoder(ids))) (gritlm/training/model.py:134-165) + what backward_pooled needs.  ``causal``: the
        'cc' embedding attention of the reference's attn string (the stock forward without ``is_causal=False``, :146-148)."""
        hidden, saved = self.forward(input_ids, attention_mask, save=save, packed=packed, causal=causal)
        inv = torch.empty((saved.B,), dtype=F32, device=self.device)
        if saved.geom.packed:
            reps = ops.pool_norm_varlen(hidden, saved.geom.cu, method, normalize, instr_len, inv_norm=inv)
        else:
            reps = ops.pool_norm(hidden, saved.mask, method, normalize, instr_len, inv_norm=inv)
        return reps, ((saved, inv, reps, method, normalize, instr_len) if save else None)

    def backward_pooled(self, state, d_reps: torch.Tensor, on_layer_done=None):
        saved, inv, reps, method, normalize, instr_len = state
        d_reps = d_reps.to(F32).contiguous()
        if saved.geom.packed:
            dh = ops.pool_norm_varlen_bwd(reps, d_reps, inv, saved.geom.cu, saved.geom.T, method, normalize, instr_len)
        else:
            dh = ops.pool_norm_bwd(reps, d_reps, inv, saved.mask, method, normalize, saved.S, instr_len)
        self.backward(saved, dh, on_layer_done=on_layer_done)

    # ------------------------------------------------------------------ generative branch
    def forward_lm(self, input_ids, attention_mask, labels, loss_gen_type: str = "mixed", loss_gen_factor: float = 1.0, save: bool = True,
                   packed: bool = True, router_aux_coef: float = 0.0):
        """NextTokenLoss(labels, lm_head(model(ids, causal))) -- gritlm/training/model.py:66-107,185-194 -- as a device scalar,
        plus the state backward_lm needs.  'mixed': mean over the non-ignored shifted tokens of this call; 'token': sum / batch."""
        if self.lm_head is None:
            raise RuntimeError("forward_lm: the engine was built without an lm_head")
        if loss_gen_type not in ("mixed", "token"):
            raise ValueError(f"Invalid loss_gen_type: {loss_gen_type}")
        B, S = input_ids.shape
        self._router_log = [] if router_aux_coef else None
        try:
            hidden, saved = self.forward(input_ids, attention_mask, save=save, packed=packed, causal=True)
        finally:
            self._router_log = None
        geom = saved.geom
        x = hidden if geom.packed else hidden.view(B * S, -1)
        # "tokens < n predict n" (:94-96): row (b, s) is scored against labels[b, s+1]; the last position of a row has no target
        lab = labels.to(device=self.device, dtype=torch.int64)
        shifted = torch.full_like(lab, -100)
        shifted[:, :-1] = lab[:, 1:]
        shifted = shifted[geom.keep].contiguous() if geom.packed else shifted.reshape(-1).contiguous()
        logits = ops.gemm_nt(x, self.lm_head.data)                                  # [T, V] bf16
        lse, loss_row = ops.ce_fwd(logits, shifted)
        n_valid = (shifted >= 0).sum().to(F32)
        if loss_gen_type == "mixed":
            inv = 1.0 / n_valid                                                     # CrossEntropyLoss(reduction="mean")
        else:
            inv = torch.full((), 1.0 / B, dtype=F32, device=self.device)            # reduction="sum" / labels.size(0)
        loss = loss_row.sum() * inv * loss_gen_factor
        aux_dl = None
        if router_aux_coef:
            aux, aux_dl = self._router_aux_loss(saved, want_grad=save)
            loss = loss + router_aux_coef * aux
        state = (saved, x, logits, shifted, lse, inv.reshape(1).contiguous(), float(loss_gen_factor), aux_dl, float(router_aux_coef)) if save else None
        return loss, state

    def _router_aux_loss(self, saved, want_grad: bool):
        raise NotImplementedError("router auxiliary loss: only on MixtralTrainEngine")

    def backward_lm(self, state, d_loss: torch.Tensor | float = 1.0, on_layer_done=None):
        """Accumulate the parameter gradients (backbone + lm_head) of d_loss * loss."""
        saved, x, logits, shifted, lse, inv, factor, aux_dl, aux_coef = state
        dev_scale = inv * d_loss if torch.is_tensor(d_loss) else inv * float(d_loss)
        if aux_dl is not None:          # d (coef * aux) / d router logits of every layer, scaled by the incoming d_loss
            sc = (d_loss.to(F32) if torch.is_tensor(d_loss) else float(d_loss)) * aux_coef
            self._aux_dlogits = [g * sc for g in aux_dl]
        dlogits = ops.ce_bwd_(logits, shifted, lse, factor, dev_scale.to(F32).reshape(1).contiguous())
        self.prepare_grads()
        if self.lm_head.grad is None:
            self.lm_head.grad = torch.zeros_like(self.lm_head)
        dx = ops.gemm_nt(dlogits, ops.transpose(self.lm_head.data))                                     # [T,H] = dlogits @ W_lm
        g = self.lm_head.grad
        self._wgrad(dlogits, x, g, ("dlogits", "x"))
        try:
            self.backward(saved, dx, on_layer_done=on_layer_done)
        finally:
            self._aux_dlogits = None

    # ------------------------------------------------------------------ backward
    def backward(self, saved: SavedForward, d_last_hidden: torch.Tensor, on_layer_done=None):
        """Accumulate parameter gradients for d loss / d last_hidden_state = ``d_last_hidden`` ([B,S,H], or [T_real,H] for a
        packed chunk) bf16.

        ``on_layer_done(list_of_grad_buffers)`` is called as soon as a layer's four weight-gradient buffers are final for this
        call (data-parallel training passes a callback that starts their all-reduce, overlapping it with the remaining layers)."""
        c = self.cfg
        self.prepare_grads()
        B, S, geom = saved.B, saved.S, saved.geom
        T = geom.T
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        H = c.hidden_size
        cos, sin = self._rope_tables(S)
        ng = self._f32_norm_grads
        nL = len(self.layers)
        dy = d_last_hidden.reshape(T, H).contiguous()
        dh = ops.rmsnorm_bwd(dy, saved.h_final, self.norm.data, eps, ng[2 * nL])
        rc_buf = rc_out = None
        for li in range(nL - 1, -1, -1):
            L, sv = self.layers[li], saved.layers[li]
            if "x1" not in sv:             # recompute policy: only the layer input was kept -> re-run the layer's forward now
                if rc_buf is None:
                    rc_buf, rc_out = self._layer_buffers(T, with_gu=True), torch.empty((T, H), dtype=BF16, device=self.device)
                sv = self._layer_fwd(L, sv["h_in"], geom, B, S, cos, sin, rc_buf, need_bwd=True, h_out=rc_out)
            # ---- MLP: h_out = h_mid + mlp(x2)
            dx2 = self._mlp_bwd(li, L, sv, dh)
            dh_mid = ops.rmsnorm_bwd(dx2, sv["h_mid"], L.ln2.data, eps, ng[2 * li + 1], dres=dh)
            # ---- attention: h_mid = h_in + o_proj(attn(rope(qkv(x1))))
            dctx = ops.gemm_nt(dh_mid, self._wt(li, "o", L.wo))                         # [T,nq*d]
            self._wgrad(dh_mid, sv["ctx"], L.go, ("dh", "ctx"))
            if geom.packed:
                dqkv = ops.attn_bidir_varlen_bwd(sv["qkv"], geom.cu, geom.max_len, sv["ctx"], dctx, sv["lse"], nq, nkv, d, causal=geom.causal, window=geom.window)
                ops.rope_qk_pos_(dqkv, cos, sin, geom.pos, nq, nkv, d, inverse=True)
            else:
                dqkv = ops.attn_bidir_bwd(sv["qkv"], geom.bits, sv["ctx"], dctx, sv["lse"], B, S, nq, nkv, d, causal=geom.causal, window=geom.window)
                ops.rope_qk_(dqkv, cos, sin, S, nq, nkv, d, inverse=True)
            dx1 = ops.gemm_nt(dqkv, self._wt(li, "qkv", L.wqkv))                        # [T,H]
            if self._deferred_wgrad is not None:
                # down_proj's and q|k|v_proj's weight gradients in one launch: 896 + 384 tiles of 256 x 256 = 5 whole waves of 256 CUs
                # (alone: 3.5 and 1.5 waves, i.e. one wave of K = T tiles per layer idle on half the chip); same bits as two launches
                aT, xT, g = self._deferred_wgrad
                self._deferred_wgrad = None
                ops.gemm_nt_pair(aT, xT, g, self._transposed_act(dqkv, "dqkv"), self._transposed_act(sv["x1"], "x"), L.gqkv)
            else:
                self._wgrad(dqkv, sv["x1"], L.gqkv, ("dqkv", "x"))
            dh = ops.rmsnorm_bwd(dx1, sv["h_in"], L.ln1.data, eps, ng[2 * li], dres=dh_mid)
            if on_layer_done is not None:
                on_layer_done(self._layer_grads(L))
        # ---- embedding + fold the fp32 side accumulators into the bf16 .grad tensors
        ops.embed_scatter_add(dh, saved.ids, self.embed.grad)          # fixed summation order: the whole step is bit-reproducible
        for li, L in enumerate(self.layers):
            ops.accum_bf16_from_f32(L.ln1.grad, ng[2 * li])
            ops.accum_bf16_from_f32(L.ln2.grad, ng[2 * li + 1])
        ops.accum_bf16_from_f32(self.norm.grad, ng[2 * nL])
        ng.zero_()

    def grad_buffers(self, small_only: bool = False) -> list[torch.Tensor]:
        """Flat list of gradient storages (few, large): the buckets of the data-parallel all-reduce.
        ``small_only``: just the 1-D norm weights and the embedding (the rest was reduced layer by layer during backward)."""
        self.prepare_grads()
        out = []
        for L in self.layers:
            out += ([] if small_only else self._layer_grads(L)) + [L.ln1.grad, L.ln2.grad]
        if self.lm_head is not None and self.lm_head.grad is not None:
            out.append(self.lm_head.grad)
        return out + [self.embed.grad, self.norm.grad]


class MixtralTrainEngine(MistralTrainEngine):
    """The same engine on a Hugging Face ``MixtralModel`` (transformers >= 5 layout: ``mlp.gate.weight [E,H]``,
    ``mlp.experts.gate_up_proj [E,2I,H]`` = [gate rows | up rows] per expert, ``mlp.experts.down_proj [E,H,I]``): attention, norms and
    the residual stream are the base class's; the MLP is the sparse-MoE block of scripts/modeling_mixtral_gritlm.py:815-882
    (softmax -> top-2 -> renormalise routing, per-expert SwiGLU MLP, weighted sum) with its autograd:

      forward   route (kernel) -> grouped gate|up GEMM on the token-sorted rows (SwiGLU + saved pre-activations in the epilogue,
                straight on the fused ``gate_up_proj`` parameter) -> grouped down GEMM -> weighted combine + residual
      backward  combine backward (dy = w * dh[token], dw = <y, dh[token]>) -> grouped dgrad with the SwiGLU backward in the epilogue
                -> grouped dgrad to the sorted inputs -> sum of a token's two rows; per-expert weight gradients with the dense NT
                GEMM on the expert's row segment (ONE host sync per layer for the 8 row counts -- the training GEMMs are
                milliseconds long; the inference path stays sync-free); the router's softmax / top-2 / renormalise backward and
                its two skinny products ([T,E] x [E,H], [E,T] x [T,H]) are HIP kernels too (grit_moe_router_bwd / _wgrad, csrc/moe.hip).

    The router auxiliary loss (load_balancing_loss_func, :80-153) belongs to the generative branch only (the reference adds it inside
    MixtralForCausalLM.forward, which GritLMTrainModel calls with output_router_logits=True for a Mixtral; never for the embedding
    tower): forward_lm(router_aux_coef=...) records the router logits, _router_aux_loss evaluates it, _mlp_bwd feeds its gradient in."""

    def _bind_mlp(self, L, mlp):
        ex = getattr(mlp, "experts", None)
        if ex is None or not hasattr(ex, "gate_up_proj") or not hasattr(ex, "down_proj") or not hasattr(mlp, "gate"):
            raise RuntimeError("MixtralTrainEngine: the sparse-MoE block must expose `gate.weight` and the fused `experts.gate_up_proj` / "
                               "`experts.down_proj` parameters (transformers >= 5); a per-expert w1/w3/w2 module list is not bound")
        L.wgu, L.wdown, L.wgate = ex.gate_up_proj, ex.down_proj, mlp.gate.weight
        c = self.cfg
        E, H, I = c.num_local_experts, c.hidden_size, c.intermediate_size
        if tuple(L.wgu.shape) != (E, 2 * I, H) or tuple(L.wdown.shape) != (E, H, I) or tuple(L.wgate.shape) != (E, H):
            raise RuntimeError(f"MixtralTrainEngine: unexpected expert parameter shapes {tuple(L.wgu.shape)}, {tuple(L.wdown.shape)}, "
                               f"{tuple(L.wgate.shape)} (fused gate_up_proj / down_proj layout of transformers >= 5 expected)")

    def _prepare_mlp_grads(self, L):
        for p in (L.wgu, L.wdown, L.wgate):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        L.ggu, L.gdown = L.wgu.grad, L.wdown.grad

    def _mlp_owners(self, li: int, name: str):
        L = self.layers[li]
        return {"gu": (L.wgu,), "down": (L.wdown,)}[name]

    def _layer_grads(self, L) -> list[torch.Tensor]:
        return [L.gqkv, L.go, L.ggu, L.gdown, L.wgate.grad]

    def _layer_buffers(self, T: int, with_gu: bool):
        c = self.cfg
        buf = super()._layer_buffers(T, with_gu=False)
        mk = lambda n: torch.empty((2 * T, n), dtype=BF16, device=self.device)      # every token visits two experts
        buf["act"], buf["y"] = mk(c.intermediate_size), mk(c.hidden_size)
        buf["gu"] = mk(2 * c.intermediate_size) if with_gu else None
        if with_gu:
            buf["h_mid"] = torch.empty((T, c.hidden_size), dtype=BF16, device=self.device)
        return buf

    def _mlp_fwd_f16(self, li: int, L, x2, h, ws, w16):
        """The sparse-MoE block of the fp16 no-grad forward ("f16_operands": h is the fp32 residual stream): routing in fp32 on h itself
        (post-attention RMSNorm folded in), experts on fp16 copies of the fused gate_up_proj / down_proj, combine into the fp32 stream."""
        T = x2.shape[0]
        _, _, wgu, wdown = w16
        experts, weights, counts, row_token, rows = ops.moe_route_f32(h, L.ln2.data, self.cfg.rms_norm_eps, L.wgate.data)
        ops.gemm_nt_grouped(x2, wgu, counts, 2 * T, out=ws["act"], epilogue=EPI_SWIGLU_STACKED, a_rows=row_token)
        ops.gemm_nt_grouped(ws["act"], wdown, counts, 2 * T, out=ws["y"])
        ops.moe_combine(ws["y"], rows, weights, h, out=h)

    def _mlp_fwd(self, L, h_mid, x2, buf, need_bwd: bool, h_out):
        T = x2.shape[0]
        experts, weights, counts, row_token, rows = ops.moe_route(x2, L.wgate.data)
        if self._router_log is not None and len(self._router_log) < len(self.layers):     # (not again for a recomputed layer)
            self._router_log.append((x2.to(F32) @ L.wgate.data.to(F32).t(), experts))
        gu, act, y = buf["gu"], buf["act"], buf["y"]
        if need_bwd:
            ops.gemm_nt_grouped_epi(x2, L.wgu.data, counts, 2 * T, EPI_SWIGLU_STACKED_SAVE, out=act, residual=gu, a_rows=row_token)
        else:
            ops.gemm_nt_grouped_epi(x2, L.wgu.data, counts, 2 * T, EPI_SWIGLU_STACKED, out=act, a_rows=row_token)
        ops.gemm_nt_grouped_epi(act, L.wdown.data, counts, 2 * T, EPI_STORE, out=y)
        ops.moe_combine(y, rows, weights, h_mid, out=h_out)
        return dict(gu=gu, act=act, y=y, experts=experts, weights=weights, counts=counts, row_token=row_token, rows=rows)

    def _mlp_bwd(self, li: int, L, sv, dh):
        T, H = dh.shape
        counts, row_token, rows = sv["counts"], sv["row_token"], sv["rows"]
        dy, dw = ops.moe_combine_bwd(dh, sv["y"], row_token, rows, sv["weights"])                       # [2T,H] bf16, [T,2] fp32
        dgu = ops.gemm_nt_grouped_epi(dy, self._wt(li, "down", L.wdown), counts, 2 * T, EPI_SWIGLU_BWD, residual=sv["gu"])   # [2T,2I]
        dxs = ops.gemm_nt_grouped_epi(dgu, self._wt(li, "gu", L.wgu), counts, 2 * T, EPI_STORE)                            # [2T,H]
        ones = torch.ones((T, 2), dtype=F32, device=self.device)
        dx2 = ops.moe_combine(dxs, rows, ones, None)                                                    # a token's two routed rows
        # ---- per-expert weight gradients on the expert's row segment
        x_sorted = sv["x2"].index_select(0, row_token.to(torch.int64))
        off = 0
        for e, n in enumerate(counts.tolist()):
            if n:
                seg = slice(off, off + n)
                self._wgrad(dy[seg], sv["act"][seg], L.gdown[e], ("dy_e", "act_e"))
                self._wgrad(dgu[seg], x_sorted[seg], L.ggu[e], ("dgu_e", "x_e"))
                off += n
        # ---- router: w = renormalised top-2 of softmax(x2 Wg^T); d w arrives from the combine backward.  Native (round 4): one kernel
        #      recomputes the fp32 logits, differentiates softmax / top-2 / renormalise (+ the auxiliary loss's pull on this layer's
        #      logits), and adds dlogits @ Wg to the experts' input gradient; a fixed-order two-level sum gives the gate's weight gradient
        aux = None if self._aux_dlogits is None else self._aux_dlogits[li].to(F32).contiguous()
        return ops.moe_router_bwd(sv["x2"], L.wgate.data, sv["experts"], dw, dx2, L.wgate.grad, aux_dlogits=aux)

    def _router_aux_loss(self, saved, want_grad: bool):
        """load_balancing_loss_func (scripts/modeling_mixtral_gritlm.py:80-153) over the router logits of every layer of one forward:
        E * sum_{slot,e} f[slot,e] * P[e], f = fraction of the (real) tokens whose top-`slot` expert is e, P = mean router probability
        of e -- the reference masks padding positions with the attention mask; here the rows of a packed chunk ARE the real
        tokens, and a padded chunk uses the mask.  Returns (aux, [d aux / d logits per layer] | None); only P is differentiable."""
        E = self.cfg.num_local_experts
        log = saved.router
        if not log or len(log) != len(self.layers):
            raise RuntimeError("router auxiliary loss: the forward did not record the router logits")
        keep = None if saved.geom.packed else (saved.mask.reshape(-1) != 0)
        leaves = [lg.detach().requires_grad_(want_grad) for lg, _ in log]
        with torch.enable_grad():
            logits = torch.cat(leaves, dim=0)
            sel = torch.cat([ex for _, ex in log], dim=0).to(torch.int64)                       # [L*T, 2]
            probs = torch.softmax(logits, dim=-1)
            onehot = torch.nn.functional.one_hot(sel, E).to(F32)                                 # [L*T, 2, E]
            if keep is None:
                f, P = onehot.mean(dim=0), probs.mean(dim=0)
            else:
                m = keep.repeat(len(log)).to(F32)
                f = (onehot * m[:, None, None]).sum(dim=0) / m.sum()
                P = (probs * m[:, None]).sum(dim=0) / m.sum()
            aux = (f * P.unsqueeze(0)).sum() * E
            if want_grad:
                aux.backward()
        return aux.detach(), ([lf.grad for lf in leaves] if want_grad else None)


class SyntheticBackbone(torch.nn.Module):
    """Parameter container with the attribute layout of HF ``MistralModel`` (embed_tokens, layers[i].self_attn.{q,k,v,o}_proj,
    layers[i].mlp.{gate,up,down}_proj, layers[i].{input,post_attention}_layernorm, norm), random-initialised directly on the
    device in bf16.  For benchmarks / tests that need the 7B shape without materialising a Hugging Face model on the host."""

    def __init__(self, cfg: EncoderConfig, device, seed: int = 0, std: float = 0.02):
        super().__init__()
        gen = torch.Generator(device=device).manual_seed(seed)
        H, I, d = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        nq, nkv = cfg.num_attention_heads, cfg.num_key_value_heads

        def lin(o, i):
            m = torch.nn.Linear(i, o, bias=False, device=device, dtype=BF16)
            m.weight.data.copy_((torch.randn((o, i), generator=gen, device=device, dtype=F32) * std).to(BF16))
            return m

        class _Norm(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.weight = torch.nn.Parameter((1.0 + 0.1 * torch.randn((H,), generator=gen, device=device, dtype=F32)).to(BF16))

        self.embed_tokens = torch.nn.Embedding(cfg.vocab_size, H, device=device, dtype=BF16)
        self.embed_tokens.weight.data.copy_((torch.randn((cfg.vocab_size, H), generator=gen, device=device, dtype=F32) * std).to(BF16))
        self.layers = torch.nn.ModuleList()
        for _ in range(cfg.num_hidden_layers):
            layer = torch.nn.Module()
            layer.self_attn = torch.nn.Module()
            layer.self_attn.q_proj, layer.self_attn.k_proj = lin(nq * d, H), lin(nkv * d, H)
            layer.self_attn.v_proj, layer.self_attn.o_proj = lin(nkv * d, H), lin(H, nq * d)
            layer.mlp = torch.nn.Module()
            layer.mlp.gate_proj, layer.mlp.up_proj, layer.mlp.down_proj = lin(I, H), lin(I, H), lin(H, I)
            layer.input_layernorm, layer.post_attention_layernorm = _Norm(), _Norm()
            self.layers.append(layer)
        self.norm = _Norm()
        self.config = cfg
