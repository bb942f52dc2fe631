"""Native bidirectional Mistral / Mixtral encoder: the host-side driver of the HIP kernels.

Replaces ``MistralModel.forward(..., is_causal=False)`` of scripts/modeling_mistral_gritlm.py:936-1096 for
the embedding path.  Per layer (reference: 3+1+3 nn.Linear GEMMs, ~20 elementwise kernels, repeat_kv,
a [B,1,S,S] mask) it launches 7 kernels:

    rmsnorm -> fused QKV GEMM with RoPE in its epilogue -> flash attention (GQA, key bitmask)
            -> o_proj GEMM + residual epilogue -> rmsnorm -> gate|up GEMM + SwiGLU epilogue
            -> down GEMM + residual epilogue

Mixtral (scripts/modeling_mixtral_gritlm.py: same attention, sparse-MoE MLP :815-882) swaps the last two launches for
router -> index -> grouped w1|w3 GEMM + SwiGLU (token gather folded into the A loads) -> grouped w2 GEMM -> weighted combine
+ residual; the per-expert Python loop with its ``.tolist()`` host syncs is gone (expert row counts stay on the device).

Weights are repacked once (QKV concatenated, gate/up rows interleaved in blocks of 16 so that the SwiGLU
epilogue is register-local); activations live in a per-token-count workspace sized for 288 GB HBM.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import ops
from ._lib import EPI_RESIDUAL, EPI_RESIDUAL_F32, EPI_SWIGLU, GritHipError

BF16 = torch.bfloat16
F16 = torch.float16

# Precision policies of the forward pass (DESIGN "precision"):
#   "bf16"           the reference's bf16 arithmetic, op by op (hidden_states rounded to bf16 after every residual add, the Linear output
#                    before it, q|k|v before the rotation, gate / up / silu separately): what its bf16 run computes.  Default.
#   "fp32_residual"  bf16 MFMA operands, residual stream in fp32 (GRIT_EPI_RESIDUAL_F32 / grit_rmsnorm_fwd_f32in).
#   "f16_operands"   fp32 residual stream AND every MFMA operand (RMSNorm output, q|k|v, P, attention output, SwiGLU activation, weights)
#                    in IEEE fp16, each rounded once from fp32: same MFMA rate, 3 more mantissa bits -- the policy that meets the
#                    north-star tolerance (1 - cos < 1e-4 against the reference's fp32 run) at depth 32.  bf16 checkpoints convert to fp16
#                    exactly for 6.1e-5 <= |w| < 65520; an activation beyond the fp16 range raises (check_f16_overflow), it never saturates
#                    silently.  Bidirectional attention; dense (Mistral) and, since round 6, sparse-MoE (Mixtral) models -- there the routing
#                    decision is taken in fp32 on the residual stream itself (grit_moe_router_top2_f32), the expert GEMMs run on fp16 copies
#                    of w1|w3 / w2 (grit_gemm_f16_nt_grouped), the weighted combine adds into the fp32 stream (grit_moe_combine_f32).
#   "f16_stream"     "f16_operands" with the residual stream itself in fp16 (16-bit residual epilogues and norms, as in the bf16 policy):
#                    1 - cos 6e-6 at depth 32 (emulated; the stream's 11-bit mantissa adds 2e-6 to the fp32 stream's 4e-6) at 0.975 of the
#                    default's docs/s instead of 0.955.  The stream of a checkpoint with activations beyond 65504 does not fit: that raises
#                    (the same flag); use "f16_operands" there.
PRECISIONS = ("bf16", "fp32_residual", "f16_operands", "f16_stream")
F16_POLICIES = ("f16_operands", "f16_stream")
# precision="auto" (GritLM): the ladder walked from the fastest policy that meets the north-star tolerance down to the ones with more
# range -- a rung is left for good the first time one of its kernels flags a value beyond the fp16 range (gritlm.py::GritLM.encode)
AUTO_LADDER = ("f16_stream", "f16_operands", "fp32_residual", "bf16")


@dataclass
class EncoderConfig:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    head_dim: int | None = None
    num_local_experts: int = 0          # > 0: Mixtral sparse-MoE MLP
    num_experts_per_tok: int = 2

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads

    @classmethod
    def from_hf(cls, hf):
        theta = getattr(hf, "rope_theta", None)
        if theta is None:  # transformers >= 5 moved it
            rp = getattr(hf, "rope_parameters", None) or {}
            theta = rp.get("rope_theta", 10000.0)
        return cls(hf.hidden_size, hf.intermediate_size, hf.num_hidden_layers, hf.num_attention_heads,
                   hf.num_key_value_heads, hf.vocab_size, hf.rms_norm_eps, float(theta), getattr(hf, "head_dim", None),
                   int(getattr(hf, "num_local_experts", 0) or 0), int(getattr(hf, "num_experts_per_tok", 2) or 2))

    @classmethod
    def from_dict(cls, d):
        keys = cls.__dataclass_fields__.keys()
        return cls(**{k: v for k, v in d.items() if k in keys})

    def check_supported(self):
        c = self
        if c.head_dim != 128:
            raise GritHipError(f"native encoder: head_dim={c.head_dim}; only 128 (Mistral-7B shape) is built")
        if c.hidden_size % 64 or c.intermediate_size % 64 or (c.num_attention_heads * c.head_dim) % 64:
            raise GritHipError("native encoder: hidden/intermediate sizes must be multiples of 64")
        if c.num_attention_heads % c.num_key_value_heads:
            raise GritHipError("native encoder: num_attention_heads must be a multiple of num_key_value_heads")
        if c.num_local_experts and (c.num_local_experts not in (4, 8, 16) or c.num_experts_per_tok != 2):
            raise GritHipError(f"native encoder: MoE with {c.num_local_experts} experts / top-{c.num_experts_per_tok}; "
                               "4, 8 or 16 experts with top-2 routing (Mixtral) are built")


def swiglu_interleave(gate: torch.Tensor, up: torch.Tensor, block: int | None = None) -> torch.Tensor:
    """[I,H],[I,H] -> [2I,H]: ``block`` rows of gate followed by the same rows of up, repeated (GRIT_EPI_SWIGLU layout;
    ``block`` = grit_swiglu_block() of the built kernel)."""
    from . import _lib
    if block is None:
        block = _lib.load().grit_swiglu_block()
    I, H = gate.shape
    assert I % block == 0, f"intermediate size {I} must be a multiple of {block}"
    return torch.stack([gate.view(I // block, block, H), up.view(I // block, block, H)], dim=1).reshape(2 * I, H).contiguous()


def rope_tables(seq_len: int, head_dim: int, theta: float, round_bf16: bool, device) -> tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [S, d/2] fp32, built like MistralRotaryEmbedding (:93-126); the reference casts its tables to
    the model dtype (:124-125), reproduced with ``round_bf16``."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = torch.outer(torch.arange(seq_len, dtype=torch.float32), inv_freq)
    cos, sin = freqs.cos(), freqs.sin()
    if round_bf16:
        cos, sin = cos.to(BF16).float(), sin.to(BF16).float()
    return cos.contiguous().to(device), sin.contiguous().to(device)


class _Layer:
    __slots__ = ("wqkv", "wo", "wgu", "wdown", "ln1", "ln2", "wgate", "w13", "w2", "h16")     # h16: fp16 copies (wqkv, wo, wgu, wdown), built on first use


def sliding_window_keys(sliding_window, attn_implementation: str | None = "sdpa") -> int:
    """How many keys (its own included) a CAUSAL query sees under ``config.sliding_window``, per attention path of the reference; 0 = no
    window.  The reference's three paths do not agree (scripts/modeling_mistral_gritlm.py):

    * ``sdpa`` (its default): :1011-1016 hands NO sliding_window to `_prepare_4d_causal_attention_mask_for_sdpa` -- plain causal
      attention at every length;
    * ``eager``: :1022-1031 passes `sliding_window=config.sliding_window`; the mask of its pinned transformers 4.37.2 keeps
      ``sliding_window`` keys (later transformers releases keep one more -- set the engine's ``window_keys`` to override);
    * ``flash_attention_2``: `window_size=(W, W)` (:548, :570) once the sequence is longer than W -- ``sliding_window + 1`` keys.

    The bidirectional embedding path never applies the window (in no path of the reference)."""
    if not sliding_window:
        return 0
    impl = attn_implementation or "sdpa"
    if impl == "sdpa":
        return 0
    if impl == "eager":
        return int(sliding_window)
    if impl == "flash_attention_2":
        return int(sliding_window) + 1
    raise ValueError(f"unknown attn_implementation {attn_implementation!r}")


class MistralEncoderEngine:
    """Forward-only native engine (inference / GradCache pass 1)."""

    def __init__(self, cfg: EncoderConfig, device="cuda"):
        cfg.check_supported()
        self.cfg = cfg
        self.device = torch.device(device)
        self.layers: list[_Layer] = []
        self.embed = None
        self.norm = None
        self._ws = {}
        self._rope = {}
        self.rope_bf16 = True
        self.record_routing = None      # tests: set to a list to collect every MoE layer's selected experts [T,2]
        self.causal = False             # True: causal attention ('cc' embedding attention of the reference's attn string)
        self.window_keys = 0            # causal attention: keys a query sees (sliding_window_keys(); 0 = no window).  The bidirectional
                                        # path ignores it, as the reference
        self.precision = "bf16"         # one of PRECISIONS (module docstring above)
        self.f16_weight_stats = None    # f16_operands: {"subnormal": n, "overflow": n, "total": n} of the bf16 -> fp16 weight conversion

    # `residual_fp32` (rounds 3-4: a bool) is kept as a view of `precision`: True <-> the stream lives in fp32
    @property
    def residual_fp32(self) -> bool:
        return self.precision in ("fp32_residual", "f16_operands")

    @residual_fp32.setter
    def residual_fp32(self, v: bool):
        self.precision = "fp32_residual" if v else "bf16"

    def supported_precisions(self) -> tuple:
        """The policies this engine's model kind / attention mode is built for, in AUTO_LADDER order."""
        if self.cfg.num_local_experts:
            return ("f16_operands", "bf16")
        return AUTO_LADDER              # (bidirectional and -- round 6, grit_attn_causal_f16_fwd -- causal attention alike)

    def set_precision(self, precision: str):
        if precision not in PRECISIONS:
            raise ValueError(f"precision={precision!r}: one of {PRECISIONS}")
        if precision in F16_POLICIES:
            if self.cfg.num_local_experts and precision != "f16_operands":
                raise GritHipError(f"native encoder: precision='{precision}' is built for the dense (Mistral) MLP only "
                                   "(the sparse-MoE engine routes on the fp32 residual stream: use 'f16_operands')")
        if precision == "fp32_residual" and self.cfg.num_local_experts:
            raise GritHipError("native encoder: precision='fp32_residual' is built for the dense (Mistral) MLP only")
        self.precision = precision
        return self

    def _f16_weights(self, L: _Layer):
        """fp16 copies of a layer's GEMM weights (14.5 GB for the 7B shape, 93 GB for the 8x7B shape, next to 288 GB of HBM), converted on
        first use: (wqkv, wo, wgu, wdown) of a dense layer, (wqkv, wo, w13, w2) of a sparse-MoE layer.  bf16 -> fp16 is exact for normal fp16
        values; what is not exact is COUNTED: |w| < 2^-14 becomes subnormal (absolute error <= 3e-8), |w| >= 65520 would become inf and is
        refused.  The copies are keyed on the sources' storage and version counters: weights reloaded or updated in place are converted again."""
        src = (L.wqkv, L.wo, L.w13, L.w2) if self.cfg.num_local_experts else (L.wqkv, L.wo, L.wgu, L.wdown)
        key = tuple((w.data_ptr(), w._version) for w in src)
        h = getattr(L, "h16", None)
        if h is None or h[0] != key:
            st = self.f16_weight_stats or {"subnormal": 0, "overflow": 0, "total": 0}
            out = []
            for w in src:
                # (counted slice by slice: the [E,2I,H] expert stacks are 12 GB each and abs() would double that)
                for part in (w if w.dim() == 3 else (w,)):
                    a = part.abs()
                    st["subnormal"] += int(((a < 6.103515625e-05) & (a > 0)).sum())
                    st["overflow"] += int((a >= 65520.0).sum())
                    st["total"] += part.numel()
                    del a
                out.append(w.to(F16))
            if st["overflow"]:
                raise GritHipError(f"precision='{self.precision}': {st['overflow']} weights exceed the fp16 range (|w| >= 65520)")
            self.f16_weight_stats = st
            L.h16 = h = (key, tuple(out))
        return h[1]

    def f16_overflowed(self, clear: bool = True) -> bool:
        """True when a kernel of an fp16 policy produced a value beyond the fp16 range on this engine's device since the last clear (waits
        for the device's current stream: one 4-byte D2H copy)."""
        return bool(ops.f16_overflow_flag(self.device, clear))

    def check_f16_overflow(self, clear: bool = True) -> None:
        """Raise if a kernel of the f16_operands policy produced a value beyond the fp16 range since the last check (waits for the
        device's current stream: call it where the embeddings are copied to the host anyway)."""
        if self.precision in F16_POLICIES and self.f16_overflowed(clear):
            raise GritHipError(f"precision='{self.precision}': an activation exceeded the fp16 range (|v| >= 65520) in this forward pass; "
                               "the embeddings of this call are invalid -- run this model with "
                               + ("precision='f16_operands' (fp32 residual stream), " if self.precision == "f16_stream" else "")
                               + "precision='fp32_residual' or 'bf16'")

    def _embed_table(self):
        """embed_tokens in the residual stream's format: bf16 (also the source of the exact widening into an fp32 stream), or an fp16 copy
        (262 MB at V 32000) for the fp16 stream, converted on first use"""
        if self.precision != "f16_stream":
            return self.embed
        key = (self.embed.data_ptr(), self.embed._version, tuple(self.embed.shape))
        if getattr(self, "_embed16", None) is None or self._embed16[0] != key:       # (re-converted when the table is reloaded / updated in place)
            if bool((self.embed.abs() >= 65520.0).any()):
                raise GritHipError("precision='f16_stream': embedding weights exceed the fp16 range")
            self._embed16 = (key, self.embed.to(F16))
        return self._embed16[1]

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_state_dict(cls, cfg: EncoderConfig, sd: dict, device="cuda", prefix: str = ""):
        """``sd``: HF MistralModel names (embed_tokens.weight, layers.N.self_attn.q_proj.weight, ...), any float dtype."""
        eng = cls(cfg, device)
        g = lambda k: sd[prefix + k].detach().to(device=eng.device, dtype=BF16)
        eng.embed = g("embed_tokens.weight").contiguous()
        for i in range(cfg.num_hidden_layers):
            p = f"layers.{i}."
            L = _Layer()
            L.wqkv = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                g(p + "self_attn.v_proj.weight")], dim=0).contiguous()
            L.wo = g(p + "self_attn.o_proj.weight").contiguous()
            if cfg.num_local_experts and (prefix + p + "mlp.experts.gate_up_proj") in sd:
                # transformers >= 5 layout: fused [E, 2I, H] = [gate rows | up rows] per expert, down_proj [E, H, I]
                gu, I = g(p + "mlp.experts.gate_up_proj"), cfg.intermediate_size
                L.wgate = g(p + "mlp.gate.weight").contiguous()
                L.w13 = torch.stack([swiglu_interleave(gu[e, :I], gu[e, I:]) for e in range(cfg.num_local_experts)]).contiguous()
                L.w2 = g(p + "mlp.experts.down_proj").contiguous()
            elif cfg.num_local_experts:
                m = p + "block_sparse_moe."        # reference / checkpoint layout (modeling_mixtral_gritlm.py:803-805, :834-836)
                L.wgate = g(m + "gate.weight").contiguous()
                L.w13 = torch.stack([swiglu_interleave(g(f"{m}experts.{e}.w1.weight"), g(f"{m}experts.{e}.w3.weight"))
                                     for e in range(cfg.num_local_experts)]).contiguous()                     # [E, 2I, H]
                L.w2 = torch.stack([g(f"{m}experts.{e}.w2.weight") for e in range(cfg.num_local_experts)]).contiguous()   # [E, H, I]
            else:
                L.wgu = swiglu_interleave(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"))
                L.wdown = g(p + "mlp.down_proj.weight").contiguous()
            L.ln1 = g(p + "input_layernorm.weight").contiguous()
            L.ln2 = g(p + "post_attention_layernorm.weight").contiguous()
            eng.layers.append(L)
        eng.norm = g("norm.weight").contiguous()
        return eng

    @classmethod
    def random_init(cls, cfg: EncoderConfig, device="cuda", seed: int = 0, std: float = 0.02):
        """Random weights of the architecture, generated on the device (bench: no checkpoints offline)."""
        eng = cls(cfg, device)
        gen = torch.Generator(device=eng.device).manual_seed(seed)
        H, I, d = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
        nq, nkv = cfg.num_attention_heads, cfg.num_key_value_heads

        def lin(o, i):
            return (torch.randn((o, i), generator=gen, device=eng.device, dtype=torch.float32) * std).to(BF16)

        def nrm():
            return (1.0 + 0.1 * torch.randn((H,), generator=gen, device=eng.device, dtype=torch.float32)).to(BF16)

        eng.embed = lin(cfg.vocab_size, H)
        for _ in range(cfg.num_hidden_layers):
            L = _Layer()
            L.wqkv = lin((nq + 2 * nkv) * d, H)
            L.wo = lin(H, nq * d)
            if cfg.num_local_experts:
                E = cfg.num_local_experts
                L.wgate = (torch.randn((E, H), generator=gen, device=eng.device, dtype=torch.float32) * 0.5).to(BF16)
                L.w13 = torch.empty((E, 2 * I, H), dtype=BF16, device=eng.device)
                L.w2 = torch.empty((E, H, I), dtype=BF16, device=eng.device)
                for e in range(E):            # expert by expert: the fp32 staging tensor stays small
                    L.w13[e].copy_(lin(2 * I, H)); L.w2[e].copy_(lin(H, I))
            else:
                L.wgu = lin(2 * I, H)          # already "interleaved": random rows
                L.wdown = lin(H, I)
            L.ln1, L.ln2 = nrm(), nrm()
            eng.layers.append(L)
        eng.norm = nrm()
        return eng

    def replica(self, device) -> "MistralEncoderEngine":
        """A copy of this engine on another GPU (the repacked weights are copied device to device; precision policy, attention mode and
        window are inherited): one replica per visible GPU is what in-process multi-GPU encode runs on (gritlm_amd/gritlm.py)."""
        eng = type(self)(self.cfg, device)
        mv = lambda t: None if t is None else t.to(eng.device, copy=True)       # a real copy even when the replica shares the device (tests)
        eng.embed, eng.norm = mv(self.embed), mv(self.norm)
        for L in self.layers:
            R = _Layer()
            for k in _Layer.__slots__:
                if hasattr(L, k) and k != "h16":                            # (fp16 copies are rebuilt on the replica's first use)
                    setattr(R, k, mv(getattr(L, k)))
            eng.layers.append(R)
        eng.causal, eng.window_keys, eng.precision = self.causal, self.window_keys, self.precision
        return eng

    def to_hf_state_dict(self) -> dict:
        """The engine's (repacked) weights under the Hugging Face ``MistralModel`` names (inverse of ``from_state_dict``: the fused QKV
        matrix split back, the gate / up rows de-interleaved); dense MLP only.  bench.py / the tests load the STOCK module from it to
        compare the two implementations on identical parameters."""
        c = self.cfg
        if c.num_local_experts:
            raise NotImplementedError("to_hf_state_dict: dense (Mistral) engines only")
        from . import _lib
        blk = _lib.load().grit_swiglu_block()
        nq, nkv, d, I, H = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.intermediate_size, c.hidden_size
        sd = {"embed_tokens.weight": self.embed, "norm.weight": self.norm}
        for i, L in enumerate(self.layers):
            p = f"layers.{i}."
            sd[p + "self_attn.q_proj.weight"] = L.wqkv[:nq * d]
            sd[p + "self_attn.k_proj.weight"] = L.wqkv[nq * d:(nq + nkv) * d]
            sd[p + "self_attn.v_proj.weight"] = L.wqkv[(nq + nkv) * d:]
            sd[p + "self_attn.o_proj.weight"] = L.wo
            gu = L.wgu.view(I // blk, 2, blk, H)
            sd[p + "mlp.gate_proj.weight"] = gu[:, 0].reshape(I, H)
            sd[p + "mlp.up_proj.weight"] = gu[:, 1].reshape(I, H)
            sd[p + "mlp.down_proj.weight"] = L.wdown
            sd[p + "input_layernorm.weight"] = L.ln1
            sd[p + "post_attention_layernorm.weight"] = L.ln2
        return sd

    # ------------------------------------------------------------------ buffers
    def _workspace(self, T: int):
        """Activation buffers for T token rows: one allocation sized for the largest T seen, handed out as row-slices
        (ragged / packed batches change T every call)."""
        cap = self._ws.get("cap", 0)
        if cap and self._ws.get("policy") != self.precision:
            cap = 0                       # the precision policy changed: the residual stream / the operands have another dtype
        if cap < T:
            c, dev = self.cfg, self.device
            qkv_w = (c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim
            self._ws.clear()
            opd = F16 if self.precision in F16_POLICIES else BF16
            mk = lambda n: torch.empty((T, n), dtype=opd, device=dev)
            h = torch.empty((T, c.hidden_size), dtype=torch.float32 if self.residual_fp32 else opd, device=dev)
            self._ws.update(cap=T, policy=self.precision, h=h, x=mk(c.hidden_size), qkv=mk(qkv_w), ctx=mk(c.num_attention_heads * c.head_dim))
            # last_hidden_state stays bf16 in every policy (the pooling kernels' input; one rounding of the final RMSNorm, averaged over the
            # sequence by the pooling: 3e-7 of 1 - cos, profiles/r05_precision_budget.json "only_out_bf16")
            self._ws.update(xo=self._ws["x"] if opd == BF16 else torch.empty((T, c.hidden_size), dtype=BF16, device=dev))
            if c.num_local_experts:           # every token visits two experts: 2T rows of expert activations
                self._ws.update(act2=torch.empty((2 * T, c.intermediate_size), dtype=opd, device=dev),
                                y2=torch.empty((2 * T, c.hidden_size), dtype=opd, device=dev))
            else:
                self._ws.update(act=mk(c.intermediate_size))
        return {k: (v[:2 * T] if k in ("act2", "y2") else v[:T]) for k, v in self._ws.items() if k not in ("cap", "policy")}

    def _mlp(self, L: _Layer, x: torch.Tensor, h: torch.Tensor, ws: dict):
        """h += MLP(x) in place (x = post-attention RMSNorm output): dense SwiGLU MLP or Mixtral's sparse-MoE block."""
        if not self.cfg.num_local_experts:
            _, _, wgu, wdown = self._weights(L)
            ops.gemm_nt(x, wgu, out=ws["act"], epilogue=EPI_SWIGLU)
            ops.gemm_nt(ws["act"], wdown, out=h, epilogue=self._epi_res(), residual=h)
            return
        T = x.shape[0]
        if self.precision == "f16_operands":
            # the routing decision in fp32 on the residual stream itself (post-attention RMSNorm folded in: nothing rounded), the experts on
            # fp16 operands, the weighted sum added into the fp32 stream
            _, _, w13, w2 = self._f16_weights(L)
            experts, weights, counts, row_token, rows = ops.moe_route_f32(h, L.ln2, self.cfg.rms_norm_eps, L.wgate)
        else:
            if self.residual_fp32:
                raise GritHipError("native encoder: precision='fp32_residual' is built for the dense (Mistral) MLP only")
            w13, w2 = L.w13, L.w2
            experts, weights, counts, row_token, rows = ops.moe_route(x, L.wgate)
        if self.record_routing is not None:
            self.record_routing.append(experts.clone())
        ops.gemm_nt_grouped(x, w13, counts, 2 * T, out=ws["act2"], epilogue=EPI_SWIGLU, a_rows=row_token)
        ops.gemm_nt_grouped(ws["act2"], w2, counts, 2 * T, out=ws["y2"])
        ops.moe_combine(ws["y2"], rows, weights, h, out=h)

    def _epi_res(self) -> int:
        return EPI_RESIDUAL_F32 if self.residual_fp32 else EPI_RESIDUAL

    def _weights(self, L: _Layer):
        """(wqkv, wo, wgu, wdown) in the operand format of the current policy"""
        if self.precision in F16_POLICIES:
            return self._f16_weights(L)              # (a sparse-MoE layer: (wqkv, wo, w13, w2))
        return (L.wqkv, L.wo, getattr(L, "wgu", None), getattr(L, "wdown", None))        # (MoE layers have no dense MLP weights)

    def _window(self, S: int) -> int:
        """``window`` argument of the attention kernels for sequences of up to S tokens (0: every earlier key is inside the window)."""
        return int(self.window_keys) if self.causal and 0 < self.window_keys < S else 0

    def _rope_tables(self, S: int):
        rounded = self.rope_bf16 and self.precision not in F16_POLICIES    # fp16 operands: the unrounded fp32 tables (fp32 rotation)
        t = self._rope.get((S, rounded))
        if t is None:
            t = rope_tables(S, self.cfg.head_dim, self.cfg.rope_theta, rounded, self.device)
            self._rope[(S, rounded)] = t
        return t

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor | None, attention_mask: torch.Tensor | None = None, borrow: bool = False,
                return_kv: bool = False, inputs_embeds: torch.Tensor | None = None, layer_range: tuple | None = None,
                final_norm: bool = True, kv_dtype: torch.dtype | None = BF16):
        """last_hidden_state [B,S,H] bf16 (after the final RMSNorm), is_causal=False semantics.

        ``borrow=True`` returns a view of the engine's workspace (valid until the next forward).
        ``return_kv=True`` additionally returns, per layer, the post-RoPE keys and the values as
        ``(k [B,nkv,S,d], v [B,nkv,S,d])`` -- what ``use_cache=True`` hands back in the reference
        (gritlm/gritlm.py:131-140; RAG doc caching, rag/eval.py:132-142).  ``kv_dtype``: bf16 (the reference's cache format, the default) or
        None = the operand format of the policy: under the fp16 policies the K / V the attention itself read, fp16, rounded ONCE from the fp32
        accumulator of the q|k|v GEMM's epilogue (bf16 from them is a second rounding: at most 1/16 bf16 ulp more than a direct one).
        ``inputs_embeds`` [B,S,H] replaces the embedding lookup (the reference's forward takes it too, modeling_mistral_gritlm.py:944, :993-994);
        ``layer_range=(a, b)`` runs decoder layers a .. b-1 only and ``final_norm=False`` returns the residual stream itself: together they
        push a GIVEN hidden state through chosen layers -- the teacher-forced per-layer parity of the Mixtral leg (tools/mixtral_bench.py)."""
        c = self.cfg
        if inputs_embeds is not None:
            B, S = inputs_embeds.shape[:2]
        else:
            B, S = input_ids.shape
        T = B * S
        window = self._window(S)
        ids = None if inputs_embeds is not None else input_ids.to(device=self.device, dtype=torch.int64).contiguous().view(-1)
        if attention_mask is None:
            attention_mask = torch.ones((B, S), dtype=torch.int64, device=self.device)
        mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
        if self.precision in F16_POLICIES:
            self.set_precision(self.precision)        # (re-checks the model kind / attention mode the policy is built for)
        ws = self._workspace(T)
        h, x, qkv, ctx, xo = ws["h"], ws["x"], ws["qkv"], ws["ctx"], ws["xo"]
        nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        cos, sin = self._rope_tables(S)
        bits = ops.mask_pack(mask)
        if inputs_embeds is not None:
            h.copy_(inputs_embeds.to(self.device).reshape(T, c.hidden_size))
        else:
            ops.embed_gather(self._embed_table(), ids, out=h)
        kv = []
        for L in (self.layers if layer_range is None else self.layers[layer_range[0]:layer_range[1]]):
            wqkv, wo = self._weights(L)[:2]
            ops.rmsnorm(h, L.ln1, eps, out=x)
            ops.gemm_nt_rope(x, wqkv, cos, sin, (nq + nkv) * d, S=S, out=qkv)         # q/k/v projections + RoPE in the epilogue
            if return_kv:
                kvw = qkv.view(B, S, nq + 2 * nkv, d)
                kdt = qkv.dtype if kv_dtype is None else kv_dtype
                kv.append((kvw[:, :, nq:nq + nkv].permute(0, 2, 1, 3).to(kdt).contiguous(),
                           kvw[:, :, nq + nkv:].permute(0, 2, 1, 3).to(kdt).contiguous()))
            ops.attn_bidir(qkv, bits, B, S, nq, nkv, d, out=ctx, causal=self.causal, window=window)
            ops.gemm_nt(ctx, wo, out=h, epilogue=self._epi_res(), residual=h)
            ops.rmsnorm(h, L.ln2, eps, out=x)
            self._mlp(L, x, h, ws)
        if final_norm:
            ops.rmsnorm(h, self.norm, eps, out=xo)
            out = xo.view(B, S, c.hidden_size)
        else:
            out = h.view(B, S, c.hidden_size)
        out = out if borrow else out.clone()
        return (out, kv) if return_kv else out

    __call__ = forward

    # ------------------------------------------------------------------ packed (un-padded) forward
    @staticmethod
    def is_right_padded(attention_mask: torch.Tensor) -> bool:
        """True when every row of the mask is 1...10...0 (what a right-padding tokenizer produces)."""
        m = attention_mask != 0
        lens = m.sum(dim=1, keepdim=True)
        ar = torch.arange(m.shape[1], device=m.device).unsqueeze(0)
        return bool(((ar < lens) == m).all())

    @torch.no_grad()
    def encode_pooled(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, method: str, normalize: bool,
                      instr_len: torch.Tensor | None = None, packed: bool | None = None) -> torch.Tensor:
        """pool(normalise(encoder(ids))) -> [B,H] fp32.  With a right-padded batch the padding is dropped before the first
        kernel ("packed" rows, cu_seqlens): GEMMs, norms and attention only ever see real tokens -- the reference's SDPA path
        computes every padded row.  Results are bit-identical to the padded path.

        HOST tensors (what the tokenizer returns) are the preferred input: the batch geometry -- right-padded or not, row lengths,
        cu_seqlens, the packed token list -- is then derived on the host, only real tokens cross PCIe, and the call issues its launches
        without a single device synchronisation (the host runs ahead: the next batch is tokenised, and in-process multi-GPU encode
        feeds the other replicas, while this one computes).  Device tensors take the same path with three small syncs."""
        c = self.cfg
        B, S = input_ids.shape
        window = self._window(S)
        with torch.cuda.device(self.device):
            if attention_mask.device.type == "cpu" and input_ids.device.type == "cpu":
                m = attention_mask != 0
                lens_h = m.sum(dim=1)
                if packed is None:
                    packed = bool(((torch.arange(S).unsqueeze(0) < lens_h.unsqueeze(1)) == m).all()) and bool((lens_h > 0).all())
                if packed:
                    pids = input_ids.to(torch.int64)[m].contiguous().to(self.device, non_blocking=True)
                    pos = torch.arange(S, dtype=torch.int32).unsqueeze(0).expand(B, S)[m].contiguous().to(self.device, non_blocking=True)
                    cu_h = torch.zeros((B + 1,), dtype=torch.int32)
                    cu_h[1:] = torch.cumsum(lens_h, dim=0)
                    cu = cu_h.to(self.device, non_blocking=True)
                    T, max_len = int(cu_h[-1]), int(lens_h.max())
            else:
                attention_mask = attention_mask.to(device=self.device, dtype=torch.int64)
                input_ids = input_ids.to(device=self.device, dtype=torch.int64)
                if packed is None:
                    packed = self.is_right_padded(attention_mask) and bool((attention_mask.sum(dim=1) > 0).all())
                if packed:
                    lens = attention_mask.sum(dim=1).to(torch.int32)
                    cu = torch.zeros((B + 1,), dtype=torch.int32, device=self.device)
                    cu[1:] = torch.cumsum(lens, dim=0)
                    keep = attention_mask.bool()
                    pids = input_ids[keep].contiguous()                              # row-major order == sequence order
                    pos = (torch.arange(S, device=self.device, dtype=torch.int32).unsqueeze(0).expand(B, S))[keep].contiguous()
                    T, max_len = int(pids.numel()), int(lens.max().item())
            if instr_len is not None:
                instr_len = instr_len.to(self.device)
            if not packed:
                mask = attention_mask.to(device=self.device, dtype=torch.int64).contiguous()
                h = self.forward(input_ids.to(device=self.device, dtype=torch.int64), mask, borrow=True)
                return ops.pool_norm(h, mask, method, normalize, instr_len)
            if self.precision in F16_POLICIES:
                self.set_precision(self.precision)
            ws = self._workspace(T)
            h, x, qkv, ctx, xo = ws["h"], ws["x"], ws["qkv"], ws["ctx"], ws["xo"]
            nq, nkv, d, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
            cos, sin = self._rope_tables(S)
            ops.embed_gather(self._embed_table(), pids, out=h)
            for L in self.layers:
                wqkv, wo = self._weights(L)[:2]
                ops.rmsnorm(h, L.ln1, eps, out=x)
                ops.gemm_nt_rope(x, wqkv, cos, sin, (nq + nkv) * d, positions=pos, out=qkv)
                ops.attn_bidir_varlen(qkv, cu, max_len, nq, nkv, d, out=ctx, causal=self.causal, window=window)
                ops.gemm_nt(ctx, wo, out=h, epilogue=self._epi_res(), residual=h)
                ops.rmsnorm(h, L.ln2, eps, out=x)
                self._mlp(L, x, h, ws)
            ops.rmsnorm(h, self.norm, eps, out=xo)
            return ops.pool_norm_varlen(xo, cu, method, normalize, instr_len)

    def flops_per_token(self, S: int) -> float:
        """Algorithmic forward FLOPs per token: projections + MLP + attention core."""
        c = self
        H, I, L = self.cfg.hidden_size, self.cfg.intermediate_size, self.cfg.num_hidden_layers
        hkv = self.cfg.num_key_value_heads * self.cfg.head_dim
        hq = self.cfg.num_attention_heads * self.cfg.head_dim
        mlp = 3 * H * I * (self.cfg.num_experts_per_tok if self.cfg.num_local_experts else 1) + H * self.cfg.num_local_experts
        return 2.0 * L * (H * (hq + 2 * hkv) + hq * H + mlp) + 4.0 * L * S * hq
