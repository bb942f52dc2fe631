"""CPU oracle for the GritLM embedding-encode / contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gritlm_amd/`` may import this
module: it is the *checker* for the HIP path (tests/, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py``), never the thing shipped or
measured.  The product path fails loudly when ``libgritlm_hip.so`` is missing.

It is a plain numpy restatement (fp32 storage, fp64 accumulation where cheap)
of the reference algorithm; every function cites the reference file:line it
follows (paths relative to the upstream checkout, ``/root/reference``).

Parity pin: the reference ships no tests/golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference's own
Python run in the build container - ``tests/golden/make_golden.py`` imports
``/root/reference`` (gritlm.GritLM.pooling, scripts/modeling_mistral_gritlm.py
MistralModel(is_causal=False), training.model.DistributedContrastiveLoss,
GradCache) and stores input/output fixtures under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` replays them through this file.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


# ----------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, the rounding torch uses for .to(bf16))
# ----------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest bf16 (RNE) and return them as fp32."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) >> 16) << 16
    out = r.astype(np.uint32).view(F32).reshape(x.shape)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, x, out)
    return out


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> uint16 bf16 bit pattern (RNE)."""
    return (bf16_round(x).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(F32)


# ----------------------------------------------------------------------------
# Encoder pieces  (scripts/modeling_mistral_gritlm.py)
# ----------------------------------------------------------------------------
def rmsnorm(x: np.ndarray, weight: np.ndarray, eps: float, emulate_bf16: bool = False) -> np.ndarray:
    """MistralRMSNorm.forward, scripts/modeling_mistral_gritlm.py:84-89.

    fp32 ``x * rsqrt(mean(x^2) + eps)``, cast to the input dtype, THEN
    multiply by the weight.  With ``emulate_bf16`` both casts are rounded to
    bf16 exactly where the reference rounds them for a bf16 model.
    """
    x = x.astype(F32)
    var = np.mean(x.astype(F64) ** 2, axis=-1, keepdims=True)
    y = (x * (1.0 / np.sqrt(var + eps))).astype(F32)
    if emulate_bf16:
        y = bf16_round(y)
        return bf16_round(weight.astype(F32) * y)
    return weight.astype(F32) * y


def rope_tables(seq_len: int, head_dim: int, theta: float) -> tuple[np.ndarray, np.ndarray]:
    """MistralRotaryEmbedding, scripts/modeling_mistral_gritlm.py:93-126.

    inv_freq = 1/theta^(2i/d); freqs = outer(arange(S), inv_freq);
    emb = cat(freqs, freqs); cos/sin tables [S, d] fp32.
    """
    inv_freq = 1.0 / (theta ** (np.arange(0, head_dim, 2, dtype=F32) / head_dim))
    t = np.arange(seq_len, dtype=F32)
    freqs = np.outer(t, inv_freq).astype(F32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def rotate_half(x: np.ndarray) -> np.ndarray:
    """scripts/modeling_mistral_gritlm.py:130-134."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """apply_rotary_pos_emb, scripts/modeling_mistral_gritlm.py:138-163.

    x: [B, heads, S, d]; positions are arange(S) (:984-989).
    """
    return x * cos[None, None] + rotate_half(x) * sin[None, None]


def causal_window_mask(S: int, window: int = 0) -> np.ndarray:
    """[S, S] bool, True where query i sees key j: j <= i, and with ``window`` > 0 additionally i - j < window (``window`` = number of
    keys a query sees, its own included).

    The sliding-window causal mask is built by a dependency that is not vendored in the reference: transformers (pinned ==4.37.2 in the
    reference's requirements), `modeling_attn_mask_utils.AttentionMaskConverter._make_causal_mask`, called from
    `_prepare_4d_causal_attention_mask(..., sliding_window=config.sliding_window)` at scripts/modeling_mistral_gritlm.py:1005-1031.
    Its published algorithm: lower-triangular mask, then fill finfo.min wherever `1 - triu(ones, diagonal=-sliding_window + 1)` is set,
    i.e. j < i - sliding_window + 1 -> window = sliding_window keys.  Later transformers releases (the one installed in the build
    container generated tests/golden/sliding_window_gqa.npz) mask `tril(ones, diagonal=-sliding_window - 1)`, i.e. j <= i - W - 1
    -> window = sliding_window + 1 keys, which is also what the reference's flash-attention path keeps (window_size=(W, W), :548 / :570).
    The fixture records which of the two its generating run used."""
    i = np.arange(S)[:, None]
    j = np.arange(S)[None, :]
    m = j <= i
    if window and window > 0:
        m &= (i - j) < window
    return m


def masked_softmax(scores, key_mask, causal=False, window=0):
    """softmax over the keys a query is allowed to see (scores [B, H, S, S]).  A query row with NO allowed key -- a padding row behind a
    sliding window -- gets p = 0 (the reference's finfo.min mask gives such rows a uniform distribution; they are padding either way,
    and a zero row keeps NaN out of the padded keys of the next layer)."""
    S = scores.shape[-1]
    allowed = np.ones((1, 1, S, S), dtype=bool)
    if key_mask is not None:
        allowed = allowed & key_mask.astype(bool)[:, None, None, :]
    if causal:
        allowed = allowed & causal_window_mask(S, window)[None, None]
    scores = np.where(allowed, scores, -np.inf)
    m = scores.max(axis=-1, keepdims=True)
    p = np.exp(scores - np.where(np.isfinite(m), m, 0.0))
    den = p.sum(axis=-1, keepdims=True)
    return p / np.where(den > 0, den, 1.0)


def attention_bidirectional(q, k, v, key_mask, acc_dtype=F64, causal=False, window=0):
    """Attention core with is_causal=False and a key-padding mask (``causal=True``: additionally key <= query, the mask of
    `_prepare_4d_causal_attention_mask(_for_sdpa)`, :1005-1016 / :1021-1031 -- the generative branch; ``window``: causal_window_mask).

    MistralSdpaAttention.forward scripts/modeling_mistral_gritlm.py:627-705
    (repeat_kv :182-191, SDPA :690-698) with the additive mask of
    `_prepare_4d_attention_mask(_for_sdpa)` (:1017-1020, :1033-1036):
    finfo.min on padded KEYS only, every query row (padded or not) is computed.

    q: [B, Hq, S, d]; k, v: [B, Hkv, S, d]; key_mask: [B, S] (0/1).
    Returns [B, S, Hq*d].  ``acc_dtype`` float64 for tests, float32 for the timed CPU baseline.
    """
    B, Hq, S, d = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    k = np.repeat(k, rep, axis=1).astype(acc_dtype)
    v = np.repeat(v, rep, axis=1).astype(acc_dtype)
    scores = np.matmul(q.astype(acc_dtype), k.transpose(0, 1, 3, 2)) / np.sqrt(d).astype(acc_dtype)
    p = masked_softmax(scores, key_mask, causal, window)
    out = np.matmul(p, v)
    return out.transpose(0, 2, 1, 3).reshape(B, S, Hq * d).astype(F32)


def silu(x):
    return x / (1.0 + np.exp(-x))


def mlp(x, w_gate, w_up, w_down):
    """MistralMLP.forward, scripts/modeling_mistral_gritlm.py:177-178."""
    g = x @ w_gate.T
    u = x @ w_up.T
    return (silu(g) * u) @ w_down.T


def mistral_encode(weights: dict, cfg: dict, input_ids: np.ndarray, attention_mask: np.ndarray | None,
                   emulate_bf16: bool = False, return_layers: bool = False, acc_dtype=F64, causal: bool = False, window: int = 0):
    """MistralModel.forward with is_causal=False, scripts/modeling_mistral_gritlm.py:936-1096.

    ``weights`` uses the HF state_dict names of MistralModel (``embed_tokens.weight``,
    ``layers.N.self_attn.q_proj.weight`` ... ``norm.weight``) as fp32 numpy arrays.
    ``cfg``: hidden_size, num_hidden_layers, num_attention_heads, num_key_value_heads,
    intermediate_size, rms_norm_eps, rope_theta (head_dim = hidden/heads).

    ``emulate_bf16`` rounds to bf16 at the points where a bf16 reference model rounds
    (after every Linear / norm / residual add / RoPE / attention output).
    """
    H = cfg["hidden_size"]; L = cfg["num_hidden_layers"]
    nh = cfg["num_attention_heads"]; nkv = cfg["num_key_value_heads"]
    d = cfg.get("head_dim") or H // nh
    eps = cfg["rms_norm_eps"]; theta = cfg["rope_theta"]
    rnd = bf16_round if emulate_bf16 else (lambda a: a.astype(F32))
    B, S = input_ids.shape
    h = weights["embed_tokens.weight"][input_ids].astype(F32)            # :994
    cos, sin = rope_tables(S, d, theta)
    if emulate_bf16:
        cos, sin = bf16_round(cos), bf16_round(sin)                       # :124-125
    layers = []
    routing = []
    kv_layers = []
    for li in range(L):                                                   # :1045-1071
        p = f"layers.{li}."
        res = h
        x = rmsnorm(h, weights[p + "input_layernorm.weight"], eps, emulate_bf16)      # :757
        q = rnd(x @ weights[p + "self_attn.q_proj.weight"].T)                          # :655-657
        k = rnd(x @ weights[p + "self_attn.k_proj.weight"].T)
        v = rnd(x @ weights[p + "self_attn.v_proj.weight"].T)
        q = q.reshape(B, S, nh, d).transpose(0, 2, 1, 3)
        k = k.reshape(B, S, nkv, d).transpose(0, 2, 1, 3)
        v = v.reshape(B, S, nkv, d).transpose(0, 2, 1, 3)
        q = rnd(apply_rope(q, cos, sin)); k = rnd(apply_rope(k, cos, sin))             # :666-668
        kv_layers.append((k.copy(), v.copy()))                                         # what use_cache=True hands back (:671-673)
        a = rnd(attention_bidirectional(q, k, v, attention_mask, acc_dtype, causal, window))      # :690-698
        a = rnd(a @ weights[p + "self_attn.o_proj.weight"].T)                          # :703
        h = rnd(res + a)                                                               # :769
        res = h
        x = rmsnorm(h, weights[p + "post_attention_layernorm.weight"], eps, emulate_bf16)  # :773
        if (p + "block_sparse_moe.gate.weight") in weights:                            # Mixtral: modeling_mixtral_gritlm.py:943
            m, sel = moe_block(x.reshape(B * S, H), weights, p + "block_sparse_moe.", cfg.get("num_experts_per_tok", 2), emulate_bf16)
            m = m.reshape(B, S, H)
            routing.append(sel.reshape(B, S, -1))
        else:
            g = rnd(x @ weights[p + "mlp.gate_proj.weight"].T)
            u = rnd(x @ weights[p + "mlp.up_proj.weight"].T)
            m = rnd(rnd(silu(g)) * u)
            m = rnd(m @ weights[p + "mlp.down_proj.weight"].T)                         # :177-178
        h = rnd(res + m)                                                               # :775
        if return_layers:
            layers.append(h.copy())
    out = rmsnorm(h, weights["norm.weight"], eps, emulate_bf16)                        # :1079
    if return_layers == "routing":
        return out, routing
    if return_layers == "kv":
        return out, kv_layers
    if return_layers:
        return out, layers
    return out


def mistral_continue(weights: dict, cfg: dict, prefix_kv: list, prefix_len: int, cont_ids: np.ndarray, lm_head: np.ndarray):
    """Causal continuation on top of a cached prefix, one sequence: what ``model.generate(cont_ids, past_key_values=prefix_kv)`` computes
    in the reference's RAG flow (rag/eval.py:237-246, :296-302) -- the prefix K/V (per layer [1,Hkv,S,d], post-RoPE; e.g. from the
    bidirectional document pass, gritlm.py:131-140) are attended to in full, the continuation tokens causally, positions continue at
    prefix_len.  Returns the fp32 logits [P, V] of every continuation position."""
    H = cfg["hidden_size"]; L = cfg["num_hidden_layers"]
    nh = cfg["num_attention_heads"]; nkv = cfg["num_key_value_heads"]
    d = cfg.get("head_dim") or H // nh
    eps = cfg["rms_norm_eps"]
    P = int(cont_ids.shape[0])
    h = weights["embed_tokens.weight"][cont_ids][None].astype(F32)                       # [1,P,H]
    cos, sin = rope_tables(prefix_len + P, d, cfg["rope_theta"])
    cos, sin = cos[prefix_len:], sin[prefix_len:]
    for li in range(L):
        p = f"layers.{li}."
        x = rmsnorm(h, weights[p + "input_layernorm.weight"], eps)
        q = (x @ weights[p + "self_attn.q_proj.weight"].T).reshape(1, P, nh, d).transpose(0, 2, 1, 3)
        k = (x @ weights[p + "self_attn.k_proj.weight"].T).reshape(1, P, nkv, d).transpose(0, 2, 1, 3)
        v = (x @ weights[p + "self_attn.v_proj.weight"].T).reshape(1, P, nkv, d).transpose(0, 2, 1, 3)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
        pk, pv = prefix_kv[li]
        kk = np.concatenate([pk[:, :, :prefix_len].astype(F64), k.astype(F64)], axis=2)
        vv = np.concatenate([pv[:, :, :prefix_len].astype(F64), v.astype(F64)], axis=2)
        kk, vv = np.repeat(kk, nh // nkv, axis=1), np.repeat(vv, nh // nkv, axis=1)
        sc = np.matmul(q.astype(F64), kk.transpose(0, 1, 3, 2)) / np.sqrt(d)            # [1,nh,P,prefix+P]
        allowed = np.concatenate([np.ones((P, prefix_len), dtype=bool), np.tril(np.ones((P, P), dtype=bool))], axis=1)
        sc = sc + np.where(allowed, 0.0, -np.inf)[None, None]
        sc = sc - sc.max(-1, keepdims=True)
        pr = np.exp(sc); pr /= pr.sum(-1, keepdims=True)
        a = np.matmul(pr, vv).transpose(0, 2, 1, 3).reshape(1, P, nh * d).astype(F32)
        h = h + a @ weights[p + "self_attn.o_proj.weight"].T
        x = rmsnorm(h, weights[p + "post_attention_layernorm.weight"], eps)
        if cfg.get("num_local_experts"):                                                 # MixtralSparseMoeBlock (modeling_mixtral_gritlm.py:839-882)
            m = moe_block(x.reshape(P, H), weights, p + "block_sparse_moe.", cfg.get("num_experts_per_tok", 2))[0].reshape(1, P, H)
        else:
            m = (silu(x @ weights[p + "mlp.gate_proj.weight"].T) * (x @ weights[p + "mlp.up_proj.weight"].T)) @ weights[p + "mlp.down_proj.weight"].T
        h = (h + m).astype(F32)
    out = rmsnorm(h, weights["norm.weight"], eps)
    return (out[0] @ lm_head.T).astype(F32)


def next_token_loss(logits: np.ndarray, labels: np.ndarray, loss_gen_type: str = "mixed", loss_gen_factor: float = 1.0) -> float:
    """NextTokenLoss.__call__, gritlm/training/model.py:93-107: tokens < n predict n, ignore_index -100;
    'mixed' = mean over the scored tokens of the batch, 'token' = sum / batch size; times loss_gen_factor.
    logits [B,S,V] (fp32, as after logits.float()), labels [B,S]."""
    sl = logits[:, :-1].astype(F64).reshape(-1, logits.shape[-1])
    tl = labels[:, 1:].reshape(-1)
    keep = tl != -100
    z = sl[keep]
    mx = z.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(z - mx).sum(axis=1))
    nll = lse - z[np.arange(z.shape[0]), tl[keep]]
    if loss_gen_type == "token":
        return float(nll.sum() / labels.shape[0] * loss_gen_factor)
    if loss_gen_type == "mixed":
        return float(nll.mean() * loss_gen_factor)
    raise ValueError(f"Invalid loss_gen_type: {loss_gen_type}")


def moe_router(x: np.ndarray, gate_w: np.ndarray, top_k: int = 2, emulate_bf16: bool = False):
    """MixtralSparseMoeBlock routing, scripts/modeling_mixtral_gritlm.py:843-850: gate Linear (model dtype) -> softmax in
    fp32 -> top-k (descending, lowest index first on ties like torch.topk on CPU) -> renormalise -> cast to the model dtype.
    x [T,H] -> (weights [T,k] fp32 (bf16-representable when emulate_bf16), experts [T,k] int64, logits [T,E])."""
    rnd = bf16_round if emulate_bf16 else (lambda a: a.astype(F32))
    logits = rnd(x.astype(F32) @ gate_w.astype(F32).T)                                  # :843
    z = logits.astype(F32) - logits.max(axis=1, keepdims=True)
    pr = np.exp(z); pr = (pr / pr.sum(axis=1, keepdims=True)).astype(F32)               # :845 softmax(dtype=float)
    sel = np.argsort(-pr, axis=1, kind="stable")[:, :top_k]                             # :846
    w = np.take_along_axis(pr, sel, axis=1)
    w = (w / w.sum(axis=1, keepdims=True)).astype(F32)                                  # :847
    return rnd(w), sel.astype(np.int64), logits                                         # :849


def moe_block(x: np.ndarray, weights: dict, prefix: str, top_k: int = 2, emulate_bf16: bool = False):
    """MixtralSparseMoeBlock.forward (:839-882) with MixtralBLockSparseTop2MLP experts (:809-812): every token goes through its
    top-k experts, each output is scaled by the routing weight (rounded in the model dtype) and the expert outputs are
    accumulated in expert order by index_add_ (:880).  x [T,H] -> ([T,H], selected experts [T,k])."""
    rnd = bf16_round if emulate_bf16 else (lambda a: a.astype(F32))
    T, H = x.shape
    w, sel, _ = moe_router(x, weights[prefix + "gate.weight"], top_k, emulate_bf16)
    E = weights[prefix + "gate.weight"].shape[0]
    out = np.zeros((T, H), dtype=F32)
    for e in range(E):                                                                  # :859
        tok, slot = np.nonzero(sel == e)
        if tok.size == 0:
            continue
        xe = x[tok].astype(F32)
        g = rnd(xe @ weights[f"{prefix}experts.{e}.w1.weight"].T)
        u = rnd(xe @ weights[f"{prefix}experts.{e}.w3.weight"].T)
        y = rnd(rnd(rnd(silu(g)) * u) @ weights[f"{prefix}experts.{e}.w2.weight"].T)    # :810-811
        y = rnd(y * w[tok, slot][:, None])                                              # :876
        out[tok] = rnd(out[tok] + y)                                                    # :880 index_add_ (a token meets an expert once)
    return out, sel


# ----------------------------------------------------------------------------
# Pooling + normalise  (gritlm/gritlm.py)
# ----------------------------------------------------------------------------
def pooling(hidden: np.ndarray, attention_mask: np.ndarray, method: str) -> np.ndarray:
    """GritLM.pooling, gritlm/gritlm.py:178-218 (fp32 result, no recast).

    hidden [b, n, d]; attention_mask [b, n] integer.  NOTE the reference mutates
    the caller's mask in place for 'weightedmean' (:211); this restatement works
    on a copy and the host wrapper reproduces the mutation.
    """
    hidden = hidden.astype(F32)
    m = attention_mask.astype(np.int64).copy()
    b, n, d = hidden.shape
    if method == "cls":                                                   # :188
        return hidden[:, 0].copy()
    if method == "lasttoken":                                             # :190-208
        rev = m[:, ::-1]
        argmax_rev = np.argmax(rev, axis=1)
        idx = np.clip(n - argmax_rev - 1, 0, None)
        masked = hidden * m[..., None].astype(F32)
        return masked[np.arange(b), idx]
    if method in ("mean", "weightedmean"):                                # :209-214
        if method == "weightedmean":
            m = m * np.cumsum(m, axis=1)                                  # :211
        s = np.sum(hidden.astype(F64) * m[..., None].astype(F64), axis=1)
        den = m.sum(axis=1, keepdims=True).astype(F64)
        with np.errstate(divide="ignore", invalid="ignore"):
            return (s / den).astype(F32)                                  # unguarded, :213-214
    raise NotImplementedError(f"Unknown pooling method: {method}")       # :215


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """F.normalize(dim=-1), gritlm/gritlm.py:156-158, training/model.py:162-164."""
    n = np.sqrt(np.sum(x.astype(F64) ** 2, axis=-1, keepdims=True))
    return (x / np.maximum(n, eps)).astype(F32)


def instruction_mask(attention_mask: np.ndarray, instruction_lens) -> np.ndarray:
    """Pooling mask with the instruction tokens zeroed.

    gritlm/gritlm.py:144-153 (same length for every row) and
    gritlm/training/model.py:151-158 (per-row lengths).  Attention still sees
    the instruction; only the pool excludes it.
    """
    m = attention_mask.copy()
    if instruction_lens is None:
        return m
    if np.isscalar(instruction_lens):
        m[:, : int(instruction_lens)] = 0
        return m
    for i, l in enumerate(instruction_lens):
        m[i, : int(l)] = 0
    return m


def encode_core(weights, cfg, input_ids, attention_mask, pooling_method="mean", normalized=True,
                instruction_lens=None, emulate_bf16=False, acc_dtype=F64):
    """Device part of GritLM.encode (gritlm/gritlm.py:129-158) == GritLMTrainModel.encode
    (gritlm/training/model.py:134-165): forward, instruction masking, pool, normalise."""
    h = mistral_encode(weights, cfg, input_ids, attention_mask, emulate_bf16=emulate_bf16, acc_dtype=acc_dtype)
    pm = instruction_mask(attention_mask, instruction_lens)
    e = pooling(h, pm, pooling_method)
    return l2_normalize(e) if normalized else e


# ----------------------------------------------------------------------------
# Mixtral router auxiliary loss  (scripts/modeling_mixtral_gritlm.py)
# ----------------------------------------------------------------------------
def router_aux_loss(gate_logits: np.ndarray, attention_mask: np.ndarray, top_k: int = 2) -> float:
    """load_balancing_loss_func, scripts/modeling_mixtral_gritlm.py:80-153, with an attention mask (the training call, :1423-1428).

    gate_logits [L, B*S, E] (one [B*S, E] block per layer, concatenated by the reference :109-111), attention_mask [B, S].
    routing_weights = softmax; selected = top-k; f[slot, e] = share of the REAL tokens (all layers) whose slot-th choice is e (:131-133);
    P[e] = mean routing weight of e over the real tokens (:144-146); loss = E * sum_{slot, e} f[slot, e] * P[e] (:148-149)."""
    L, N, E = gate_logits.shape
    z = gate_logits.astype(F64).reshape(L * N, E)
    z = z - z.max(axis=1, keepdims=True)
    w = np.exp(z)
    w /= w.sum(axis=1, keepdims=True)
    order = np.argsort(-w, axis=1, kind="stable")[:, :top_k]                       # torch.topk: descending, first index on ties
    onehot = np.zeros((L * N, top_k, E), F64)
    np.put_along_axis(onehot, order[:, :, None], 1.0, axis=2)
    m = np.tile(attention_mask.reshape(-1).astype(F64), L)
    f = (onehot * m[:, None, None]).sum(axis=0) / m.sum()
    P = (w * m[:, None]).sum(axis=0) / m.sum()
    return float((f * P[None, :]).sum() * E)


# ----------------------------------------------------------------------------
# Contrastive loss  (gritlm/training/model.py)
# ----------------------------------------------------------------------------
def infonce(q: np.ndarray, p: np.ndarray, temperature: float):
    """DistributedContrastiveLoss.__call__ after the gather, gritlm/training/model.py:42-47.

    scores = q p^T / tau; target[i] = i * (Np // Nq); CrossEntropyLoss(mean).
    Returns (loss, dq, dp, scores) with dq/dp the exact gradients of the loss.
    """
    q64, p64 = q.astype(F64), p.astype(F64)
    nq, np_ = q.shape[0], p.shape[0]
    g = np_ // nq
    scores = q64 @ p64.T / temperature
    tgt = np.arange(nq) * g
    mx = scores.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(scores - mx).sum(axis=1))
    loss = float(np.mean(lse - scores[np.arange(nq), tgt]))
    soft = np.exp(scores - lse[:, None])
    soft[np.arange(nq), tgt] -= 1.0
    ds = soft / nq / temperature
    dq = ds @ p64
    dp = ds.T @ q64
    return loss, dq.astype(F32), dp.astype(F32), scores.astype(F32)


def gather_with_local(shards: list[np.ndarray], rank: int) -> np.ndarray:
    """_dist_gather_tensor, gritlm/training/model.py:49-60: all_gather then torch.cat in
    rank order; the local shard is re-inserted (it is the only one carrying grad)."""
    return np.concatenate(shards, axis=0)


def distributed_infonce(q_shards, p_shards, temperature, rank):
    """What ONE rank computes with negatives_cross_device=True
    (gritlm/training/model.py:36-47): the full global loss, gradients only for its
    own rows (other ranks' shards are constants).  DDP then averages weight grads."""
    q = gather_with_local(q_shards, rank); p = gather_with_local(p_shards, rank)
    loss, dq, dp, _ = infonce(q, p, temperature)
    bq, bp = q_shards[0].shape[0], p_shards[0].shape[0]
    return loss, dq[rank * bq:(rank + 1) * bq], dp[rank * bp:(rank + 1) * bp]


def pool_normalize_backward(hidden, pool_mask, method, normalized, grad_out):
    """Analytic backward of pooling(:188-214)+normalize(:156-158) w.r.t. hidden: every method is a per-row weighting of the
    positions (cls: position 0, :188-189; lasttoken: last position when left-padded else mask.sum-1, :190-207)."""
    hidden = hidden.astype(F64)
    m = pool_mask.astype(np.int64).copy()
    if method == "weightedmean":
        m = m * np.cumsum(m, axis=1)
    elif method == "cls":
        m = np.zeros_like(m); m[:, 0] = 1
    elif method == "lasttoken":
        left = bool(pool_mask[:, -1].sum() == pool_mask.shape[0])
        idx = np.full((m.shape[0],), m.shape[1] - 1) if left else pool_mask.sum(axis=1) - 1
        m = np.zeros_like(m); m[np.arange(m.shape[0]), idx] = 1
    den = m.sum(axis=1, keepdims=True).astype(F64)
    w = m / den
    pooled = np.einsum("bsd,bs->bd", hidden, w)
    g = grad_out.astype(F64)
    if normalized:
        n = np.maximum(np.sqrt((pooled ** 2).sum(-1, keepdims=True)), 1e-12)
        y = pooled / n
        g = (g - y * (y * g).sum(-1, keepdims=True)) / n
    return (w[..., None] * g[:, None, :]).astype(F32)


# ----------------------------------------------------------------------------
# Analytic backward restatements (fp64) -- what torch autograd computes for the
# reference ops; used to check the HIP backward kernels op by op.  The end-to-end
# pin is tests/golden/gradcache_tiny.npz (parameter gradients of the reference).
# ----------------------------------------------------------------------------
def rmsnorm_backward(dy, x, weight, eps):
    """d/dx, d/dw of MistralRMSNorm.forward (:84-89), roundings treated as identity."""
    x = x.astype(F64); dy = dy.astype(F64); w = weight.astype(F64)
    H = x.shape[-1]
    rs = 1.0 / np.sqrt(np.mean(x ** 2, axis=-1, keepdims=True) + eps)
    xhat = x * rs
    dxhat = dy * w
    dx = rs * (dxhat - xhat * np.mean(dxhat * xhat, axis=-1, keepdims=True))
    dw = np.sum(dy * xhat, axis=tuple(range(x.ndim - 1)))
    return dx.astype(F32), dw.astype(F32)


def swiglu_backward(g, u, dact):
    """d/dgate, d/dup of act_fn(gate) * up (:177-178, silu)."""
    g = g.astype(F64); u = u.astype(F64); d = dact.astype(F64)
    s = 1.0 / (1.0 + np.exp(-g))
    return (d * u * s * (1.0 + g * (1.0 - s))).astype(F32), (d * g * s).astype(F32)


def attention_bidirectional_backward(q, k, v, key_mask, dout, causal=False, window=0):
    """Backward of attention_bidirectional.  q [B,Hq,S,d], k,v [B,Hkv,S,d], dout [B,S,Hq*d]
    -> dq [B,Hq,S,d], dk, dv [B,Hkv,S,d] (GQA: kv gradients summed over the group)."""
    B, Hq, S, d = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    q64 = q.astype(F64)
    kk = np.repeat(k, rep, axis=1).astype(F64); vv = np.repeat(v, rep, axis=1).astype(F64)
    scale = 1.0 / np.sqrt(d)
    p = masked_softmax(np.matmul(q64, kk.transpose(0, 1, 3, 2)) * scale, key_mask, causal, window)
    do = dout.astype(F64).reshape(B, S, Hq, d).transpose(0, 2, 1, 3)
    dv_full = np.matmul(p.transpose(0, 1, 3, 2), do)
    dp = np.matmul(do, vv.transpose(0, 1, 3, 2))
    ds = p * (dp - np.sum(dp * p, axis=-1, keepdims=True))
    dq = np.matmul(ds, kk) * scale
    dk_full = np.matmul(ds.transpose(0, 1, 3, 2), q64) * scale
    dk = dk_full.reshape(B, Hkv, rep, S, d).sum(2)
    dv = dv_full.reshape(B, Hkv, rep, S, d).sum(2)
    return dq.astype(F32), dk.astype(F32), dv.astype(F32)
