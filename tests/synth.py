"""Deterministic synthetic models / inputs shared by the golden generator, the tests,
``__graft_entry__.smoke`` and ``bench.py``'s cpu_baseline leg.

Weights are drawn with numpy's ``default_rng`` (bit-stable across machines) and
rounded to bf16-representable fp32, so the reference (fp32 or bf16), the oracle and
the HIP path all see *identical* parameter values.
"""
from __future__ import annotations

import numpy as np

CONFIGS = {
    # head_dim is 128 everywhere: the shape of Mistral-7B (scripts/training/train_gritlm_7b.sh:54)
    "tiny": dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                 num_attention_heads=2, num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=10000.0),
    "gqa": dict(vocab_size=384, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0),
    # the true 7B LAYER shape (H 4096, I 14336, 32/8 heads), one layer, small vocabulary: parity fixture at the shape bench.py runs
    "7b-l1": dict(vocab_size=4096, hidden_size=4096, intermediate_size=14336, num_hidden_layers=1,
                  num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0),
    # 7B layer shape, EIGHT distinct layers, small vocabulary: the contrastive step at depth (tests/golden/train_7b-d8.npz, round 6)
    "7b-d8": dict(vocab_size=4096, hidden_size=4096, intermediate_size=14336, num_hidden_layers=8,
                  num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0),
    # 7B layer shape, two layers, small vocabulary: decode-path parity at the real GEMV / KV shapes
    "7b-l2s": dict(vocab_size=4096, hidden_size=4096, intermediate_size=14336, num_hidden_layers=2,
                   num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0),
    # the true 7B layer shape, 2 layers (CPU baseline workload)
    "7b-l2": dict(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=2,
                  num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0),
    "7b": dict(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
               num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=10000.0),
    # Mixtral family (scripts/modeling_mixtral_gritlm.py): sparse MoE MLP, 8 experts, top-2
    "moe-tiny": dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                     num_attention_heads=2, num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=1e6,
                     num_local_experts=8, num_experts_per_tok=2),
    "moe-gqa": dict(vocab_size=384, hidden_size=512, intermediate_size=768, num_hidden_layers=3,
                    num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=1e6,
                    num_local_experts=8, num_experts_per_tok=2),
    # ONE layer at the true Mixtral-8x7B layer shape (E 8, H 4096, I 14336, 32/8 heads), small vocabulary: the parity fixture at the shape
    # BASELINE configs[3] runs (5.6 GB of fp32 expert weights)
    "8x7b-l1": dict(vocab_size=4096, hidden_size=4096, intermediate_size=14336, num_hidden_layers=1,
                    num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=1e6,
                    num_local_experts=8, num_experts_per_tok=2),
    "8x7b": dict(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-5, rope_theta=1e6,
                 num_local_experts=8, num_experts_per_tok=2),
}


def _bf16_round(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def make_weights(cfg: dict, seed: int = 0, std: float = 0.02) -> dict:
    """HF ``MistralModel`` state_dict (fp32 numpy, bf16-representable values).

    Linear/embedding ~ N(0, std^2) (HF _init_weights, scripts/modeling_mistral_gritlm.py:819-828);
    RMSNorm weights 1 + 0.1 N(0,1) so a missing weight multiply cannot hide.
    """
    rng = np.random.default_rng(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    d = H // nh

    def lin(o, i):
        return _bf16_round(rng.standard_normal((o, i), dtype=np.float32) * std)

    def nrm():
        return _bf16_round(1.0 + 0.1 * rng.standard_normal(H, dtype=np.float32))

    w = {"embed_tokens.weight": lin(V, H)}
    for li in range(cfg["num_hidden_layers"]):
        p = f"layers.{li}."
        w[p + "self_attn.q_proj.weight"] = lin(nh * d, H)
        w[p + "self_attn.k_proj.weight"] = lin(nkv * d, H)
        w[p + "self_attn.v_proj.weight"] = lin(nkv * d, H)
        w[p + "self_attn.o_proj.weight"] = lin(H, nh * d)
        if "num_local_experts" in cfg:
            # router weights drawn wide (std 0.5): clear top-2 margins, so bf16 noise flips the routing of few tokens
            w[p + "block_sparse_moe.gate.weight"] = _bf16_round(rng.standard_normal((cfg["num_local_experts"], H), dtype=np.float32) * 0.5)
            for e in range(cfg["num_local_experts"]):
                w[p + f"block_sparse_moe.experts.{e}.w1.weight"] = lin(I, H)
                w[p + f"block_sparse_moe.experts.{e}.w2.weight"] = lin(H, I)
                w[p + f"block_sparse_moe.experts.{e}.w3.weight"] = lin(I, H)
        else:
            w[p + "mlp.gate_proj.weight"] = lin(I, H)
            w[p + "mlp.up_proj.weight"] = lin(I, H)
            w[p + "mlp.down_proj.weight"] = lin(H, I)
        w[p + "input_layernorm.weight"] = nrm()
        w[p + "post_attention_layernorm.weight"] = nrm()
    w["norm.weight"] = nrm()
    return w


def make_batch(cfg: dict, batch: int, seq: int, seed: int = 1234, min_len: int | None = None):
    """input_ids ~ U{3..V-1}, right-padded attention_mask (tokenizer padding_side='right',
    gritlm/gritlm.py:61).  ``min_len=None`` -> all rows full length."""
    rng = np.random.default_rng(seed)
    ids = rng.integers(3, cfg["vocab_size"], size=(batch, seq), dtype=np.int64)
    mask = np.ones((batch, seq), dtype=np.int64)
    if min_len is not None:
        lens = rng.integers(min_len, seq + 1, size=batch)
        lens[0] = seq
        for i, l in enumerate(lens):
            mask[i, l:] = 0
            ids[i, l:] = 0
    return ids, mask


def hf_config(cfg: dict):
    """transformers.MistralConfig / MixtralConfig for a synthetic config (tests that build HF model dirs)."""
    from transformers import MistralConfig, MixtralConfig
    cls, extra = MistralConfig, {}
    if "num_local_experts" in cfg:
        cls, extra = MixtralConfig, dict(num_local_experts=cfg["num_local_experts"], num_experts_per_tok=cfg["num_experts_per_tok"])
    c = cls(**extra, vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                      intermediate_size=cfg["intermediate_size"],
                      num_hidden_layers=cfg["num_hidden_layers"],
                      num_attention_heads=cfg["num_attention_heads"],
                      num_key_value_heads=cfg["num_key_value_heads"],
                      rms_norm_eps=cfg["rms_norm_eps"], max_position_embeddings=4096,
                      sliding_window=None, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                      tie_word_embeddings=False)
    c.rope_theta = cfg["rope_theta"]
    return c


# ---------------------------------------------------------------------------------------------
# Offline model directories (no network): WordLevel tokenizer + save_pretrained model
# ---------------------------------------------------------------------------------------------
WORDS = [f"w{i}" for i in range(400)]


def make_tokenizer(path: str):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}
    for w in WORDS:
        vocab[w] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])   # add_bos_token=True
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>")
    fast.save_pretrained(path)
    return fast


def make_sentences(n: int, seed: int = 5, min_words: int = 3, max_words: int = 100) -> list[str]:
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.integers(min_words, max_words + 1))
        out.append(" ".join(WORDS[int(j)] for j in rng.integers(0, len(WORDS), size=k)))
    return out


def build_mistral_dir(path: str, cfg_name="tiny", seed: int = 0, dtype="float32", weights: dict | None = None) -> str:
    """MistralForCausalLM with the synthetic weights of make_weights (or the given ``weights``) + tokenizer, saved to ``path``;
    ``cfg_name``: a CONFIGS key or a config dict."""
    import torch
    from transformers import MistralForCausalLM
    cfg = CONFIGS[cfg_name] if isinstance(cfg_name, str) else cfg_name
    hc = hf_config(cfg)
    model = MistralForCausalLM(hc)
    w = make_weights(cfg, seed) if weights is None else weights
    sd = {"model." + k: torch.from_numpy(v) for k, v in w.items()}
    rng = np.random.default_rng(seed + 99)
    sd["lm_head.weight"] = torch.from_numpy(_bf16_round(rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"]), dtype=np.float32) * 0.02))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    model = model.to(getattr(torch, dtype))
    model.save_pretrained(path)
    make_tokenizer(path)
    return path


def build_mixtral_dir(path: str, cfg_name: str = "moe-tiny", seed: int = 0, dtype="float32") -> str:
    """MixtralForCausalLM with the synthetic weights of make_weights (reference names block_sparse_moe.experts.N.w1/w2/w3 mapped to the
    installed transformers' fused gate_up_proj / down_proj parameters) + tokenizer, saved to ``path``."""
    import torch
    from transformers import MixtralForCausalLM
    cfg = CONFIGS[cfg_name]
    model = MixtralForCausalLM(hf_config(cfg))
    w = make_weights(cfg, seed)
    E = cfg["num_local_experts"]
    sd = {}
    for k, v in w.items():
        if "block_sparse_moe" not in k:
            sd["model." + k] = torch.from_numpy(v)
    for li in range(cfg["num_hidden_layers"]):
        p = f"layers.{li}.block_sparse_moe."
        sd[f"model.layers.{li}.mlp.gate.weight"] = torch.from_numpy(w[p + "gate.weight"])
        sd[f"model.layers.{li}.mlp.experts.gate_up_proj"] = torch.from_numpy(np.stack(
            [np.concatenate([w[f"{p}experts.{e}.w1.weight"], w[f"{p}experts.{e}.w3.weight"]], axis=0) for e in range(E)]))
        sd[f"model.layers.{li}.mlp.experts.down_proj"] = torch.from_numpy(np.stack([w[f"{p}experts.{e}.w2.weight"] for e in range(E)]))
    rng = np.random.default_rng(seed + 99)
    sd["lm_head.weight"] = torch.from_numpy(_bf16_round(rng.standard_normal((cfg["vocab_size"], cfg["hidden_size"]), dtype=np.float32) * 0.02))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
    model = model.to(getattr(torch, dtype))
    model.save_pretrained(path)
    make_tokenizer(path)
    return path


def build_gptneo_dir(path: str, seed: int = 0) -> str:
    """Tiny GPT-Neo (architecture of SGPT-125M-weightedmean, README.md:38; BASELINE.json configs[0] plumbing case)."""
    import torch
    from transformers import GPTNeoConfig, GPTNeoForCausalLM
    hc = GPTNeoConfig(vocab_size=len(WORDS) + 4, hidden_size=64, num_layers=2, num_heads=4, intermediate_size=128,
                      attention_types=[[["global", "local"], 1]], max_position_embeddings=256, window_size=64,
                      bos_token_id=1, eos_token_id=2, pad_token_id=0)
    model = GPTNeoForCausalLM(hc)
    rng = np.random.default_rng(seed)
    sd = model.state_dict()
    new = {}
    for k in sorted(sd.keys()):
        t = sd[k]
        if not t.dtype.is_floating_point or "masked_bias" in k or k.endswith(".attn.attention.bias"):
            continue
        if "ln_" in k and k.endswith("weight"):
            v = 1.0 + 0.1 * rng.standard_normal(tuple(t.shape), dtype=np.float32)
        elif k.endswith("bias"):
            v = 0.02 * rng.standard_normal(tuple(t.shape), dtype=np.float32)
        else:
            v = 0.05 * rng.standard_normal(tuple(t.shape), dtype=np.float32)
        new[k] = torch.from_numpy(v)
    model.load_state_dict(new, strict=False)
    model.save_pretrained(path)
    make_tokenizer(path)
    return path
