#!/usr/bin/env python3
"""BASELINE configs[4] "RAG doc-caching": encode passages with get_cache=True (native bidirectional pass that also emits the per-layer
post-RoPE K/V, gritlm/gritlm.py:131-140, rag/eval.py:124-150) and generate tokens for a query REUSING a passage's KV (rag/eval.py:237-246,
cache == "doc").  Encoding + KV emission run on the HIP engine; the token-by-token decode is timed twice: the Hugging Face module's
generate() on the spliced cache (what the reference does) and the native decoder (gritlm_amd/decoder.py, csrc/decode.hip).   python tools/rag_cache_bench.py [--passages 512 --seq 2048 --new-tokens 128] [--tiny --cpu]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from gritlm_amd import GritLM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passages", type=int, default=512)
ap.add_argument("--seq", type=int, default=2048)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--new-tokens", type=int, default=128)
ap.add_argument("--queries", type=int, default=4)
ap.add_argument("--tiny", action="store_true", help="tiny synthetic model (logic check)")
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
dev = "cpu" if a.cpu else "cuda"
dtype = torch.float32 if a.cpu else torch.bfloat16

import tempfile  # noqa: E402
from transformers import AutoTokenizer, MistralConfig, MistralForCausalLM  # noqa: E402

td = tempfile.mkdtemp()
synth.make_tokenizer(td)
tok = AutoTokenizer.from_pretrained(td, padding_side="right")
if a.tiny:
    hc = synth.hf_config(synth.CONFIGS["tiny"])
else:
    hc = MistralConfig(vocab_size=32000, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                       num_key_value_heads=8, rms_norm_eps=1e-5, max_position_embeddings=32768, sliding_window=None, pad_token_id=0,
                       bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    hc.rope_theta = 10000.0
t0 = time.perf_counter()
# build directly in the target dtype: a later .to(bfloat16) would also round the rotary inv_freq BUFFER to bf16 (from_pretrained keeps it
# fp32), and positions ~2000 then rotate by visibly wrong angles
torch.set_default_dtype(dtype)
with torch.device(dev):
    lm = MistralForCausalLM(hc)
torch.set_default_dtype(torch.float32)
lm = lm.eval()
assert lm.model.rotary_emb.inv_freq.dtype == torch.float32
m = GritLM.__new__(GritLM)
torch.nn.Module.__init__(m)
m.model, m.tokenizer, m.device, m.embedding_attr, m.projection = lm, tok, dev, "model", None
m.pooling_method, m.normalized, m.attn, m.embed_eos, m.num_gpus, m.engine, m._native = "mean", True, "bbcc", "", 1, None, None
m.engines, m._precision = [], "bf16"
m.generate = lm.generate
m._maybe_build_engine()
t_init = time.perf_counter() - t0

passages = synth.make_sentences(a.passages, seed=11, min_words=a.seq, max_words=a.seq + 50)      # truncated to --seq tokens
if dev == "cuda":
    torch.cuda.synchronize()
t0 = time.perf_counter()
emb, caches = [], []
for s in range(0, a.passages, a.batch):
    e, c = m.encode(passages[s:s + a.batch], batch_size=a.batch, max_length=a.seq, get_cache=True, convert_to_tensor=True,
                    add_special_tokens=False)
    emb.append(e); caches.append(c)
if dev == "cuda":
    torch.cuda.synchronize()
t_enc = time.perf_counter() - t0

# per-layer (k, v) of one passage -> the installed transformers' cache object
from transformers import DynamicCache  # noqa: E402


def passage_cache(i):
    c = caches[i // a.batch]
    j = i % a.batch
    out = DynamicCache()
    layers = c.layers if hasattr(c, "layers") else None
    for li in range(hc.num_hidden_layers):
        k, v = (layers[li].keys, layers[li].values) if layers is not None else (c[li][0], c[li][1])
        out.update(k[j:j + 1].clone(), v[j:j + 1].clone(), li)
    return out


q_ids = tok(synth.make_sentences(a.queries, seed=12, min_words=20, max_words=20), return_tensors="pt", add_special_tokens=False)["input_ids"].to(dev)
if dev == "cuda":
    torch.cuda.synchronize()
t0 = time.perf_counter()
n_new = 0
for qi in range(a.queries):
    pc = passage_cache(qi)
    plen = pc.get_seq_length()
    ids = q_ids[qi:qi + 1]
    mask = torch.ones((1, plen + ids.shape[1]), dtype=torch.long, device=dev)
    # the prompt = [cached passage tokens (placeholders, never re-encoded)] + query tokens, as rag/eval.py hands `past_key_values` to generate
    full = torch.cat([torch.zeros((1, plen), dtype=torch.long, device=dev), ids], dim=1)
    out = lm.generate(input_ids=full, attention_mask=mask, past_key_values=pc, max_new_tokens=a.new_tokens, min_new_tokens=a.new_tokens,
                      do_sample=False, pad_token_id=0)
    n_new += out.shape[1] - full.shape[1]
    hf_tokens = out[0, full.shape[1]:] if qi == 0 else hf_tokens
if dev == "cuda":
    torch.cuda.synchronize()
t_gen = time.perf_counter() - t0

# the same generation on the native decoder (GEMV / decode-attention kernels, one HIP graph per step)
native = None
if m.engine is not None:
    from gritlm_amd.decoder import MistralDecoder
    dec = MistralDecoder(m.engine, lm.lm_head.weight)

    def slices(i):
        c = caches[i // a.batch]
        j = i % a.batch
        layers = c.layers if hasattr(c, "layers") else None
        return [((layers[li].keys if layers is not None else c[li][0])[j:j + 1], (layers[li].values if layers is not None else c[li][1])[j:j + 1])
                for li in range(hc.num_hidden_layers)]

    def decode_ms_per_token(kv):
        """Decode-only rate: the slope between a run of n and a run of 4 n new tokens on the same prefix (prompt, cache copy and launch
        set-up cancel); the MEDIAN of three such pairs -- one host-side hiccup in one run (seen: +35 ms) otherwise moves the figure by 3 %."""
        slopes = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.generate(q_ids[:1], a.new_tokens, past_key_values=kv)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            dec.generate(q_ids[:1], 4 * a.new_tokens, past_key_values=kv)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            slopes.append(((t2 - t1) - (t1 - t0)) / (3 * a.new_tokens) * 1e3)
        return sorted(slopes)[1], slopes

    dec.generate(q_ids[:1], 4, past_key_values=slices(0))            # warm-up (allocations, kernel load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for qi in range(a.queries):
        toks = dec.generate(q_ids[qi:qi + 1], a.new_tokens, past_key_values=slices(qi))
        if qi == 0:
            first = toks[0]
    torch.cuda.synchronize()
    t_nat = time.perf_counter() - t0
    ms_bf16, ms_bf16_runs = decode_ms_per_token(slices(0))
    # the reference's latency setting (rag/eval.py --latency: max_new_tokens 16, one query on one cached document): time per query, where
    # the prompt tokens on top of the cache weigh as much as the 16 decode steps -- native (prompt chunk + graph replay) and Hugging Face
    lat = {"new_tokens": 16, "what": "seconds per query at the reference's latency setting (16 new tokens on one cached 2048-token passage), median of 5"}
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.generate(q_ids[:1], 16, past_key_values=slices(0))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lat["native_s"] = sorted(ts)[2]
    dec.prompt_chunk = False
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.generate(q_ids[:1], 16, past_key_values=slices(0))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lat["native_token_by_token_prompt_s"] = sorted(ts)[2]
    dec.prompt_chunk = True
    ts = []
    for _ in range(3):
        pc = passage_cache(0)
        full = torch.cat([torch.zeros((1, pc.get_seq_length()), dtype=torch.long, device=dev), q_ids[:1]], dim=1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lm.generate(input_ids=full, attention_mask=torch.ones_like(full), past_key_values=pc, max_new_tokens=16, min_new_tokens=16, do_sample=False, pad_token_id=0)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    lat["hugging_face_generate_s"] = sorted(ts)[1]
    lat["query_tokens"] = int(q_ids.shape[1])
    # PARITY of the decode path (VERDICT r04 #2c): TEACHER-FORCED next-token logits of the native decoder against the reference-equivalent
    # module IN FP32 (the same weights widened, exact) on the SAME cached passage KV and the same query prefix, at several prefix lengths.
    # Random-init weights give nearly flat logits, so greedy-token identity says nothing (a 2 % match is the expected outcome of two
    # correct bf16 implementations); the centred logit VECTORS do: 1 - cos and relative l2 against fp32, numeric bounds.  The stock
    # bf16 module through PyTorch-ROCm (what the reference's generate() computes on this GPU) is measured beside it.
    import copy
    lm32 = copy.deepcopy(lm).float()
    ks = [k for k in (1, 7, q_ids.shape[1]) if k <= q_ids.shape[1]]
    with torch.no_grad():
        def module_logits(mod, dt, kv=None):
            pc = DynamicCache()
            for li, (k_, v_) in enumerate(slices(0) if kv is None else kv):
                pc.update(k_.to(dt).clone(), v_.to(dt).clone(), li)
            plen = pc.get_seq_length()
            return mod(input_ids=q_ids[:1], past_key_values=pc, attention_mask=torch.ones((1, plen + q_ids.shape[1]), dtype=torch.long, device=dev),
                       position_ids=torch.arange(plen, plen + q_ids.shape[1], device=dev).unsqueeze(0)).logits[0].float()
        ref_logits = module_logits(lm32, torch.float32)              # [query tokens, V]: position k-1 = next-token logits after k tokens
        hf_logits_all = module_logits(lm, dtype)

    def cmp(x, r):
        x, r = x.double() - x.double().mean(), r.double() - r.double().mean()
        top = lambda t: set(torch.topk(t, 10).indices.tolist())
        return {"one_minus_cos": float(1 - torch.nn.functional.cosine_similarity(x, r, dim=0)), "rel_l2": float((x - r).norm() / r.norm()),
                "top10_overlap": len(top(x) & top(r)) / 10.0, "argmax_equal": bool(x.argmax() == r.argmax())}
    # the ENCODE half of the flow at the north-star tolerance (round 6): encode(get_cache=True) under each precision policy on two of the
    # passages, embeddings against the reference-equivalent module in FP32 (same weights, same token ids)
    enc_parity = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_reference as TR
        sub = passages[:2]
        tk = tok(sub, padding=True, truncation=True, return_tensors="pt", max_length=a.seq, add_special_tokens=False)
        e32 = TR.encode(lm32.model, tk["input_ids"].to(dev), tk["attention_mask"].to(dev)).double()
        enc_parity = {"what": f"encode(get_cache=True) of 2 passages x {a.seq} tokens under each precision policy: pooled embeddings vs the "
                              "reference-equivalent module in FP32 on the same weights; max 1 - cos (north-star: 1e-4)"}
        f16_flow = {"what": "the whole flow under an fp16 policy: encode(get_cache=True) with native_kv_cache (fp16 K/V as the attention read them) "
                            "-> native decoder on fp16 operands (fp32 stream, grit_gemv_f16 / grit_attn_decode_rope_f16).  decode parity: "
                            "teacher-forced next-token logits after k query tokens vs the reference-equivalent module in FP32 on the same cached "
                            "K/V (widened, exact) and the same prefix -- the definition of the bf16 leg above"}
        for pol in ("bf16", "f16_stream", "f16_operands"):
            m.set_precision(pol)
            m.native_kv_cache = pol != "bf16"
            e_, c_ = m.encode(sub, batch_size=2, max_length=a.seq, get_cache=True, convert_to_tensor=True, add_special_tokens=False)
            l0 = c_.layers[0] if hasattr(c_, "layers") else None
            enc_parity[pol] = {"max_one_minus_cos": float((1 - torch.nn.functional.cosine_similarity(e_.double(), e32, dim=1)).max()),
                               "cache_dtype": str((l0.keys if l0 is not None else c_[0][0]).dtype)}
            if pol != "bf16":
                kv16 = [((l.keys, l.values) if hasattr(l, "keys") else (l[0], l[1])) for l in (c_.layers if hasattr(c_, "layers") else c_)]
                kv16 = [(k_[:1], v_[:1]) for k_, v_ in kv16]
                with torch.no_grad():
                    ref16 = module_logits(lm32, torch.float32, kv16)
                pk = {}
                for k in ks:
                    _, nl = dec.generate(q_ids[:1, :k], 1, past_key_values=kv16, return_logits=True)
                    pk[str(k)] = cmp(nl[0, 0].float(), ref16[k - 1])
                assert dec.last_precision == "f16", dec.last_precision
                dec.generate(q_ids[:1], 4, past_key_values=kv16)
                ms, ms_runs = decode_ms_per_token(kv16)
                f16_flow[pol] = {"decode_native_vs_fp32": pk, "max_one_minus_cos": max(v["one_minus_cos"] for v in pk.values()),
                                 "max_rel_l2": max(v["rel_l2"] for v in pk.values()), "decode_ms_per_token": ms, "decode_ms_per_token_runs": ms_runs,
                                 "decode_frac_of_weight_streaming_roofline": sum(p.numel() for p in lm.parameters()) * 2 / 8e12 * 1e3 / ms,
                                 "encode_max_one_minus_cos": enc_parity[pol]["max_one_minus_cos"], "kv_cache_dtype": enc_parity[pol]["cache_dtype"]}
                del kv16
            del e_, c_
        enc_parity["north_star_met"] = bool(min(enc_parity["f16_stream"]["max_one_minus_cos"], enc_parity["f16_operands"]["max_one_minus_cos"]) < 1e-4)
        f16_flow["north_star_met"] = bool(all(max(f16_flow[p_]["max_one_minus_cos"], f16_flow[p_]["encode_max_one_minus_cos"]) < 1e-4
                                              for p_ in ("f16_stream", "f16_operands")))
    except Exception as ex:  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        enc_parity = {"error": repr(ex)[:300]}
        f16_flow = {"error": repr(ex)[:300]}
    finally:
        m.set_precision("bf16")
        m.native_kv_cache = False
    del lm32
    torch.cuda.empty_cache()
    per_k, per_k_hf = {}, {}
    for k in ks:
        _, nl = dec.generate(q_ids[:1, :k], 1, past_key_values=slices(0), return_logits=True)
        per_k[str(k)] = cmp(nl[0, 0].float(), ref_logits[k - 1])
        per_k_hf[str(k)] = cmp(hf_logits_all[k - 1], ref_logits[k - 1])
    BOUND_COS, BOUND_L2 = 2.0e-3, 7.0e-2            # bf16 arithmetic over 32 layers: the encoder measures 7e-4 of 1 - cos on this model family
    worst_cos = max(v["one_minus_cos"] for v in per_k.values()); worst_l2 = max(v["rel_l2"] for v in per_k.values())
    parity = {"what": "teacher-forced next-token logits (centred) after k query tokens on top of one cached 2048-token passage: native decoder "
                      "(bf16, csrc/decode.hip) vs the reference-equivalent Hugging Face module in FP32 on the same weights, the same cached KV "
                      "and the same prefix", "prefix_lengths": ks, "native_vs_fp32": per_k, "stock_bf16_module_vs_fp32": per_k_hf,
              "max_one_minus_cos": worst_cos, "max_rel_l2": worst_l2, "bound_one_minus_cos": BOUND_COS, "bound_rel_l2": BOUND_L2,
              "logits_std": float(ref_logits[-1].std()), "within_bound": bool(worst_cos < BOUND_COS and worst_l2 < BOUND_L2),
              "level": f"bf16 arithmetic (the reference's decode dtype, the default policy): the decode path's logits sit {worst_cos / 1e-4:.0f}x above "
                       "the 1e-4 the north-star states for encode() -- the level of the stock bf16 module on the same cache "
                       "(stock_bf16_module_vs_fp32).  Under the fp16 policies BOTH halves of the flow meet 1e-4: f16_flow",
              "encode_get_cache_by_policy": enc_parity, "f16_flow": f16_flow}
    native = {"generate_s_per_query": t_nat / a.queries, "latency_16_new_tokens": lat, "parity": parity,
              "tokens_per_s_incl_prompt_and_cache_copy": a.new_tokens * a.queries / t_nat,
              "decode_ms_per_token": ms_bf16, "decode_ms_per_token_runs": ms_bf16_runs,
              "hbm_roofline_ms_per_token": sum(p.numel() for p in lm.parameters()) * 2 / 8e12 * 1e3}
tokens = a.passages * a.seq
kv_gb = sum(sum(x.numel() * x.element_size() for x in ((l.keys, l.values) if hasattr(l, "keys") else l)) for c in caches
            for l in (c.layers if hasattr(c, "layers") else c)) / 1e9
enc_frac = (tokens / t_enc) * m.engine.flops_per_token(a.seq) / 2.5e15 if m.engine is not None else None
print(json.dumps({"metric": "RAG doc-caching: encode passages (+KV) and generate from the cached KV", "passages": a.passages, "seq": a.seq,
                  "parity": native["parity"] if native else None,
                  "encode_mfma_roofline_frac": enc_frac,
                  "decode_frac_of_weight_streaming_roofline": (native["hbm_roofline_ms_per_token"] / native["decode_ms_per_token"]) if native else None,
                  "encode_s": t_enc, "passages_per_s": a.passages / t_enc, "encode_tokens_per_s": tokens / t_enc, "kv_cache_gb": kv_gb,
                  "native_engine": m.engine is not None, "generate_s_per_query": t_gen / a.queries, "new_tokens_per_query": n_new / a.queries,
                  "decode_tokens_per_s": n_new / t_gen, "decode_path": "Hugging Face generate() on the spliced cache", "native_decode": native,
                  "model_init_s": t_init, "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9 if dev == "cuda" else None}))
