"""Pin the CPU oracle (oracle/gritlm_oracle.py) against fixtures produced by the reference's own
Python (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import gritlm_oracle as O
import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("method", ["mean", "weightedmean", "cls", "lasttoken"])
def test_pooling_matches_reference(golden_dir, method):
    g = _load(golden_dir, "pooling.npz")
    out = O.pooling(g["hidden"], g["mask"], method)
    np.testing.assert_allclose(out, g[f"pool_{method}"], rtol=1e-5, atol=1e-6)
    # bf16 hidden in, fp32 accumulate (gritlm/gritlm.py:212-214)
    outb = O.pooling(O.bf16_round(g["hidden"]), g["mask"], method)
    np.testing.assert_allclose(outb, g[f"pool_{method}_bf16in"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.bf16_round(outb), g[f"pool_{method}_recast"], rtol=0, atol=1e-2)


def test_pooling_unknown_method_raises(golden_dir):
    g = _load(golden_dir, "pooling.npz")
    with pytest.raises(NotImplementedError):
        O.pooling(g["hidden"], g["mask"], "weighted_mean")   # README's own typo, README.md:38


def test_weightedmean_mask_mutation_recorded(golden_dir):
    g = _load(golden_dir, "pooling.npz")
    m = g["mask"]
    np.testing.assert_array_equal(g["mask_after_weightedmean"], m * np.cumsum(m, axis=1))
    np.testing.assert_array_equal(g["mask_after_mean"], m)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_infonce_matches_reference(golden_dir, tag):
    g = _load(golden_dir, "infonce.npz")
    loss, dq, dp, _ = O.infonce(g[f"{tag}_q"], g[f"{tag}_p"], float(g[f"{tag}_tau"]))
    assert abs(loss - float(g[f"{tag}_loss"])) < 1e-4 * max(1.0, abs(loss))
    np.testing.assert_allclose(dq, g[f"{tag}_dq"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dp, g[f"{tag}_dp"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("fixture,suffix", [("train_7b-l1.npz", ""), ("train_moe-tiny.npz", "_f32")])
def test_training_fixtures_loss_is_the_infonce_of_their_reps(golden_dir, fixture, suffix):
    """The training-step fixtures (reference GritLMTrainModel.forward at the 7B layer shape / around its Mixtral) carry reps and loss:
    the oracle's InfoNCE on those reps must give that loss -- two more reference-generated vectors for the loss, and a consistency
    check of the fixtures the GPU training checks read."""
    g = _load(golden_dir, fixture)
    loss, _, _, _ = O.infonce(g["q_reps" + suffix], g["p_reps" + suffix], float(g["tau"]))
    ref = float(g["loss" + suffix])
    assert abs(loss - ref) < 2e-4 * max(1.0, abs(ref)), (loss, ref)
    assert np.allclose(np.linalg.norm(g["q_reps" + suffix], axis=1), 1.0, atol=1e-5)


def test_router_aux_loss_matches_reference(golden_dir):
    """oracle.router_aux_loss (Mixtral's load_balancing_loss_func) on the reference's own router logits == the reference's aux_loss."""
    g = _load(golden_dir, "generative_moe-tiny.npz")
    aux = O.router_aux_loss(g["router_logits"], g["attention_mask"])
    assert abs(aux - float(g["aux_loss"])) < 1e-5 * float(g["aux_loss"]), (aux, float(g["aux_loss"]))
    # and the loss decomposes as the reference says: loss = loss_noaux + coef * aux
    assert abs(float(g["loss"]) - (float(g["loss_noaux"]) + float(g["router_aux_loss_coef"]) * float(g["aux_loss"]))) < 1e-4 * float(g["loss"])


def test_distributed_infonce_matches_reference_gloo_run(golden_dir):
    g = _load(golden_dir, "infonce_dist2.npz")
    world = int(g["world"]); q, p, tau = g["q"], g["p"], float(g["tau"])
    bq, bp = q.shape[0] // world, p.shape[0] // world
    qs = [q[r * bq:(r + 1) * bq] for r in range(world)]
    ps = [p[r * bp:(r + 1) * bp] for r in range(world)]
    for r in range(world):
        loss, dq, dp = O.distributed_infonce(qs, ps, tau, r)
        assert abs(loss - float(g[f"loss_rank{r}"])) < 1e-4
        np.testing.assert_allclose(dq, g[f"dq_rank{r}"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(dp, g[f"dp_rank{r}"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("cfg_name", ["tiny", "gqa"])
def test_torch_reference_equals_reference_fixture(golden_dir, cfg_name):
    """oracle/torch_reference.py (stock transformers.MistralModel + explicit 4-D bidirectional mask; bench.py's cpu / rocm-torch
    baselines) reproduces the reference's MistralModel(is_causal=False) outputs and the reference's pooled embeddings."""
    import torch
    import torch_reference as TR
    g = _load(golden_dir, f"encoder_{cfg_name}.npz")
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    model = TR.build_model(cfg, torch.float32, "cpu", {k: torch.from_numpy(v) for k, v in w.items()})
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    h = TR.hidden_states(model, ids, mask).numpy()
    valid = g["attention_mask"].astype(bool)
    assert np.abs(h - g["last_hidden_state"])[valid].max() < 1e-5
    e = TR.encode(model, ids, mask).numpy()
    assert np.all(1 - np.sum(e * g["emb_mean"], axis=1) < 1e-6)


def _depth32_case(g, tag):
    import bench
    ids, mask = g[f"{tag}_input_ids"], g[f"{tag}_attention_mask"]
    cfg, w, i2, _ = bench.oracle_full_depth_case(sample_docs=ids.shape[0], seq=ids.shape[1], layers=int(g["layers"]))
    assert np.array_equal(i2, ids)                                         # generator determinism: the fixture's model is regenerated here
    return cfg, w, ids, mask


def test_full_depth_oracle_matches_reference(golden_dir):
    """ALL 32 layers at the 7B layer shape: the numpy oracle against the REFERENCE's own fp32 run (tests/golden/encoder_7b-depth32.npz,
    `short` case: 1 doc x 64 tokens; 0.9 TFLOP of fp32 BLAS) -- the full depth is pinned on the reference, not on the oracle."""
    g = _load(golden_dir, "encoder_7b-depth32.npz")
    cfg, w, ids, mask = _depth32_case(g, "short")
    h = O.mistral_encode(w, cfg, ids, mask, acc_dtype=np.float32)
    hp, ref = h.reshape(-1, h.shape[-1])[g["short_probe_rows"]], g["short_probe_hidden"]
    assert np.linalg.norm(hp - ref) / np.linalg.norm(ref) < 2e-5, np.linalg.norm(hp - ref) / np.linalg.norm(ref)
    e = O.l2_normalize(O.pooling(h, mask, "mean"))
    assert np.all(1 - np.sum(e * g["short_emb"], axis=1) < 1e-6)
    # what bf16 costs the REFERENCE at this depth (its own bf16 run against its own fp32 run): the yardstick the GPU checks quote
    for tag in ("short", "ragged", "full"):
        d = 1 - np.sum(g[f"{tag}_emb"] * g[f"{tag}_emb_bf16"], axis=1)
        assert np.all(d < 2e-3) and np.all(d > 1e-5), d


def test_torch_reference_full_depth_and_mask_rule(golden_dir):
    """oracle/torch_reference.py at full depth, BOTH mask paths of the reference (:1017-1020): the all-valid `short` case takes the
    mask-is-None path, and must equal the reference's fp32 run; with the explicit 4-D mask the stock module gives the same answer on the
    host (fp32: the two SDPA paths agree to rounding)."""
    import torch
    import torch_reference as TR
    g = _load(golden_dir, "encoder_7b-depth32.npz")
    cfg, w, ids, mask = _depth32_case(g, "short")
    assert TR.reference_mask(torch.from_numpy(mask), torch.float32) is None
    padded = mask.copy(); padded[0, -1] = 0
    assert TR.reference_mask(torch.from_numpy(padded), torch.float32) is not None
    sd = {k: torch.from_numpy(v) for k, v in w.items() if not k.startswith("layers.") or k.startswith("layers.0.")}
    model = TR.build_model(dict(cfg, num_hidden_layers=1), torch.float32, "cpu", sd)
    model.layers = torch.nn.ModuleList([model.layers[0]] * int(g["layers"]))      # the fixture's model: one layer's arrays, 32 times
    tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    for rule in ("reference", "explicit"):
        e = TR.encode(model, tid, tm, mask_rule=rule).numpy()
        assert np.all(1 - np.sum(e * g["short_emb"], axis=1) < 1e-6), rule
    h1 = TR.hidden_states(model, tid, tm, layers=1)
    assert not torch.allclose(h1, TR.hidden_states(model, tid, tm)), "`layers` must truncate the stack"


def test_torch_reference_stack_equals_stock_forward():
    """The restated layer loop of torch_reference.hidden_states == the stock MistralModel.forward, bit for bit, when both get the explicit mask."""
    import torch
    import torch_reference as TR
    cfg = synth.CONFIGS["gqa"]
    model = TR.build_model(cfg, torch.float32, "cpu")
    ids, mask = synth.make_batch(cfg, 3, 72, 5, 20)
    tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    with torch.no_grad():
        stock = model(input_ids=tid, attention_mask=TR.bidirectional_mask(tm, model.dtype))[0]
    assert torch.equal(stock, TR.hidden_states(model, tid, tm, "explicit"))
    assert torch.equal(stock, TR.hidden_states(model, tid, tm, "reference"))        # padded batch: the reference builds the same mask


def test_encoder_7b_layer_shape_matches_reference(golden_dir):
    """Oracle vs the reference at the TRUE 7B layer shape (H 4096, I 14336, 32/8 heads, one layer, 2 x 512 ragged tokens):
    full-K (4096 / 14336) accumulation, 64 probe rows of last_hidden_state + pooled embeddings."""
    g = _load(golden_dir, "encoder_7b-l1.npz")
    cfg = synth.CONFIGS["7b-l1"]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    i2, m2 = synth.make_batch(cfg, ids.shape[0], ids.shape[1], 777, 200)
    assert np.array_equal(i2, ids) and np.array_equal(m2, mask)            # generator determinism
    h = O.mistral_encode(w, cfg, ids, mask, acc_dtype=np.float32)          # fp32 BLAS: 0.45 TFLOP, seconds
    hp = h.reshape(-1, h.shape[-1])[g["probe_rows"]]
    ref = g["probe_hidden"]
    assert np.abs(hp - ref).max() < 2e-3 * np.abs(ref).max(), np.abs(hp - ref).max()
    assert np.linalg.norm(hp - ref) / np.linalg.norm(ref) < 1e-5
    for method in ("mean", "weightedmean"):
        e = O.l2_normalize(O.pooling(h, mask, method))
        assert np.all(1 - np.sum(e * g[f"emb_{method}"], axis=1) < 1e-6)


@pytest.mark.parametrize("cfg_name", ["tiny", "gqa"])
def test_encoder_matches_reference(golden_dir, cfg_name):
    g = _load(golden_dir, f"encoder_{cfg_name}.npz")
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    # regenerated inputs are the fixture's inputs (generator determinism)
    h = O.mistral_encode(w, cfg, ids, mask)
    ref = g["last_hidden_state"]
    valid = mask.astype(bool)
    err = np.abs(h - ref)[valid].max()
    assert err < 2e-4, err
    # padded query rows are computed too (mask is on keys only)
    assert np.abs(h - ref)[~valid].max() < 2e-4 if (~valid).any() else True
    for method in ("mean", "weightedmean", "cls", "lasttoken"):
        e = O.l2_normalize(O.pooling(h, mask, method))
        cos = np.sum(e * g[f"emb_{method}"], axis=1)
        assert np.all(1 - cos < 1e-6), (method, cos)
    for method in ("mean", "weightedmean"):
        e = O.encode_core(w, cfg, ids, mask, method, True, g["instruction_lens"])
        np.testing.assert_allclose(e, g[f"train_reps_{method}"], atol=2e-5)


def test_encoder_bf16_emulation_tracks_bf16_reference(golden_dir):
    g = _load(golden_dir, "encoder_tiny.npz")
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    h = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], emulate_bf16=True)
    ref_b, ref_f = g["last_hidden_state_bf16"], g["last_hidden_state"]
    valid = g["attention_mask"].astype(bool)
    rel = lambda a, b: np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid])
    # emulation is as close to the bf16 reference as the bf16 reference is to fp32
    assert rel(h, ref_b) < 1.5 * rel(ref_b, ref_f) + 1e-3, (rel(h, ref_b), rel(ref_b, ref_f))


@pytest.mark.parametrize("cfg_name", ["moe-tiny", "moe-gqa"])
def test_mixtral_encoder_matches_reference(golden_dir, cfg_name):
    """Sparse-MoE (Mixtral) bidirectional encode of the oracle vs scripts/modeling_mixtral_gritlm.py (fp32): hidden states,
    the routing decisions of every layer, pooled embeddings; and the bf16 emulation tracks the bf16 reference."""
    g = _load(golden_dir, f"encoder_{cfg_name}.npz")
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    ids, mask = g["input_ids"], g["attention_mask"]
    h, routing = O.mistral_encode(w, cfg, ids, mask, return_layers="routing")
    valid = mask.astype(bool)
    assert np.abs(h - g["last_hidden_state"])[valid].max() < 3e-4
    got = np.sort(np.stack(routing), axis=-1)[:, valid]
    want = np.sort(g["routing"], axis=-1)[:, valid]
    assert (got == want).all(-1).mean() > 0.999          # fp32 vs fp32: only exact near-ties may differ
    for method in ("mean", "weightedmean"):
        e = O.l2_normalize(O.pooling(h, mask, method))
        assert np.all(1 - np.sum(e * g[f"emb_{method}"], axis=1) < 1e-6)
    hb = O.mistral_encode(w, cfg, ids, mask, emulate_bf16=True)
    ref_b, ref_f = g["last_hidden_state_bf16"], g["last_hidden_state"]
    rel = lambda a, b: np.linalg.norm((a - b)[valid]) / np.linalg.norm(b[valid])
    assert rel(hb, ref_b) < 1.5 * rel(ref_b, ref_f) + 1e-3, (rel(hb, ref_b), rel(ref_b, ref_f))


def test_generative_branch_matches_reference(golden_dir):
    """Causal encode + lm_head + NextTokenLoss of the oracle vs the reference's MistralForCausalLM / GritLMTrainModel(generative=...)."""
    g = _load(golden_dir, "generative_tiny.npz")
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, 0)
    h = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], causal=True)
    logits = h @ g["lm_head"].T
    valid = g["attention_mask"].astype(bool)
    assert np.abs(logits - g["logits"])[valid].max() < 3e-4
    hb = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], causal=False)
    assert np.abs(hb - h)[valid].max() > 1e-2                      # the causal flag does something
    for kind in ("mixed", "token"):
        got = O.next_token_loss(logits, g["labels"], kind, float(g[f"factor_{kind}"]))
        assert abs(got - float(g[f"loss_gen_{kind}"])) < 2e-4 * max(1.0, abs(got)), (kind, got, float(g[f"loss_gen_{kind}"]))


def test_sliding_window_causal_encode_matches_reference(golden_dir):
    """Causal attention under Mistral's sliding window: the oracle's window mask vs the reference's eager path (window 16, sequences of up
    to 200 tokens).  The fixture records how many keys a query saw in the generating run (the mask comes from transformers, and its
    releases differ by one key: oracle/gritlm_oracle.py::causal_window_mask)."""
    g = _load(golden_dir, "sliding_window_gqa.npz")
    cfg = synth.CONFIGS[str(g["cfg_name"])]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    W, keys = int(g["sliding_window"]), int(g["window_keys"])
    assert keys in (W, W + 1)
    valid = g["attention_mask"].astype(bool)
    h = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], causal=True, window=keys)
    assert np.abs(h - g["last_hidden_state"])[valid].max() < 1e-4
    other = O.mistral_encode(w, cfg, g["input_ids"], g["attention_mask"], causal=True, window=2 * W + 1 - keys)
    assert np.abs(other - g["last_hidden_state"])[valid].max() > 1e-2          # one key more or less is far outside the tolerance
    m = O.causal_window_mask(6, 3)
    assert m.sum(axis=1).tolist() == [1, 2, 3, 3, 3, 3] and bool(m[5, 3]) and not bool(m[5, 2]) and not bool(m[2, 3])
    assert np.array_equal(O.causal_window_mask(5, 0), np.tril(np.ones((5, 5), dtype=bool)))


def test_sliding_window_keys_per_attention_path():
    """Which window the engines use, per attention path of the reference (gritlm_amd/encoder.py::sliding_window_keys)."""
    from gritlm_amd.encoder import sliding_window_keys
    assert sliding_window_keys(None, "eager") == 0 and sliding_window_keys(4096, None) == 0
    assert sliding_window_keys(4096, "sdpa") == 0                 # the reference's sdpa branch passes no window: full causal attention
    assert sliding_window_keys(4096, "eager") == 4096 and sliding_window_keys(4096, "flash_attention_2") == 4097
    with pytest.raises(ValueError):
        sliding_window_keys(4096, "flex")


def test_prefix_continuation_consistent_with_pinned_causal_encode(golden_dir):
    """mistral_continue (generation on top of cached K/V) == the causal encode pinned by generative_tiny.npz when the prefix K/V
    come from a causal pass; a bidirectional (document) prefix gives different logits."""
    g = _load(golden_dir, "generative_tiny.npz")
    cfg = synth.CONFIGS["tiny"]
    w = synth.make_weights(cfg, 0)
    ids = g["input_ids"][:1, :40]
    ones = np.ones_like(ids)
    h, kv = O.mistral_encode(w, cfg, ids, ones, causal=True, return_layers="kv")
    full = h[0] @ g["lm_head"].T
    assert np.abs(full - g["logits"][0, :40]).max() < 3e-4          # (prefix of a causal sequence is independent of what follows)
    cont = O.mistral_continue(w, cfg, kv, 25, ids[0, 25:], g["lm_head"])
    assert np.abs(cont - full[25:]).max() < 1e-4
    _, kv_bi = O.mistral_encode(w, cfg, ids[:, :25], ones[:, :25], causal=False, return_layers="kv")
    cont_bi = O.mistral_continue(w, cfg, kv_bi, 25, ids[0, 25:], g["lm_head"])
    assert np.abs(cont_bi - cont).max() > 1e-2


def test_moe_router_topk_and_renormalisation():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((64, 32)).astype(np.float32)
    gw = rng.standard_normal((8, 32)).astype(np.float32)
    w, sel, logits = O.moe_router(x, gw)
    assert np.allclose(w.sum(axis=1), 1.0, atol=1e-6) and (w[:, 0] >= w[:, 1]).all()
    assert (sel[:, 0] == np.argmax(logits, axis=1)).all() and (sel[:, 0] != sel[:, 1]).all()
    import torch
    pr = torch.softmax(torch.from_numpy(x @ gw.T), dim=1, dtype=torch.float)
    tw, ts = torch.topk(pr, 2, dim=-1)
    assert np.array_equal(ts.numpy(), sel) and np.allclose((tw / tw.sum(-1, keepdim=True)).numpy(), w, atol=1e-6)


def test_rope_tables_and_rotate_half():
    cos, sin = O.rope_tables(8, 16, 10000.0)
    assert cos.shape == (8, 16) and np.allclose(cos[:, :8], cos[:, 8:])
    x = np.arange(16, dtype=np.float32)[None, None, None, :]
    r = O.rotate_half(x)
    assert np.array_equal(r[0, 0, 0, :8], -x[0, 0, 0, 8:]) and np.array_equal(r[0, 0, 0, 8:], x[0, 0, 0, :8])


def test_pool_normalize_backward_finite_difference():
    rng = np.random.default_rng(3)
    h = rng.standard_normal((2, 5, 6)).astype(np.float32)
    m = np.array([[1, 1, 1, 0, 0], [0, 1, 1, 1, 1]])
    go = rng.standard_normal((2, 6)).astype(np.float32)
    for method in ("mean", "weightedmean"):
        g = O.pool_normalize_backward(h, m, method, True, go)
        f = lambda hh: float(np.sum(O.l2_normalize(O.pooling(hh, m, method)).astype(np.float64) * go))
        eps = 1e-3
        for idx in [(0, 1, 2), (1, 4, 5), (0, 4, 0)]:
            hp = h.copy(); hp[idx] += eps; hm = h.copy(); hm[idx] -= eps
            fd = (f(hp) - f(hm)) / (2 * eps)
            assert abs(fd - g[idx]) < 2e-3, (method, idx, fd, g[idx])


@pytest.mark.parametrize("cfg_name", ["moe-tiny", "moe-gqa"])
def test_mixtral_fp32_restatement(golden_dir, cfg_name):
    """oracle/torch_reference.py::mixtral_hidden_states_fp32 (the layer-streamed fp32 restatement tools/mixtral_bench.py checks the
    configs[3] leg against) reproduces the REFERENCE's MixtralModel(is_causal=False) fp32 outputs and pooled embeddings
    (scripts/modeling_mixtral_gritlm.py:815-934, fixtures generated by the reference itself) -- and so does the engine-weight route
    (repacked QKV / interleaved expert weights widened back), which is what the GPU leg uses."""
    import torch
    import torch_reference as TR
    g = _load(golden_dir, f"encoder_{cfg_name}.npz")
    cfg = synth.CONFIGS[cfg_name]
    w = synth.make_weights(cfg, int(g["seed_w"]))
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    nq, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    args = (cfg["num_hidden_layers"], sd["embed_tokens.weight"], sd["norm.weight"], ids, mask, nq, nkv, cfg["hidden_size"] // nq,
            cfg["rms_norm_eps"], cfg["rope_theta"])
    wl = lambda li: TR.mixtral_layer_weights_from_state_dict(sd, li, cfg["num_local_experts"])
    h = TR.mixtral_hidden_states_fp32(wl, *args).numpy()
    valid = g["attention_mask"].astype(bool)
    assert np.abs(h - g["last_hidden_state"])[valid].max() < 2e-5
    e = TR.mixtral_encode_fp32(wl, *args).numpy()
    assert np.all(1 - np.sum(e * g["emb_mean"], axis=1) < 1e-6)

    # the route the GPU leg takes: weights widened back from an engine-style repack (fused QKV, gate / up rows interleaved in blocks of 16)
    class _L:
        pass

    class _E:
        pass
    blk, eng = 16, _E()
    eng.cfg = type("C", (), dict(num_attention_heads=nq, num_key_value_heads=nkv, head_dim=cfg["hidden_size"] // nq,
                                 intermediate_size=cfg["intermediate_size"], hidden_size=cfg["hidden_size"], num_local_experts=cfg["num_local_experts"]))()
    eng.layers = []
    for li in range(cfg["num_hidden_layers"]):
        W, L = wl(li), _L()
        I, H, E = cfg["intermediate_size"], cfg["hidden_size"], cfg["num_local_experts"]
        L.wqkv = torch.cat([W["wq"], W["wk"], W["wv"]]).bfloat16(); L.wo = W["wo"].bfloat16(); L.wgate = W["gate"].bfloat16()
        L.ln1, L.ln2 = W["ln1"].bfloat16(), W["ln2"].bfloat16()
        L.w13 = torch.stack([W["w1"].view(E, I // blk, blk, H), W["w3"].view(E, I // blk, blk, H)], dim=2).reshape(E, 2 * I, H).bfloat16()
        L.w2 = W["w2"].bfloat16()
        eng.layers.append(L)
    h2 = TR.mixtral_hidden_states_fp32(lambda li: TR.mixtral_layer_weights_from_engine(eng, li, blk), *args).numpy()
    assert np.array_equal(h2, h)                       # the fixtures' weights are bf16-representable: the round trip is exact


def test_numeric_bounds_match_committed_fixtures(golden_dir):
    """tests/golden/numeric_bounds.json (the yardsticks the GPU checks read instead of the fixtures' host-dependent `*_bf16` arrays) is
    what make_bounds.py computes from the committed fixtures: regenerating a fixture on another host without re-freezing the bounds --
    i.e. moving a tolerance silently -- fails here; every key a GPU check reads is present and positive or a fraction."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_bounds", os.path.join(golden_dir, "make_bounds.py"))
    mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
    frozen = json.load(open(os.path.join(golden_dir, "numeric_bounds.json")))
    assert frozen["frozen_on"]                                   # the generating host's CPU model is recorded
    now = mb.compute()
    assert set(now) == set(frozen["values"])
    for k, v in now.items():
        assert v == pytest.approx(frozen["values"][k], rel=1e-6, abs=1e-12), k
        assert np.isfinite(v) and v >= 0.0, k
    import re
    src = open(os.path.join(os.path.dirname(golden_dir), "gpu_checks.py")).read()
    for lit in re.findall(r'_yard\(\)\[f?"([^"]+)"\]', src):      # every key pattern a check reads resolves to frozen keys
        pat = re.escape(lit)
        pat = re.sub(r'\\\{[a-z_]+\\\}', r'[^/]+' if lit.startswith(("encoder_", "gritlm_")) else r'.+', pat)
        assert any(re.fullmatch(pat, k) for k in frozen["values"]), lit


def test_torch_reference_training_forward_is_the_encode_with_exact_gradients():
    """oracle/torch_reference.py::encode_with_grad / contrastive_loss (the reference side of bench.py's contrastive parity object): the
    differentiable encode equals the no-grad `encode` bit for bit, gradient checkpointing changes no gradient, and the loss is the oracle's
    InfoNCE (gritlm/training/model.py:36-47) on the same representations."""
    import torch
    import torch_reference as TR
    import gritlm_oracle as O
    cfg = synth.CONFIGS["tiny"]
    model = TR.build_model(cfg, torch.float32, "cpu", seed=3)
    ids, mask = synth.make_batch(cfg, 6, 20, seed=8, min_len=5)
    tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ref = TR.encode(model, tid, tm)
    grads = {}
    for ck in (False, True):
        model.zero_grad(set_to_none=True)
        model.train()
        e = TR.encode_with_grad(model, tid, tm, checkpoint=ck)
        assert torch.equal(e.detach(), ref)
        loss = TR.contrastive_loss(e[:2], e[2:], 0.02)
        loss.backward()
        grads[ck] = {n: p.grad.clone() for n, p in model.named_parameters()}
        l_oracle = O.infonce(e[:2].detach().numpy(), e[2:].detach().numpy(), 0.02)[0]
        assert abs(float(loss.detach()) - l_oracle) < 2e-5
    for n in grads[False]:
        assert torch.allclose(grads[False][n], grads[True][n], rtol=0, atol=1e-7 * float(grads[False][n].abs().max() + 1e-30)), n
