"""CPU: the training entry point end to end on reference-format jsonl rows (query / pos / neg, 7 negatives like
gritlm/training/toy_data) through the drop-in module path ``gritlm.training.run``; GradCache switch; outputs."""
import json
import os

import numpy as np
import torch

import synth


def _toy(path, instruct=False):
    W = synth.WORDS
    rows = []
    for i in range(0, 40, 2):
        q, pos, negs = " ".join(W[i:i + 5]), " ".join(W[i + 1:i + 9]), [" ".join(W[j:j + 7]) for j in range(i + 20, i + 27)]
        if instruct:
            rows.append({"query": ["w1 w2", q], "pos": [["w3", pos]], "neg": [["w3", n] for n in negs]})
        else:
            rows.append({"query": q, "pos": [pos], "neg": negs})
    with open(path, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows))
    return path


def test_cli_gradcache_two_steps(tmp_path):
    from gritlm.training.run import main          # the drop-in alias
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    out = str(tmp_path / "out")
    loss = main(["--model_name_or_path", d, "--train_data", data, "--output_dir", out, "--per_device_train_batch_size", "2",
                 "--gradient_accumulation_steps", "2", "--no_gen_gas", "--no_emb_gas", "--train_group_size", "8", "--pooling_method", "mean",
                 "--max_steps", "2", "--learning_rate", "1e-4", "--query_max_len", "16", "--passage_max_len", "24", "--report_to", "none",
                 "--use_cpu"])
    assert np.isfinite(loss)
    files = set(os.listdir(out))
    assert {"config.json", "dataset_num_samples.json", "tokenizer.json"} <= files
    assert json.load(open(os.path.join(out, "dataset_num_samples.json"))) == {"toy.jsonl": 20}
    from safetensors import safe_open
    wfile = [f for f in files if f.endswith(".safetensors") or f.endswith(".bin")][0]
    if wfile.endswith(".safetensors"):
        with safe_open(os.path.join(out, wfile), "pt") as f:
            keys = set(f.keys())
    else:
        keys = set(torch.load(os.path.join(out, wfile)).keys())
    assert any(k.endswith("layers.0.self_attn.k_proj.weight") for k in keys)      # reference parameter names


def test_collator_instruction_format(tmp_path):
    from transformers import AutoTokenizer
    from gritlm_amd.training.data import EmbeddingCollator, EmbeddingDataset, load_embedding_rows
    synth.make_tokenizer(str(tmp_path / "tok"))
    tok = AutoTokenizer.from_pretrained(str(tmp_path / "tok"), padding_side="right")
    rows = load_embedding_rows(_toy(str(tmp_path / "toy_i.jsonl"), instruct=True))
    ds = EmbeddingDataset(rows, train_group_size=4, max_char_len=1000, seed=0)
    batch = EmbeddingCollator(tok, 32, 48)([ds[0], ds[1], ds[2]])
    assert batch["query"]["input_ids"].shape[0] == 3 and batch["passage"]["input_ids"].shape[0] == 12
    assert batch["query"]["instruction_lens"].shape == (3,) and batch["passage"]["instruction_lens"].shape == (12,)
    # the instruction prefix is a strict prefix of the row: there is text left to embed
    for i, l in enumerate(batch["query"]["instruction_lens"].tolist()):
        assert batch["query"]["attention_mask"][i, l] == 1


def test_cli_direct_step_variants(tmp_path):
    """non-GradCache step and the reference's q-only / p-only / split_emb variants (gradcache_trainer.py:584-605, 656-718)."""
    from gritlm_amd.training.run import main
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    base = ["--model_name_or_path", d, "--train_data", data, "--per_device_train_batch_size", "3", "--train_group_size", "4",
            "--pooling_method", "weightedmean", "--max_steps", "1", "--learning_rate", "1e-4", "--query_max_len", "16", "--passage_max_len", "24",
            "--report_to", "none", "--use_cpu"]
    losses = []
    for i, extra in enumerate(([], ["--emb_q_only"], ["--emb_p_only"], ["--split_emb"])):
        losses.append(main(base + ["--output_dir", str(tmp_path / f"o{i}")] + extra))
    assert all(np.isfinite(l) for l in losses)
    assert max(losses) - min(losses) < 1e-4            # same seed, same first batch: the forward loss is identical in all variants


def test_generative_collator_matches_reference(tmp_path):
    """GenerativeCollator vs the reference's CustomCollator outputs recorded in tests/golden/generative_tiny.npz (multi-turn
    samples, instruction turns masked to -100, prefixlm variant)."""
    from transformers import AutoTokenizer
    from gritlm_amd.training.data import GenerativeCollator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "generative_tiny.npz"))
    samples = json.loads(str(g["coll_samples"]))
    synth.make_tokenizer(str(tmp_path / "tok"))
    tok = AutoTokenizer.from_pretrained(str(tmp_path / "tok"), padding_side="right")
    if not tok.pad_token and tok.bos_token:
        tok.pad_token = tok.bos_token
    for prefixlm, tag in ((False, ""), (True, "_prefixlm")):
        feats = GenerativeCollator(tok, 40, prefixlm)(samples)
        assert np.array_equal(feats["input_ids"].numpy(), g["coll_input_ids" + tag])
        assert np.array_equal(feats["attention_mask"].numpy(), g["coll_attention_mask" + tag])
        assert np.array_equal(feats["labels"].numpy(), g["coll_labels" + tag])
    assert (g["coll_labels"] != g["coll_labels_prefixlm"]).any()


def test_cli_unified_two_steps(tmp_path):
    """--mode unified on CPU (Hugging Face path): generative rows ("text") + embedding rows in one directory, generative step first,
    then the GradCache embedding step; both losses finite."""
    from gritlm.training import run
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    os.makedirs(tmp_path / "data")
    _toy(str(tmp_path / "data" / "emb.jsonl"))
    W = synth.WORDS
    with open(tmp_path / "data" / "gen.jsonl", "w") as f:
        f.write("\n".join(json.dumps({"text": [" ".join(W[i:i + 4]), " ".join(W[i + 30:i + 42])]}) for i in range(24)))
    loss = run.main(["--model_name_or_path", d, "--train_data", str(tmp_path / "data"), "--output_dir", str(tmp_path / "out"), "--mode", "unified",
                     "--per_device_train_batch_size", "2", "--gradient_accumulation_steps", "2", "--no_gen_gas", "--no_emb_gas",
                     "--per_device_generative_bs", "2", "--train_group_size", "8", "--pooling_method", "mean", "--max_steps", "2",
                     "--learning_rate", "1e-4", "--query_max_len", "16", "--passage_max_len", "24", "--generative_max_len", "32",
                     "--loss_gen_type", "mixed", "--report_to", "none", "--use_cpu"])
    assert np.isfinite(loss) and run.main.last_loss_gen is not None and np.isfinite(run.main.last_loss_gen)


def _base(tmp_path, d, data, out, *extra):
    return ["--model_name_or_path", d, "--train_data", data, "--output_dir", str(tmp_path / out), "--train_group_size", "4",
            "--pooling_method", "mean", "--learning_rate", "1e-3", "--query_max_len", "16", "--passage_max_len", "24", "--report_to", "none",
            "--use_cpu", "--lr_scheduler_type", "constant", "--weight_decay", "0", *extra]


def _weights(out_dir):
    from safetensors.torch import load_file
    f = [x for x in os.listdir(out_dir) if x.endswith(".safetensors")]
    return load_file(os.path.join(out_dir, f[0])) if f else torch.load(os.path.join(out_dir, "pytorch_model.bin"))


def test_cli_unsupported_flags_raise(tmp_path):
    """Flags the reference accepts and this entry point does not implement must raise, e.g. --lora would otherwise
    silently run a full fine-tune."""
    import pytest
    from gritlm_amd.training.run import main
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    for flag in (["--lora"], ["--qlora"], ["--split_emb_full"]):
        with pytest.raises(NotImplementedError):
            main(_base(tmp_path, d, data, "o", "--per_device_train_batch_size", "2", "--max_steps", "1", *flag))
    with pytest.raises(ValueError):          # run.py:105-106
        main(_base(tmp_path, d, data, "o", "--per_device_train_batch_size", "2", "--max_steps", "1", "--no_emb_gas"))


def test_cli_gradient_accumulation_without_gradcache(tmp_path):
    """GAS > 1 without --negatives_cross_device accumulates (HF Trainer semantics): the optimizer steps once per GAS micro-batches on
    the mean gradient -- with SGD-like AdamW at step 1 the update direction equals that of ONE step on the doubled batch order."""
    from gritlm_amd.training.run import main
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    main(_base(tmp_path, d, data, "acc", "--per_device_train_batch_size", "2", "--gradient_accumulation_steps", "3", "--max_steps", "2"))
    main(_base(tmp_path, d, data, "noacc", "--per_device_train_batch_size", "2", "--max_steps", "2"))
    w0 = {k: torch.from_numpy(v) for k, v in synth.make_weights(synth.CONFIGS["tiny"], 0).items()}
    wa, wn = _weights(str(tmp_path / "acc")), _weights(str(tmp_path / "noacc"))
    k = "layers.0.self_attn.q_proj.weight"
    assert not torch.equal(wa[k], wn[k])                       # six micro-batches vs two: different trajectories
    assert (wa[k].float() - w0[k]).abs().max() > 0


def test_cli_checkpoint_and_resume_reproduce_the_uninterrupted_run(tmp_path):
    """--save_steps writes checkpoint-<step>/ (weights + optimizer + scheduler + step); --resume_from_checkpoint continues with the same
    data order: 2 steps + resume for 2 more == 4 uninterrupted steps (fp32, CPU, deterministic)."""
    from gritlm_amd.training.run import main
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    common = ["--per_device_train_batch_size", "2", "--save_strategy", "steps", "--save_steps", "2", "--save_safetensors", "true"]
    main(_base(tmp_path, d, data, "full", *common, "--max_steps", "4"))
    ck = str(tmp_path / "full" / "checkpoint-2")
    assert {"optimizer.pt", "scheduler.pt", "trainer_state.json"} <= set(os.listdir(ck))
    main(_base(tmp_path, d, data, "resumed", *common, "--max_steps", "4", "--resume_from_checkpoint", ck))
    wf, wr = _weights(str(tmp_path / "full")), _weights(str(tmp_path / "resumed"))
    for k in wf:
        assert torch.allclose(wf[k].float(), wr[k].float(), rtol=0, atol=1e-6), k


def test_cli_projection_head_is_trained_and_saved(tmp_path):
    from gritlm_amd.training.run import main
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    data = _toy(str(tmp_path / "toy.jsonl"))
    torch.manual_seed(0)
    main(_base(tmp_path, d, data, "p", "--per_device_train_batch_size", "2", "--max_steps", "2", "--projection", "32"))
    sd = torch.load(str(tmp_path / "p" / "projection.pt"))
    torch.manual_seed(0)
    init = torch.nn.Linear(256, 32)
    assert sd["weight"].shape == (32, 256) and not torch.allclose(sd["weight"], init.weight)        # the optimizer moved it
