"""Host-side data path against values the REFERENCE's own code produced (tests/golden/data_pipeline.json, written by
tests/golden/make_data_golden.py importing gritlm/training/{data,run}.py): the multi-data-set sampler's index stream, the
too-long-instruction filter, the index choice under --use_unique_indices; plus the per-file loader (subsample, --num_samples cap,
dataset_num_samples.json) and the CLI on a directory of several embedding files (every global batch from ONE file)."""
import json
import os
import random
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import synth  # noqa: E402
from gritlm_amd.training import data as D  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "data_pipeline.json")))


@pytest.mark.parametrize("case", GOLD["sampler"], ids=lambda c: f"{c['ds_lens']}x{c['total_batch_size']}")
def test_multi_dataset_order_is_the_reference_samplers_stream(case):
    g = torch.Generator().manual_seed(case["seed"])
    assert D.multi_dataset_order(case["ds_lens"], case["total_batch_size"], g) == case["epoch1"]
    assert D.multi_dataset_order(case["ds_lens"], case["total_batch_size"], g) == case["epoch2"]        # the generator carries on
    # what the stream is for: a global batch comes from one data set, except the mixed batches built from the tails
    bounds = [sum(case["ds_lens"][:j + 1]) for j in range(len(case["ds_lens"]))]
    owner = lambda i: next(j for j, b in enumerate(bounds) if i < b)
    tb = case["total_batch_size"]
    batches = [case["epoch1"][k:k + tb] for k in range(0, len(case["epoch1"]), tb)]
    pure = sum(len({owner(i) for i in b}) == 1 for b in batches)
    assert pure >= sum(n // tb for n in case["ds_lens"]) and all(len(b) == tb for b in batches)
    assert len(set(case["epoch1"])) == len(case["epoch1"])                                             # no index twice in an epoch


def test_per_device_batches_are_dealt_to_the_ranks_in_turn():
    order = list(range(48))                                          # 2 global batches of bs 3 x gas 2 x world 4
    shares = [D.deal_to_rank(order, 3, r, 4) for r in range(4)]
    assert shares[0] == [0, 1, 2, 12, 13, 14, 24, 25, 26, 36, 37, 38] and shares[3][:3] == [9, 10, 11]
    assert sorted(i for s in shares for i in s) == order
    # micro-batches m = 0, 1 of optimizer step 0 on every rank: together exactly the first global batch (indices 0 .. 23)
    assert sorted(i for s in shares for i in s[:6]) == list(range(24))


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    from transformers import AutoTokenizer
    d = str(tmp_path_factory.mktemp("tok"))
    synth.make_tokenizer(d)
    return AutoTokenizer.from_pretrained(d, padding_side="right")


@pytest.mark.parametrize("lens", sorted(GOLD["filter"]))
def test_instruction_filter_keeps_what_the_reference_keeps(tok, lens):
    qmax, pmax = (int(x) for x in lens.split(","))
    kept = [not D.instruction_too_long(tok, ex, qmax, pmax) for ex in GOLD["filter_examples"]]
    assert kept == GOLD["filter"][lens]
    assert not kept[5] and not kept[11] and not kept[17]             # empty text / 400-character instruction / empty negative text


@pytest.mark.parametrize("lens", sorted(GOLD["unique_indices"]))
def test_unique_indices_follow_the_reference_dataset(lens):
    len_emb, len_gen = (int(x) for x in lens.split(","))
    rng = random.Random(0)
    unique = len_emb != len_gen
    emb = D.ItemPicker(len_emb, rng, unique and len_emb < len_gen)
    gen = D.ItemPicker(len_gen, rng, unique and len_gen < len_emb)
    total = max(len_emb, len_gen)
    picks = [[f"q{e}", f"t{g}"] for e, g in (D.pick_items(emb, gen, i % total) for i in range(2 * total))]
    assert picks == GOLD["unique_indices"][lens]


def test_item_picker_past_the_end_and_rank_shares():
    rng = random.Random(1)
    p = D.ItemPicker(4, rng)
    assert [p(i) for i in range(4)] == [0, 1, 2, 3] and all(0 <= p(i) < 4 for i in range(4, 50))
    a, b = D.ItemPicker(7, rng, True, 0, 2), D.ItemPicker(7, rng, True, 1, 2)
    assert sorted(a(99) for _ in range(4)) == [0, 2, 4, 6] and sorted(b(0) for _ in range(3)) == [1, 3, 5]
    assert sorted(a(0) for _ in range(4)) == [0, 2, 4, 6]                                              # refilled when empty


def _write(path, rows):
    with open(path, "w") as f:
        f.write("\n".join(json.dumps(r) for r in rows) + "\n")


def _emb_rows(n, tag, instruct=False):
    W = synth.WORDS
    rows = []
    for i in range(n):
        q, pos, negs = f"{tag} " + " ".join(W[i:i + 4]), " ".join(W[i + 1:i + 6]), [" ".join(W[j:j + 5]) for j in range(i + 10, i + 13)]
        rows.append({"query": ["w1 w2", q], "pos": [["w3", pos]], "neg": [["w3", n_] for n_ in negs]} if instruct else
                    {"query": q, "pos": [pos], "neg": negs})
    return rows


def test_load_datasets_per_file(tok, tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    _write(d / "a.jsonl", _emb_rows(12, "a"))
    _write(d / "b.jsonl", _emb_rows(30, "b", instruct=True))
    _write(d / "c.jsonl", [{"text": ["w1 w2 w3", "w4 w5"]}] * 5 + [{"text": [" ".join(synth.WORDS[:60]), "w1"]}])
    _write(d / "d.jsonl", [{"other": 1}])
    random.seed(0)
    emb, gen, kept = D.load_datasets(str(d), "unified", tok, 16, 24, 32, max_example_num_per_dataset=20, num_samples={"b.jsonl": 7})
    assert [n for n, _ in emb] == ["a.jsonl", "b.jsonl"] and [n for n, _ in gen] == ["c.jsonl"]
    assert kept == {"a.jsonl": 12, "b.jsonl": 7, "c.jsonl": 5}        # b: 30 -> 20 (per-file maximum) -> 7 (--num_samples); c: the 60-word instruction is dropped
    assert len({json.dumps(r) for r in emb[1][1]}) == 7                # a subsample WITHOUT replacement
    assert all(r["query"][1].startswith("b ") for r in emb[1][1])
    emb2, gen2, kept2 = D.load_datasets(str(d), "embedding", tok, 16, 24, 32)
    assert gen2 == [] and kept2 == {"a.jsonl": 12, "b.jsonl": 30}       # --num_samples / the maximum are caps, not targets
    with pytest.raises(AssertionError, match="Missing num_samples for b.jsonl"):
        D.load_datasets(str(d), "embedding", tok, 16, 24, 32, num_samples={"a.jsonl": 3})
    # plain-string data sets are neither filtered nor capped by --num_samples (the reference applies both inside the (instruction, text) branch)
    _, _, kept3 = D.load_datasets(str(d), "embedding", tok, 16, 24, 32, num_samples={"b.jsonl": 1000})
    assert kept3 == {"a.jsonl": 12, "b.jsonl": 30}


def test_cli_on_several_embedding_files_draws_each_global_batch_from_one_file(tok, tmp_path, monkeypatch):
    from gritlm_amd.training import run
    model_dir = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    d = tmp_path / "data"
    d.mkdir()
    _write(d / "a.jsonl", _emb_rows(16, "a"))
    _write(d / "b.jsonl", _emb_rows(9, "b"))
    (d / "empty.jsonl").write_text("")                                   # an empty file is skipped
    (tmp_path / "num.json").write_text(json.dumps({"a.jsonl": 100, "b.jsonl": 100}))
    seen = []
    orig = D.EmbeddingCollator.__call__

    def spy(self, samples):
        seen.append([s[0].split()[0] for s in samples])               # the data-set tag in front of every query
        return orig(self, samples)
    monkeypatch.setattr(D.EmbeddingCollator, "__call__", spy)
    argv = ["--model_name_or_path", model_dir, "--train_data", str(d), "--output_dir", str(tmp_path / "out"), "--per_device_train_batch_size", "4",
            "--train_group_size", "2", "--pooling_method", "mean", "--max_steps", "5", "--learning_rate", "1e-4", "--query_max_len", "16",
            "--passage_max_len", "24", "--report_to", "none", "--use_cpu", "--num_samples", str(tmp_path / "num.json")]
    with pytest.raises(AssertionError, match="dataloader_drop_last"):
        run.main(argv)
    loss = run.main(argv + ["--dataloader_drop_last"])
    assert loss == loss and len(seen) == 5
    # 16 + 9 rows, global batch 4: four batches of file a, two of file b, the tail of b (1 row) is dropped -> an epoch has 6 batches,
    # every one of them from a single file
    assert all(len(set(b)) == 1 for b in seen)
    assert json.load(open(tmp_path / "out" / "dataset_num_samples.json")) == {"a.jsonl": 16, "b.jsonl": 9}


def test_cli_unified_with_unique_indices(tmp_path):
    from gritlm_amd.training import run
    model_dir = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32")
    d = tmp_path / "data"
    d.mkdir()
    _write(d / "emb.jsonl", _emb_rows(12, "e"))
    _write(d / "gen.jsonl", [{"text": [" ".join(synth.WORDS[i:i + 3]), " ".join(synth.WORDS[i + 5:i + 12])]} for i in range(5)])
    loss = run.main(["--model_name_or_path", model_dir, "--train_data", str(d), "--output_dir", str(tmp_path / "out"), "--mode", "unified",
                     "--per_device_train_batch_size", "2", "--train_group_size", "2", "--pooling_method", "mean", "--max_steps", "4",
                     "--learning_rate", "1e-4", "--query_max_len", "16", "--passage_max_len", "24", "--generative_max_len", "32",
                     "--report_to", "none", "--use_cpu", "--use_unique_indices"])
    assert loss == loss and run.main.last_loss_gen == run.main.last_loss_gen
