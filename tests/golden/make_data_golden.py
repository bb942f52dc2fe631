#!/usr/bin/env python3
"""Golden values for the host-side data path, produced by the REFERENCE's own code (imported from /root/reference):
  * gritlm/training/data.py::CustomRandomSampler -- the index stream for several (dataset lengths, global batch size, seed) cases;
  * gritlm/training/run.py::filter_too_long_instructions -- which (instruction, text) samples survive, with the synthetic test tokenizer;
  * gritlm/training/data.py::CustomDataset.set_indices / __getitem__ index choice under --use_unique_indices (single process).
    python tests/golden/make_data_golden.py      (writes tests/golden/data_pipeline.json)"""
import json
import os
import random
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
import synth  # noqa: E402
from gritlm.training.data import CustomDataset, CustomRandomSampler  # noqa: E402
from gritlm.training import run as ref_run  # noqa: E402

out = {"sampler": [], "filter": {}, "unique_indices": {}}
for ds_lens, total_bs, seed in [([100, 150, 50], 8, 0), ([10, 3, 25], 4, 1), ([5, 5], 8, 2), ([16, 8], 8, 3), ([7], 2, 4), ([3, 0, 9], 4, 5),
                                ([33, 17, 4, 90], 16, 6)]:
    s = CustomRandomSampler(data_source=list(range(sum(ds_lens))))
    s.total_batch_size, s.ds_lens, s._num_samples = total_bs, ds_lens, sum(ds_lens)
    s.generator = torch.Generator().manual_seed(seed)
    first = list(iter(s))
    second = list(iter(s))                       # a second epoch continues the same generator
    out["sampler"].append({"ds_lens": ds_lens, "total_batch_size": total_bs, "seed": seed, "epoch1": first, "epoch2": second})

import datasets  # noqa: E402
from transformers import AutoTokenizer  # noqa: E402
td = tempfile.mkdtemp()
synth.make_tokenizer(td)
tok = AutoTokenizer.from_pretrained(td, padding_side="right")
W = synth.WORDS
examples = []
for i in range(24):
    instr = " ".join(W[(i * 3) % 40:(i * 3) % 40 + 1 + (i % 9)])
    if i % 7 == 3:
        instr = instr + " :\n"                    # stripped characters at the end of an instruction
    q = [instr, " ".join(W[i:i + 4])]
    if i == 5:
        q = [instr, ""]                           # empty text: filtered
    if i == 11:
        q = ["x" * 400, "text"]                   # longer than 10 x query_max_len characters: filtered before tokenising
    pos = [[" ".join(W[j:j + 1 + (i % 5)]), " ".join(W[j + 2:j + 6])] for j in (i, i + 1)]
    neg = [[" ".join(W[j:j + 2 + (i % 11)]), " ".join(W[j + 3:j + 5])] for j in (i + 5, i + 9, i + 12)]
    if i == 17:
        neg[1][1] = ""                            # an empty negative text: filtered
    examples.append({"query": q, "pos": pos, "neg": neg})
ds = datasets.Dataset.from_list(examples)
for qmax, pmax in [(8, 12), (12, 8), (6, 6), (32, 32)]:
    kept = ref_run.filter_too_long_instructions(tok, ds, qmax, pmax)
    kept_q = [e["query"] for e in kept]
    out["filter"][f"{qmax},{pmax}"] = [ex["query"] in kept_q for ex in examples]
out["filter_examples"] = examples

# --use_unique_indices, one process: the smaller data set's indices are handed out from a set that is refilled when empty
for len_emb, len_gen in [(5, 12), (12, 5), (6, 6)]:
    args = types.SimpleNamespace(use_unique_indices=True, train_group_size=2)
    emb = [{"query": f"q{i}", "pos": [f"p{i}"], "neg": [f"n{i}"]} for i in range(len_emb)]
    gen = [{"text": f"t{i}"} for i in range(len_gen)]
    random.seed(0)
    cd = CustomDataset([emb, gen], args, tokenizer=None, mode="unified", full_bs=2, generative_bs=None, max_seq_len=64)
    picks = []
    for item in range(2 * max(len_emb, len_gen)):
        q, p, g = cd[item % max(len_emb, len_gen)]
        picks.append([q, g])
    out["unique_indices"][f"{len_emb},{len_gen}"] = picks
json.dump(out, open(os.path.join(HERE, "data_pipeline.json"), "w"))
print("wrote data_pipeline.json", {k: len(v) for k, v in out.items()})
