"""Host-side data path: (query, pos, negs) sampling, generative (instruction, response, ...) turns and the GRIT prompt format.

Out of the accelerated scope -- string / tokeniser work on the host.  It exists so that
``python -m gritlm_amd.training.run`` is self-contained; the batch layout it emits is the one the reference's collator
emits (gritlm/training/data.py:230-269): ``query [B, Lq]``, ``passage [B*G, Lp]`` with row ``i*G`` the positive of query
``i``, plus per-row ``instruction_lens`` when samples are (instruction, text) pairs.
"""
from __future__ import annotations

import json
import math
import os
import random
from dataclasses import dataclass

import torch

BASE_BOS, USER_BOS, USER_EOS, EMBED_BOS, EMBED_EOS = "<s>", "<|user|>\n", "", "\n<|embed|>\n", ""
TURN_SEP, ASSISTANT_BOS, ASSISTANT_EOS = "\n", "\n<|assistant|>\n", "</s>"        # gritlm/training/run.py:18-30


def load_embedding_rows(path: str, limit: int | None = None) -> list[dict]:
    files = sorted(os.path.join(path, f) for f in os.listdir(path)) if os.path.isdir(path) else [path]
    rows = []
    for f in files:
        with open(f) as fh:
            part = [json.loads(line) for line in fh if line.strip()]
        part = [r for r in part if "query" in r]
        rows += part[:limit] if limit else part
    return rows


def load_generative_rows(path: str, limit: int | None = None) -> list:
    """Rows with a "text" field: a string, or [instruction, response, instruction, response, ...] (run.py:166-179)."""
    files = sorted(os.path.join(path, f) for f in os.listdir(path)) if os.path.isdir(path) else [path]
    rows = []
    for f in files:
        with open(f) as fh:
            part = [json.loads(line) for line in fh if line.strip()]
        part = [r["text"] for r in part if "text" in r]
        rows += part[:limit] if limit else part
    return rows


@dataclass
class GenerativeCollator:
    """The generative half of the reference's CustomCollator (gritlm/training/data.py:214-228, :248-282): chat formatting of
    multi-turn samples, labels = input_ids with padding (except position 0) and every instruction turn set to -100
    (``prefixlm``: every turn before the last assistant utterance, :279)."""
    tokenizer: object
    generative_max_len: int = 128
    prefixlm: bool = False

    def __call__(self, samples):
        tok = self.tokenizer
        lens = None
        if isinstance(samples[0], (tuple, list)):
            lens = [[len(tok.tokenize((BASE_BOS if i == 0 else "") + USER_BOS + z + USER_EOS + ASSISTANT_BOS)) if i % 2 == 0
                     else len(tok.tokenize(z.strip() + ASSISTANT_EOS)) for i, z in enumerate(f[:-1])] for f in samples]
            samples = [BASE_BOS + TURN_SEP.join(USER_BOS + f[i] + USER_EOS + ASSISTANT_BOS + f[i + 1].strip() + ASSISTANT_EOS
                                                for i in range(0, len(f), 2)) for f in samples]
        feats = dict(tok(samples, padding=True, truncation=True, max_length=self.generative_max_len, return_tensors="pt",
                         add_special_tokens=False))
        labels = feats["input_ids"].clone()
        labels[:, 1:][labels[:, 1:] == tok.pad_token_id] = -100          # position 0 may legitimately be the pad (= bos) token
        if lens:
            for i, turn_lens in enumerate(lens):
                cur = 0
                for j, l in enumerate(turn_lens):
                    if (j % 2 == 0) or self.prefixlm:
                        labels[i, cur:cur + l] = -100
                    cur += l
        feats["labels"] = labels
        return feats


class EmbeddingDataset(torch.utils.data.Dataset):
    def __init__(self, rows: list[dict], train_group_size: int, max_char_len: int, seed: int = 0):
        self.rows, self.g, self.max_char_len = rows, train_group_size, max_char_len
        self.rng = random.Random(seed)

    def __len__(self):
        return len(self.rows)

    def _clip(self, x):
        return x[: self.max_char_len] if isinstance(x, str) else [t[: self.max_char_len] for t in x]

    def __getitem__(self, i):
        row = self.rows[i]
        query = self._clip(row["query"])
        pos = self._clip(self.rng.choice(row["pos"]))
        need = self.g - 1
        negs = row["neg"]
        if len(negs) < need:
            negs = negs * math.ceil(need / max(len(negs), 1))
        negs = [self._clip(n) for n in self.rng.sample(negs, need)]
        return query, [pos] + negs


def _prompt(sample, embed_eos):
    """(instruction, text) -> GRIT embedding prompt; returns (prompt, instruction_prefix)."""
    instr = sample[0].strip("\t\n :")
    prefix = BASE_BOS + USER_BOS + instr + USER_EOS + EMBED_BOS if instr else BASE_BOS + EMBED_BOS.lstrip()
    return prefix + sample[1] + embed_eos, prefix


@dataclass
class EmbeddingCollator:
    tokenizer: object
    query_max_len: int = 32
    passage_max_len: int = 128
    embed_eos: str = EMBED_EOS

    def _encode(self, texts, max_len):
        return self.tokenizer(texts, padding=True, truncation=True, max_length=max_len, return_tensors="pt", add_special_tokens=False)

    def __call__(self, samples):
        queries = [s[0] for s in samples]
        passages = [p for s in samples for p in s[1]]
        q_lens = p_lens = None
        if isinstance(queries[0], (tuple, list)):
            qp = [_prompt(q, self.embed_eos) for q in queries]
            pp = [_prompt(p, self.embed_eos) for p in passages]
            q_lens = [len(self.tokenizer.tokenize(pre)) for _, pre in qp]
            p_lens = [len(self.tokenizer.tokenize(pre)) for _, pre in pp]
            queries, passages = [t for t, _ in qp], [t for t, _ in pp]
        feats = {"query": dict(self._encode(queries, self.query_max_len)), "passage": dict(self._encode(passages, self.passage_max_len))}
        if q_lens is not None:
            feats["query"]["instruction_lens"] = torch.tensor(q_lens)
            feats["passage"]["instruction_lens"] = torch.tensor(p_lens)
        return feats
