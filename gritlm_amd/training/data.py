"""Host-side data path: (query, pos, negs) sampling, generative (instruction, response, ...) turns and the GRIT prompt format.

Out of the accelerated scope -- string / tokeniser work on the host -- but it decides WHAT the hot path is fed, so since
round 4 it follows the reference's data path step by step (each function cites it): one data set per file with the reference's
subsample / too-long-instruction filter / ``--num_samples`` cap (gritlm/training/run.py:122-204), global batches drawn from one data
set each when there are several (``CustomRandomSampler``, data.py:284-352 -- the index stream is identical to the reference's for a
given generator), the row choice of ``CustomDataset.__getitem__`` incl. ``--use_unique_indices`` (data.py:53-139).  The batch layout
the collators emit is the reference collator's (data.py:230-269): ``query [B, Lq]``, ``passage [B*G, Lp]`` with row ``i*G`` the
positive of query ``i``, plus per-row ``instruction_lens`` when samples are (instruction, text) pairs.
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


def _data_files(path: str) -> list[str]:
    """A JSONL file, or every file of a directory (sorted: the reference takes os.listdir's order, which no two file systems share)."""
    return sorted(os.path.join(path, f) for f in os.listdir(path)) if os.path.isdir(path) else [path]


def instruction_too_long(tokenizer, example: dict, query_max_len: int, passage_max_len: int) -> bool:
    """The per-sample test of ``filter_too_long_instructions`` (gritlm/training/run.py:38-52) for samples whose query / passages are
    (instruction, text) pairs: a sample is DROPPED when an instruction alone fills the sequence (its tokens are masked out of the pooling,
    so nothing would be left) or a text is empty.  Character-length shortcut first (10 x max_len), then the formatted instruction's
    token count."""
    def over(pair, max_len):
        if len(pair[0]) > max_len * 10 or not pair[1]:
            return True
        return len(tokenizer.tokenize(BASE_BOS + USER_BOS + pair[0].strip("\t\n :") + USER_EOS + EMBED_BOS)) >= max_len
    if over(example["query"], query_max_len):
        return True
    return any(over(ex, passage_max_len) for ex in list(example["pos"]) + list(example["neg"]))


def load_datasets(path: str, mode: str, tokenizer=None, query_max_len: int = 32, passage_max_len: int = 128, generative_max_len: int = 128,
                  max_example_num_per_dataset: int = 100_000_000, num_samples: dict | None = None):
    """The data-set loop of gritlm/training/run.py:122-204, one entry per FILE: ``(embedding, generative, counts)`` with
    ``embedding = [(file name, rows)]`` (rows with a "query"; used by modes embedding / unified) and ``generative = [(file name, texts)]``
    (rows with a "text"; unified / generative).  Per file, in the reference's order: a uniform subsample without replacement when the
    file holds more than ``max_example_num_per_dataset`` rows; for (instruction, text) samples the too-long-instruction filter and then
    the ``num_samples[file name]`` cap (a file missing from ``num_samples`` is an error, as in the reference); multi-turn generative
    samples whose first instruction fills ``generative_max_len`` are dropped.  ``counts`` = rows kept per file
    (``dataset_num_samples.json``).  Subsamples use the ``random`` module (seeded by ``set_seed``), as the reference's do."""
    emb, gen, counts = [], [], {}
    for f in _data_files(path):
        with open(f) as fh:
            rows = [json.loads(line) for line in fh if line.strip()]
        if not rows:
            continue
        name = os.path.basename(f)
        if len(rows) > max_example_num_per_dataset:
            rows = [rows[i] for i in random.sample(range(len(rows)), max_example_num_per_dataset)]
        if mode in ("embedding", "unified") and "query" in rows[0]:
            if isinstance(rows[0]["query"], (tuple, list)):
                if tokenizer is not None:
                    rows = [r for r in rows if not instruction_too_long(tokenizer, r, query_max_len, passage_max_len)]
                if num_samples:
                    if name not in num_samples:
                        raise AssertionError(f"Missing num_samples for {name}")
                    if len(rows) > num_samples[name]:
                        rows = [rows[i] for i in random.sample(range(len(rows)), num_samples[name])]
            counts[name] = len(rows)
            emb.append((name, rows))
            continue
        if mode in ("unified", "generative") and "text" in rows[0]:
            texts = [r["text"] for r in rows]
            if isinstance(texts[0], (tuple, list)) and tokenizer is not None:
                texts = [t for t in texts if len(tokenizer.tokenize(USER_BOS + t[0] + USER_EOS + ASSISTANT_BOS)) < generative_max_len]
            counts[name] = len(texts)
            gen.append((name, texts))
    return emb, gen, counts


def multi_dataset_order(ds_lens: list[int], total_batch_size: int, generator: torch.Generator) -> list[int]:
    """One epoch of the reference's ``CustomRandomSampler`` (gritlm/training/data.py:284-352): indices into the CONCATENATION of the
    data sets such that a global batch (``total_batch_size`` consecutive indices) comes from ONE data set wherever possible -- the
    in-batch negatives of the contrastive loss then share the task.  Each data set is permuted and cut into global batches; the
    incomplete tails are concatenated in a random order and cut again (an incomplete last mixed batch is dropped); finally the batches
    are permuted.  Same draws from ``generator`` in the same order as the reference: the index stream is identical
    (tests/golden/data_pipeline.json was written by the reference's sampler)."""
    starts = [sum(ds_lens[:j]) for j in range(len(ds_lens))]
    per_ds = [[start + i for i in torch.randperm(n, generator=generator).tolist()] for n, start in zip(ds_lens, starts)]
    batches, tails = [], []
    for idx in per_ds:
        cut = [idx[k:k + total_batch_size] for k in range(0, len(idx), total_batch_size)] or [[]]      # torch.split of an empty tensor: one empty piece
        if len(cut[-1]) < total_batch_size:
            tails.append(cut.pop())
        batches.append(cut)
    full = [b for cut in batches for b in cut]
    if tails:
        order = torch.randperm(len(tails), generator=generator).tolist()
        rest = [i for k in order for i in tails[k]]
        mixed = [rest[k:k + total_batch_size] for k in range(0, len(rest), total_batch_size)] or [[]]
        if len(mixed[-1]) < total_batch_size:
            mixed.pop()
        full += mixed
    order = torch.randperm(len(full), generator=generator).tolist()
    return [i for k in order for i in full[k]]


def pick_items(emb_pick, gen_pick, item: int, want_gen: bool = True):
    """(embedding row, generative row) for data-set index ``item`` as ``CustomDataset.__getitem__`` chooses them (data.py:81-139).  One
    detail of the reference is kept on purpose: an index popped from the embedding data set's unique-index pool REPLACES ``item`` for
    the generative lookup of the same sample as well (data.py:94-97 reassigns ``item`` before :132-137 read it)."""
    e = emb_pick(item) if emb_pick is not None else None
    if emb_pick is not None and emb_pick.pool is not None:
        item = e
    g = gen_pick(item) if (gen_pick is not None and want_gen) else None
    return e, g


def deal_to_rank(order: list[int], per_device_batch: int, rank: int, world: int) -> list[int]:
    """This rank's indices of a global index stream: the stream is cut into per-device batches and batch j belongs to rank j % world
    (what accelerate's batch-sampler shard does under the HF Trainer), so a global batch of ``per_device_batch x gas x world`` consecutive
    indices -- one data set, see ``multi_dataset_order`` -- is spread over all ranks and accumulation steps."""
    return [i for j in range(rank, len(order) // per_device_batch, world) for i in order[j * per_device_batch:(j + 1) * per_device_batch]]


class ItemPicker:
    """Which row of a data set of ``length`` rows answers dataset index ``item`` (gritlm/training/data.py:53-77, :92-97, :132-137):
    the index itself while it is in range; past the end a uniformly random row; with ``--use_unique_indices`` (and data sets of
    different lengths, unified mode) the SMALLER data set hands out the indices of this rank's share (``range(length)[rank::world]``)
    from a set that is refilled when it runs empty, whatever ``item`` is."""

    def __init__(self, length: int, rng: random.Random, unique: bool = False, rank: int = 0, world: int = 1):
        self.length, self.rng, self.unique, self.rank, self.world = length, rng, unique, rank, world
        self.pool = self._refill() if unique else None

    def _refill(self):
        return set(list(range(self.length))[self.rank::self.world])

    def __call__(self, item: int) -> int:
        if self.pool is not None:
            if not self.pool:
                self.pool = self._refill()
            return self.pool.pop()
        if item >= self.length:
            return self.rng.randint(0, self.length - 1)
        return item


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
