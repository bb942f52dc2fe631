"""Device-side pieces of the reference's RAG flow (rag/index.py, rag/eval.py) that sit next to the encode hot path: the exhaustive
inner-product index -- ``DistributedIndex`` with the reference's methods, attributes and on-disk format, its search on the native
kernel (``grit_knn_topk``) -- and the passage loader.  Document-KV caching and generation from the cache live in
``GritLM.encode(get_cache=True)`` and ``gritlm_amd.decoder.MistralDecoder``."""
from __future__ import annotations

import json
import math
import os
import pickle
from typing import Optional

import torch

from . import ops


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class DenseIndex:
    """The search half of rag/index.py's ``DistributedIndex`` for one process: ``embeddings`` is the [H, N] matrix it builds
    (index.py:141: ``index.embeddings[:, total:total+len] = embeddings.T``) kept in fp32 on the device."""

    def __init__(self, embeddings_hn: torch.Tensor):
        if embeddings_hn.dim() != 2:
            raise ValueError("DenseIndex: [H, N] embedding matrix expected")
        self.embeddings = embeddings_hn.to(torch.float32)

    @classmethod
    def from_rows(cls, embeddings_nh: torch.Tensor):
        """From encode() output [N, H] (no copy: the kernel takes strides)."""
        return cls(embeddings_nh.to(torch.float32).t())

    @torch.no_grad()
    def search_knn(self, queries: torch.Tensor, topk: int):
        """``_compute_scores_and_indices`` (index.py:97-104): (scores [Q, topk] descending, indices [Q, topk])."""
        q = queries.to(device=self.embeddings.device, dtype=torch.float32).contiguous()
        return ops.knn_topk(q, self.embeddings, topk, transposed=True)


class DistributedIndex:
    """rag/index.py:20-147 behind the same names: ``embeddings`` [dim, n_passages] (the caller fills columns in place, rag/eval.py:145),
    ``doc_map`` {column -> passage dict}, ``dtype``, ``is_in_gpu``; ``init_embeddings`` / ``save_index`` / ``load_index`` /
    ``search_knn`` / ``is_index_trained``.  Files are the reference's (``embeddings.<shard>.pt`` = ``torch.save`` of the [dim, n]
    column block, ``passages.<shard>.pt`` = pickled list of passage dicts): an index saved by either side loads on the other.

    The scores of ``search_knn`` come from ``grit_knn_topk`` (exact-f32 MFMA similarity + chunked bitonic top-k; equal scores are
    returned lower column first, where ``torch.topk`` promises nothing) on the [dim, N] layout as it lies.  The kernel computes in fp32: an
    index of another ``dtype`` is widened for the search (the reference multiplies in the index dtype).

    With an initialised process group every rank holds its shard of the passages (``load_passages`` deals them round-robin) and calls
    ``search_knn`` with ITS queries: the queries of all ranks are gathered, every rank searches its shard, and each rank receives the
    candidates for its own queries from all shards and keeps the best ``topk`` -- what index.py:107-134 describes (its
    ``serialize_listdocs`` / ``deserialize_listdocs`` helpers are not part of the reference repository; passages travel as Python
    objects here)."""

    def __init__(self, dtype=torch.float32):
        self.embeddings = None
        self.doc_map = dict()
        self.is_in_gpu = bool(torch.cuda.is_available())
        self.dtype = dtype

    # ------------------------------------------------------------------ construction
    def init_embeddings(self, passages, dim: Optional[int]):
        self.doc_map = {i: doc for i, doc in enumerate(passages)}
        self.embeddings = torch.zeros((dim, len(passages)), dtype=self.dtype, device="cuda" if self.is_in_gpu else "cpu")

    # ------------------------------------------------------------------ persistence (the reference's shard files)
    @staticmethod
    def _get_saved_embedding_path(save_dir: str, shard: int) -> str:
        return os.path.join(save_dir, f"embeddings.{shard}.pt")

    @staticmethod
    def _get_saved_passages_path(save_dir: str, shard: int) -> str:
        return os.path.join(save_dir, f"passages.{shard}.pt")

    @staticmethod
    def _shards_of(total_saved_shards: int):
        rank, world = _rank_world()
        if total_saved_shards % world != 0:
            raise AssertionError("N workers must be a multiple of shards to save")
        per_worker = total_saved_shards // world
        return range(rank * per_worker, (rank + 1) * per_worker), per_worker

    def save_index(self, path: str, total_saved_shards: int, overwrite_saved_passages: bool = False) -> None:
        """This rank's columns as ``total_saved_shards / world`` consecutive blocks; the embeddings are always rewritten, the passage
        lists only when missing or ``overwrite_saved_passages``."""
        if self.embeddings is None:
            raise AssertionError("save_index: no embeddings")
        n = self.embeddings.shape[1]
        if n != len(self.doc_map):
            raise AssertionError(len(self.doc_map))
        shard_ids, per_worker = self._shards_of(total_saved_shards)
        width = math.ceil(n / per_worker) if n else 0
        os.makedirs(path, exist_ok=True)
        for shard, start in zip(shard_ids, range(0, n, width) if width else ()):
            stop = min(start + width, n)
            ppath = self._get_saved_passages_path(path, shard)
            if overwrite_saved_passages or not os.path.exists(ppath):
                with open(ppath, "wb") as f:
                    pickle.dump([self.doc_map[i] for i in range(start, stop)], f, protocol=pickle.HIGHEST_PROTOCOL)
            # .clone(): torch.save of a column VIEW writes the whole underlying storage into every shard file
            torch.save(self.embeddings[:, start:stop].clone(), self._get_saved_embedding_path(path, shard))

    def load_index(self, path: str, total_saved_shards: int):
        """This rank's ``total_saved_shards / world`` shard files, in shard order (no index structure: the matrix IS the index)."""
        shard_ids, _ = self._shards_of(total_saved_shards)
        blocks, self.doc_map = [], {}
        for shard in shard_ids:
            with open(self._get_saved_passages_path(path, shard), "rb") as f:
                for p in pickle.load(f):
                    self.doc_map[len(self.doc_map)] = p
            e = torch.load(self._get_saved_embedding_path(path, shard), map_location="cpu")
            blocks.append(e.cuda() if self.is_in_gpu else e)
        self.embeddings = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=1)

    # ------------------------------------------------------------------ search
    def _compute_scores_and_indices(self, allqueries: torch.Tensor, topk: int):
        """index.py:97-104 (``queries @ embeddings`` -> ``torch.topk``) on the native kernel: scores [Q, topk] fp32 descending and
        the column of each.  ``topk`` larger than the shard returns the whole shard."""
        emb = self.embeddings
        if emb.dtype != torch.float32:
            # the kernel takes fp32: widen a bf16 / fp16 index ONCE and keep the copy until the index tensor is replaced or written to
            # (a fresh full-size copy per search_knn call tripled the footprint and re-read the whole index per query batch)
            key = (emb.data_ptr(), tuple(emb.shape), emb.dtype, emb._version)
            if getattr(self, "_emb32_key", None) != key:
                self._emb32, self._emb32_key = emb.to(torch.float32), key
            emb = self._emb32
        q = allqueries.to(device=emb.device, dtype=torch.float32).contiguous()
        return ops.knn_topk(q, emb, min(topk, emb.shape[1]), transposed=True)

    @torch.no_grad()
    def search_knn(self, queries: torch.Tensor, topk: int):
        """Exhaustive inner-product search: ``(docs, scores)``, ``docs[i]`` = the ``topk`` passage dicts for query i (best first),
        ``scores[i]`` their scores as Python floats."""
        rank, world = _rank_world()
        if world == 1:
            scores, cols = self._compute_scores_and_indices(queries, topk)
            return [[self.doc_map[c] for c in row] for row in cols.tolist()], scores.tolist()
        import torch.distributed as dist
        every = [None] * world
        dist.all_gather_object(every, queries.detach().to("cpu", torch.float32))
        bounds = [0]
        for q in every:
            bounds.append(bounds[-1] + q.shape[0])
        scores, cols = self._compute_scores_and_indices(torch.cat(every, dim=0), topk)
        scores, cols = scores.tolist(), cols.tolist()
        # candidates of THIS shard for the queries of rank k, sent to rank k
        outgoing = [([[self.doc_map[c] for c in row] for row in cols[bounds[k]:bounds[k + 1]]], scores[bounds[k]:bounds[k + 1]])
                    for k in range(world)]
        incoming = [None] * world
        dist.all_gather_object(incoming, outgoing)
        docs_out, scores_out = [], []
        for i in range(queries.shape[0]):
            cand = [(s, r, j) for r in range(world) for j, s in enumerate(incoming[r][rank][1][i])]
            cand.sort(key=lambda t: (-t[0], t[1], t[2]))          # best first; ties: lower shard, then the shard's own order
            cand = cand[:topk]
            docs_out.append([incoming[r][rank][0][i][j] for _, r, j in cand])
            scores_out.append([s for s, _, _ in cand])
        return docs_out, scores_out

    def is_index_trained(self) -> bool:
        return True


def load_passages(filenames, maxload: int = -1):
    """rag/index.py:150-193: the lines of the JSONL files are numbered through all files and line i belongs to rank ``i % world``;
    ``maxload`` (> -1) stops after that many lines overall; a passage with a non-empty ``section`` gets ``title = "title: section"``;
    an empty line of this rank's share yields ``None`` (the reference prints "empty line" and appends None)."""
    rank, world = _rank_world()
    passages, counter = [], 0
    for fname in filenames:
        with open(fname) as f:
            for line in f:
                if -1 < maxload <= counter:
                    break
                if counter % world == rank:
                    item = None
                    if line.strip() != "":
                        item = json.loads(line)
                        if "title" in item and "section" in item and len(item["section"]) > 0:
                            item["title"] = f"{item['title']}: {item['section']}"
                    else:
                        print("empty line")
                    passages.append(item)
                counter += 1
    return passages


def load_or_initialize_index(args, logger, dim: int, dtypes=None):
    """rag/index.py:195-218: the index of ``args.load_index_path`` (with its passages) or an empty one sized for ``args.passages``
    (optionally ``args.limit_start:args.limit``; ``args.customd`` = one passage read from a file, or ``"<s>" * int(customd)``)."""
    table = dtypes or {"bfloat16": torch.bfloat16, "float32": torch.float32, "float16": torch.float16}
    index = DistributedIndex(dtype=table[args.idxdtype])
    if args.load_index_path is not None:
        logger.info(f"Loading index from: {args.load_index_path}")
        index.load_index(args.load_index_path, args.save_index_n_shards)
        passages = [index.doc_map[i] for i in range(len(index.doc_map))]
        return index, passages
    logger.info(f"Loading passages from: {args.passages}")
    passages = load_passages(args.passages)
    logger.info(f"Loaded {len(passages)} passages")
    if getattr(args, "limit", None) is not None:
        passages = passages[args.limit_start:args.limit]
        logger.info(f"Limiting to {len(passages)} passages ({args.limit_start}-{args.limit})")
    customd = getattr(args, "customd", None)
    if customd:
        if os.path.exists(customd):
            with open(customd) as f:
                passages = [{"text": f.read(), "title": ""}]
        else:
            passages = [{"text": "<s>" * int(customd), "title": ""}]
    logger.info(f"Example passage: {passages[0]}")
    index.init_embeddings(passages, dim)
    return index, passages
