"""Device-side pieces of the reference's RAG flow (rag/index.py, rag/eval.py) that sit next to the encode hot path:
the exhaustive inner-product index search.  Document-KV caching and generation from the cache live in
``GritLM.encode(get_cache=True)`` and ``gritlm_amd.decoder.MistralDecoder``."""
from __future__ import annotations

import torch

from . import ops


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
