"""GradCache step for the contrastive objective -- the step structure of the vendored luyug/GradCache
(gritlm/training/GradCache/src/grad_cache/grad_cache.py:244-280) as driven by the reference trainer
(gritlm/training/gradcache_trainer.py:373-400, :691), re-hosted for the native engine:

  1. split query / passage batches into chunks of ``chunk_size`` rows             (split_inputs :72-102)
  2. PASS 1, no grad: representations of every chunk                               (forward_no_grad :169-191)
  3. loss on the full (cross-rank gathered) batch, d loss / d reps = the cache      (build_cache :193-211)
  4. PASS 2 per chunk: forward with activations kept, backward seeded with the cached rows; the reference's
     surrogate ``dot(reps, cached_grad).backward()`` (:241-242) IS "grad_output = cached_grad"
  5. data-parallel: one averaged all-reduce of the (few, large) gradient buffers after the last chunk -- what DDP
     does when ``no_sync`` is lifted on the last chunk (:230-236)
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.distributed as dist


def split_inputs(model_input: Dict, chunk_size: int) -> List[Dict]:
    """Dict of tensors (and per-row lists such as ``instruction_lens``) -> list of row-chunk dicts."""
    keys = list(model_input.keys())
    n = None
    for v in model_input.values():
        n = v.shape[0] if isinstance(v, torch.Tensor) else len(v)
        break
    out = []
    for s in range(0, n, chunk_size):
        out.append({k: model_input[k][s:s + chunk_size] for k in keys})
    return out


def sync_gradients(model) -> None:
    """Average gradients over ranks (DDP semantics: the effective gradient is (1/W) * grad of the global loss)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    eng = getattr(model, "train_engine", None)
    bufs = eng.grad_buffers() if eng is not None else [p.grad for p in model.parameters() if p.grad is not None]
    w = dist.get_world_size()
    handles = [dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True) for b in bufs]
    for h, b in zip(handles, bufs):
        h.wait()
        b.div_(w)


class GradCacheStep:
    def __init__(self, model, chunk_size: int):
        self.model = model
        self.chunk_size = int(chunk_size)

    @torch.no_grad()
    def _reps_no_grad(self, chunks):
        return torch.cat([self.model.encode(c) for c in chunks], dim=0)

    def __call__(self, query: Dict, passage: Dict, sync: bool = True) -> torch.Tensor:
        model = self.model
        q_chunks, p_chunks = split_inputs(query, self.chunk_size), split_inputs(passage, self.chunk_size)
        # pass 1
        q_reps, p_reps = self._reps_no_grad(q_chunks), self._reps_no_grad(p_chunks)
        # loss + representation-gradient cache
        q_leaf, p_leaf = q_reps.detach().requires_grad_(), p_reps.detach().requires_grad_()
        loss = model.emb_loss_fn(q_leaf, p_leaf)
        loss.backward()
        caches = (q_leaf.grad, p_leaf.grad)
        # pass 2
        for chunks, cache in ((q_chunks, caches[0]), (p_chunks, caches[1])):
            row = 0
            for c in chunks:
                reps = model.encode(c)
                reps.backward(gradient=cache[row:row + reps.shape[0]].to(reps.dtype))
                row += reps.shape[0]
        if sync:
            sync_gradients(model)
        return loss.detach()
