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


def _rows(chunk: Dict) -> int:
    v = next(iter(chunk.values()))
    return v.shape[0] if isinstance(v, torch.Tensor) else len(v)


class OverlappedGradSync:
    """Data-parallel gradient averaging overlapped with the backward of the LAST GradCache chunk (what DDP does when ``no_sync`` is
    lifted on the last chunk, grad_cache.py:230-236): the native engine reports each layer's packed gradient buffers as soon as
    they are final and their all-reduce (RCCL, own stream) runs under the remaining layers' backward kernels."""

    def __init__(self, model):
        self.model, self.pending = model, []
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 \
            and getattr(model, "train_engine", None) is not None

    def arm(self):
        if self.active:
            self.model._on_layer_done = lambda bufs: self.pending.extend((b, dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True)) for b in bufs)

    def finish(self):
        if not self.active:
            sync_gradients(self.model)
            return
        self.model._on_layer_done = None
        w = dist.get_world_size()
        small = self.model.train_engine.grad_buffers(small_only=True)
        self.pending.extend((b, dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True)) for b in small)
        for b, h in self.pending:
            h.wait()
            b.div_(w)
        self.pending.clear()


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


class ChunkGather:
    """Cross-rank gather of pooled representations, issued chunk by chunk as pass 1 produces them.

    Every chunk's all-gather is asynchronous (RCCL runs it on its own stream over xGMI) and lands directly in the rank-major
    [W, n_local, H] buffer, so the exchange overlaps the forward of the following chunks -- the query tower's gather
    overlaps the whole document tower, and only the last passage chunk's gather is exposed.  Replaces the two blocking
    list-API gathers + torch.cat of ``_dist_gather_tensor`` (gritlm/training/model.py:49-60); result order is identical."""

    def __init__(self, n_local: int, width: int, dtype, device):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.buf = torch.empty((self.world, n_local, width), dtype=dtype, device=device)
        self.handles, self.keep, self.row = [], [], 0

    def add(self, reps: torch.Tensor):
        reps = reps.detach().contiguous()
        n = reps.shape[0]
        views = [self.buf[r, self.row:self.row + n] for r in range(self.world)]
        self.handles.append(dist.all_gather(views, reps, async_op=True))
        self.keep.append(reps)
        self.row += n

    def finish(self) -> torch.Tensor:
        for h in self.handles:
            h.wait()
        self.keep.clear()
        return self.buf.view(self.world * self.buf.shape[1], self.buf.shape[2])


class GradCacheStep:
    def __init__(self, model, chunk_size: int):
        self.model = model
        self.chunk_size = int(chunk_size)

    @torch.no_grad()
    def _reps_no_grad(self, chunks, gather: "ChunkGather | None" = None):
        out = []
        for c in chunks:
            r = self.model.encode(c)
            if gather is not None:
                gather.add(r)
            out.append(r)
        return torch.cat(out, dim=0)

    def __call__(self, query: Dict, passage: Dict, sync: bool = True) -> torch.Tensor:
        model = self.model
        q_chunks, p_chunks = split_inputs(query, self.chunk_size), split_inputs(passage, self.chunk_size)
        loss_fn = model.emb_loss_fn
        cross = bool(getattr(loss_fn, "negatives_cross_device", False))
        nq = sum(_rows(c) for c in q_chunks); npas = sum(_rows(c) for c in p_chunks)
        # pass 1 (the cross-rank exchange rides along, chunk by chunk)
        gq = gp = None
        q_list, p_list = [], []
        with torch.no_grad():
            for chunks, lst, which in ((q_chunks, q_list, "q"), (p_chunks, p_list, "p")):
                for c in chunks:
                    r = model.encode(c)
                    if cross:
                        if which == "q" and gq is None:
                            gq = ChunkGather(nq, r.shape[1], r.dtype, r.device)
                        if which == "p" and gp is None:
                            gp = ChunkGather(npas, r.shape[1], r.dtype, r.device)
                        (gq if which == "q" else gp).add(r)
                    lst.append(r)
        q_reps, p_reps = torch.cat(q_list, dim=0), torch.cat(p_list, dim=0)
        # loss + representation-gradient cache
        q_leaf, p_leaf = q_reps.detach().requires_grad_(), p_reps.detach().requires_grad_()
        if cross:
            loss = loss_fn.with_gathered(q_leaf, p_leaf, gq.finish(), gp.finish())
        else:
            loss = loss_fn(q_leaf, p_leaf)
        loss.backward()
        caches = (q_leaf.grad, p_leaf.grad)
        # pass 2 (the gradient all-reduce of the data-parallel ranks rides under the last chunk's backward)
        gsync = OverlappedGradSync(model) if sync else None
        work = [(c, caches[0], "q") for c in q_chunks] + [(c, caches[1], "p") for c in p_chunks]
        rows = {"q": 0, "p": 0}
        for idx, (c, cache, which) in enumerate(work):
            if gsync is not None and idx == len(work) - 1:
                gsync.arm()
            reps = model.encode(c)
            r0 = rows[which]
            reps.backward(gradient=cache[r0:r0 + reps.shape[0]].to(reps.dtype))
            rows[which] = r0 + reps.shape[0]
        if gsync is not None:
            gsync.finish()
        return loss.detach()
