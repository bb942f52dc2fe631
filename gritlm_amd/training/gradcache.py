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

import os
from typing import Dict, List

import torch
import torch.distributed as dist

from .._lib import GritHipError


def dist_active() -> bool:
    """Data-parallel collectives needed?  True for world_size > 1; GRIT_DIST_WORLD1=1 keeps every collective in the step on a
    ONE-rank process group too (test hook: the RCCL calls are then issued for real on a single GPU, tests/gpu_checks.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GRIT_DIST_WORLD1") == "1"


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


def _all_reduce_mean(buf: torch.Tensor):
    """Asynchronous all-reduce that leaves the MEAN over ranks in ``buf``.  RCCL averages inside the collective
    (``ReduceOp.AVG``: no second pass over the 14.5 GB of gradients after the exchange); gloo has no AVG, so the CPU test
    path sums and divides.  Returns (work handle, divisor to apply after ``wait()`` or None)."""
    if dist.get_backend() == "nccl":
        return dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True), None
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), dist.get_world_size()


def _finish_mean(pending) -> None:
    for buf, (work, div) in pending:
        work.wait()
        if div is not None:
            buf.div_(div)


class OverlappedGradSync:
    """Data-parallel gradient averaging overlapped with the backward of the LAST GradCache chunk (what DDP does when ``no_sync`` is
    lifted on the last chunk, grad_cache.py:230-236): the native engine reports each layer's packed gradient buffers as soon as
    they are final and their averaging all-reduce (RCCL ``ncclAvg``, own stream) runs under the remaining layers' backward kernels;
    nothing but the waits is left on the critical path after the last chunk."""

    def __init__(self, model):
        self.model, self.pending = model, []
        self.active = dist_active() and getattr(model, "train_engine", None) is not None

    def arm(self):
        if self.active:
            self.model._on_layer_done = lambda bufs: self.pending.extend((b, _all_reduce_mean(b)) for b in bufs)

    def finish(self):
        if not self.active:
            sync_gradients(self.model)
            return
        self.model._on_layer_done = None
        small = self.model.train_engine.grad_buffers(small_only=True)
        self.pending.extend((b, _all_reduce_mean(b)) for b in small)
        _finish_mean(self.pending)
        self.pending.clear()


def sync_gradients(model) -> None:
    """Average gradients over ranks (DDP semantics: the effective gradient is (1/W) * grad of the global loss)."""
    if not dist_active():
        return
    eng = getattr(model, "train_engine", None)
    bufs = eng.grad_buffers() if eng is not None else [p.grad for p in model.parameters() if p.grad is not None]
    _finish_mean([(b, _all_reduce_mean(b)) for b in bufs])


class ChunkGather:
    """Cross-rank gather of pooled representations, issued chunk by chunk as pass 1 produces them.

    Every chunk is ONE asynchronous ``all_gather_into_tensor`` (RCCL ``ncclAllGather`` straight into a contiguous [W, n_chunk, H]
    slab: no list API, no staging copies; runs on RCCL's own stream over xGMI), so the exchange overlaps the forward of the
    following chunks -- the query tower's gather overlaps the whole document tower, and only the last passage chunk's gather is
    exposed.  ``finish`` stitches the slabs into the rank-major [W * n_local, H] matrix with one copy.  Replaces the two blocking
    list-API gathers + torch.cat of ``_dist_gather_tensor`` (gritlm/training/model.py:49-60); result order is identical (rank r owns
    rows [r * n_local, (r + 1) * n_local), the order the targets ``arange(B) * G`` rely on, :45-46)."""

    def __init__(self, n_local: int, width: int, dtype, device, group_rows: "int | None" = None):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.n_local, self.width, self.dtype, self.device = n_local, width, dtype, device
        self.items = []            # (slab [W, n, H], work handle, source kept alive until the collective completed)
        self.calls = 0
        # The collective schedule must be the SAME on every rank, whatever each rank's model calls looked like: a rank whose batch was
        # padded to a longer sequence may run pass 1 in smaller calls (GradCacheStep._pass1_rows caps a call by tokens).  With
        # ``group_rows`` set, rows are gathered in groups of exactly that many (a configuration constant) plus one remainder group at
        # ``flush`` -- a function of n_local alone, which all ranks share (all_gather needs equal shapes anyway).
        self.group_rows = int(group_rows) if group_rows else 0
        self.pending, self.pending_rows = [], 0
        # GRIT_NATIVE_COMM=1: the same gathers on RCCL directly through the C ABI (grit_comm_allgather_packed, gritlm_amd/comm.py) on a
        # side stream ordered with events, instead of torch.distributed
        self.native = None
        if torch.device(device).type == "cuda" and dtype == torch.float32:
            from .. import comm as _comm
            if _comm.enabled():
                self.native = _comm.NativeComm.get(device)

    def add(self, reps: torch.Tensor):
        reps = reps.detach()
        if not self.group_rows:
            return self._gather(reps.contiguous())
        if not self.pending and reps.shape[0] == self.group_rows:        # the usual case: one call = one group, no copy
            return self._gather(reps.contiguous())
        self.pending.append(reps)
        self.pending_rows += reps.shape[0]
        while self.pending_rows >= self.group_rows:
            self._gather(self._take(self.group_rows))

    def flush(self):
        """The remainder group (fewer than ``group_rows`` rows), once the tower's last call has been added."""
        if self.pending_rows:
            self._gather(self._take(self.pending_rows))

    def _take(self, n: int) -> torch.Tensor:
        buf = self.pending[0] if len(self.pending) == 1 else torch.cat(self.pending, dim=0)
        head, tail = buf[:n].contiguous(), buf[n:]
        self.pending = [tail] if tail.shape[0] else []
        self.pending_rows = int(tail.shape[0])
        return head

    def _gather(self, reps: torch.Tensor):
        n = reps.shape[0]
        self.calls += 1
        if self.native is not None:
            h = self.native.allgather_packed(reps, None)
            self.items.append((h.q_all.view(self.world, n, self.width), h, reps))
            return
        slab = torch.empty((self.world, n, self.width), dtype=self.dtype, device=self.device)
        work = dist.all_gather_into_tensor(slab.view(self.world * n, self.width), reps, async_op=True)
        self.items.append((slab, work, reps))

    def finish(self) -> torch.Tensor:
        self.flush()
        for _, work, _ in self.items:
            work.wait()
        slabs = [s for s, _, _ in self.items]
        out = slabs[0] if len(slabs) == 1 else torch.cat(slabs, dim=1)
        assert out.shape[1] == self.n_local, (out.shape, self.n_local)
        self.items.clear()
        return out.reshape(self.world * self.n_local, self.width)


class _Span:
    """CUDA-event bracket on the current stream (bench.py's exposed-communication / loss timing); no-op on CPU tensors."""

    def __init__(self, sink: "dict | None", key: str, enabled: bool):
        self.sink, self.key, self.on = sink, key, bool(enabled and sink is not None)

    def __enter__(self):
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            self.sink.setdefault("_events", []).append((self.key, self.a, self.b))
        return False


def pass1_chunk_rows(model, chunk_size: int) -> int:
    """Rows per model call in PASS 1.  The reference uses ``chunk_size`` for both passes because the chunk is a memory knob of
    the activation-keeping pass 2.  Pass 1 keeps nothing, and on the native engine a row's representation does not depend on
    which rows share its launch (every kernel is batch-invariant: ``packed_encode*`` / ``train_packed_vs_padded`` compare
    bit for bit), so pass 1 may run ``GRIT_GRADCACHE_PASS1_MULT`` (default 4) chunks per call: the same bits from 4x fewer,
    4x larger launches (GEMMs at M = 65536 instead of 16384 for 32 x 512-token chunks, 18 instead of 72 gathers at
    BASELINE configs[2]).  The Hugging Face path keeps the reference's behaviour (CPU BLAS is not batch-invariant)."""
    eng = getattr(model, "train_engine", None)
    if eng is None or getattr(model, "projection", None) is not None:
        return int(chunk_size)
    try:
        mult = int(os.environ.get("GRIT_GRADCACHE_PASS1_MULT", "4"))
    except ValueError:
        mult = 4
    return int(chunk_size) * max(1, mult)


class GradCacheStep:
    def __init__(self, model, chunk_size: int, pass1_chunk_size: "int | None" = None, precision: "str | None" = None):
        """``precision`` (native engine only; "bf16" | "f16_operands" | "f16_stream"; default: GRIT_PASS1_PRECISION or the engine's current
        setting): the policy of PASS 1, the no-grad forward that defines the representations and the loss.  Under the fp16 policies the
        loss matches the reference's fp32 loss to 1e-3 at depth 32 (bf16: ~1e-2 at tau 0.02); pass 2 re-encodes in the reference's bf16
        arithmetic either way (its saved activations feed the bf16 backward kernels), so the cached representation gradients meet a forward
        whose representations differ from pass 1's by the bf16 policy's own error (1 - cos ~ 5e-4 at depth 32): measured in
        tests/gpu_checks.py::check_gradcache_f16_pass1 and bench.py's contrastive `parity` object."""
        self.model = model
        eng = getattr(model, "train_engine", None)
        explicit = precision is not None
        precision = precision or os.environ.get("GRIT_PASS1_PRECISION")
        if precision and eng is not None:
            try:
                eng.set_nograd_precision(precision)
            except GritHipError:
                # a process-wide default (the environment variable) that this model kind does not take -- the sparse-MoE engine routes on
                # the fp32 residual stream -- lands on the fp16 policy it does take; an explicit argument is refused as it stands
                if explicit or precision != "f16_stream":
                    raise
                print(f"GradCacheStep: GRIT_PASS1_PRECISION={precision} is not built for this model kind; pass 1 runs under 'f16_operands'",
                      flush=True)
                eng.set_nograd_precision("f16_operands")
        self.precision = getattr(eng, "nograd_precision", "bf16") if eng is not None else "bf16"
        self.chunk_size = int(chunk_size)
        self.pass1_chunk_size = int(pass1_chunk_size) if pass1_chunk_size else pass1_chunk_rows(model, self.chunk_size)
        self.profile = None        # set to a dict to collect per-step timings (ms): loss (similarity GEMM + CE + rep grads), the
                                   # stream time blocked on the rep gather (exposed, non-overlapped part) and on the gradient all-reduce

    def _pass1_rows(self, model_input: Dict) -> int:
        """Rows per pass-1 call for this tower: ``pass1_chunk_size``, capped so that one call stays at or below
        ``GRIT_GRADCACHE_PASS1_TOKENS`` padded tokens (default 65536 = 128 rows x 512) -- the no-grad forward's scratch activations grow
        with the call (4.3 GB at 65536 tokens of the 7B shape), and a long-sequence tower (2048-token passages) must not turn the
        reference's memory knob into a 4x larger allocation -- in whole multiples of the pass-2 chunk and never below it."""
        ids = model_input.get("input_ids") if isinstance(model_input, dict) else None
        seq = int(ids.shape[1]) if isinstance(ids, torch.Tensor) and ids.dim() == 2 else 0
        rows = self.pass1_chunk_size
        if seq > 0:
            try:
                cap_tokens = int(os.environ.get("GRIT_GRADCACHE_PASS1_TOKENS", "65536"))
            except ValueError:
                cap_tokens = 65536
            cap_rows = (cap_tokens // seq) // self.chunk_size * self.chunk_size
            rows = min(rows, max(self.chunk_size, cap_rows))
        return max(self.chunk_size, rows)

    def profile_summary(self) -> dict:
        """Resolve the recorded CUDA events into {key: total ms} (call after torch.cuda.synchronize())."""
        out = {}
        if self.profile:
            for key, a, b in self.profile.pop("_events", []):
                out[key] = out.get(key, 0.0) + a.elapsed_time(b)
            self.profile.update(out)
        return dict(self.profile or {})

    def __call__(self, query: Dict, passage: Dict, sync: bool = True) -> torch.Tensor:
        model = self.model
        q_chunks, p_chunks = split_inputs(query, self.chunk_size), split_inputs(passage, self.chunk_size)
        loss_fn = model.emb_loss_fn
        cross = bool(getattr(loss_fn, "negatives_cross_device", False))
        nq = sum(_rows(c) for c in q_chunks); npas = sum(_rows(c) for c in p_chunks)
        # pass 1 (the cross-rank exchange rides along, chunk by chunk); its calls may span several pass-2 chunks (pass1_chunk_rows)
        if self.pass1_chunk_size != self.chunk_size:
            q1_chunks = split_inputs(query, self._pass1_rows(query))
            p1_chunks = split_inputs(passage, self._pass1_rows(passage))
        else:
            q1_chunks, p1_chunks = q_chunks, p_chunks
        gq = gp = None
        q_list, p_list = [], []
        with torch.no_grad():
            for chunks, lst, which in ((q1_chunks, q_list, "q"), (p1_chunks, p_list, "p")):
                for c in chunks:
                    r = model.encode(c)
                    if cross:
                        # groups of pass1_chunk_size rows: the same collectives on every rank even when _pass1_rows differs between
                        # ranks (batches padded to different lengths)
                        if which == "q" and gq is None:
                            gq = ChunkGather(nq, r.shape[1], r.dtype, r.device, self.pass1_chunk_size)
                        if which == "p" and gp is None:
                            gp = ChunkGather(npas, r.shape[1], r.dtype, r.device, self.pass1_chunk_size)
                        (gq if which == "q" else gp).add(r)
                    lst.append(r)
                if cross and (gq if which == "q" else gp) is not None:
                    (gq if which == "q" else gp).flush()         # the tower's remainder goes out before the next tower starts
        q_reps, p_reps = torch.cat(q_list, dim=0), torch.cat(p_list, dim=0)
        self.last_reps = (q_reps, p_reps)          # (pass 1's representations: parity probes read them)
        # loss + representation-gradient cache
        q_leaf, p_leaf = q_reps.detach().requires_grad_(), p_reps.detach().requires_grad_()
        gpu = q_reps.is_cuda
        if cross:
            with _Span(self.profile, "exposed_gather_ms", gpu):
                q_all, p_all = gq.finish(), gp.finish()
            if self.profile is not None:
                self.profile["gather_collectives"] = gq.calls + gp.calls
                self.profile["gather_bytes_received"] = (q_all.numel() + p_all.numel()) * q_all.element_size() * (gq.world - 1) // gq.world
            with _Span(self.profile, "loss_fwd_bwd_ms", gpu):
                loss = loss_fn.with_gathered(q_leaf, p_leaf, q_all, p_all)
                loss.backward()
        else:
            with _Span(self.profile, "loss_fwd_bwd_ms", gpu):
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
            with _Span(self.profile, "exposed_grad_allreduce_ms", gpu):
                gsync.finish()
        return loss.detach()
