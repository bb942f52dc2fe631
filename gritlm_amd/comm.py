"""Cross-rank gather of the pooled representations on RCCL through the C ABI (``grit_comm_*``, include/gritlm_hip.h).

Replaces ``DistributedContrastiveLoss._dist_gather_tensor`` (gritlm/training/model.py:49-60) below ``torch.distributed``: one grouped
``ncclAllGather`` of the query and the passage rows straight into the rank-major matrices the loss kernel reads, issued on a side stream
(optionally CU-masked, ``cu_mask`` compute units) that is ordered against torch's current stream with events -- the caller's compute
stream never blocks on the collective until it calls ``wait()`` on the returned handle.

The process group that ``torch.distributed`` already has is used once, to hand rank 0's ``ncclUniqueId`` to the other ranks.
``GRIT_NATIVE_COMM=1`` makes the GradCache step (training/gradcache.py::ChunkGather) and the loss use this path instead of
``dist.all_gather_into_tensor``; it is exercised on the GPU with a one-rank communicator (tests/gpu_checks.py::check_native_comm)
-- the build box has one GPU -- and is therefore opt-in until a multi-GPU run has been seen."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


def enabled() -> bool:
    return os.environ.get("GRIT_NATIVE_COMM") == "1"


class GatherHandle:
    """Result of an asynchronous gather: ``wait()`` orders torch's current stream behind the collective and returns the tensors."""

    def __init__(self, q_all, p_all, event, keep, device=None):
        self.q_all, self.p_all, self._event, self._keep = q_all, p_all, event, keep
        self._device = device if device is not None else (q_all if q_all is not None else p_all).device

    def wait(self):
        """Order the COLLECTIVE'S DEVICE's current stream behind the gather (not whatever device happens to be current)."""
        if self._event is not None:
            torch.cuda.current_stream(self._device).wait_event(self._event)
            self._event = None
        self._keep = None
        return self.q_all, self.p_all

    def __del__(self):
        # A handle dropped without wait() (an exception between add() and finish()) must not hand its buffers back to the caching
        # allocator while the collective still reads / writes them on the side stream: the outputs were allocated, and the inputs produced,
        # on the compute stream, so ordering that stream behind the collective makes every later reuse of the memory safe.
        try:
            if self._event is not None:
                torch.cuda.current_stream(self._device).wait_event(self._event)
        except Exception:  # noqa: BLE001  -- interpreter shutdown
            pass


class NativeComm:
    _instances: dict = {}

    def __init__(self, device, cu_mask: int = 0):
        if not (dist.is_available() and dist.is_initialized()):
            raise ValueError("NativeComm needs an initialised torch.distributed process group (it carries the unique id to the ranks)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if self.rank == 0:
            check(self.lib.grit_comm_unique_id(buf), "grit_comm_unique_id")
        box = [buf.raw]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            check(self.lib.grit_comm_init(C.c_char_p(box[0]), self.world, self.rank, C.byref(h)), "grit_comm_init")
            self.handle = h
            self._raw_stream = None
            if cu_mask > 0:
                s = C.c_void_p()
                check(self.lib.grit_stream_create_cu_mask(int(cu_mask), C.byref(s)), "grit_stream_create_cu_mask")
                self._raw_stream = s
                self.stream = torch.cuda.ExternalStream(s.value, device=self.device)
            else:
                self.stream = torch.cuda.Stream(device=self.device)

    @classmethod
    def get(cls, device, cu_mask: int | None = None) -> "NativeComm":
        key = (torch.device(device).index, os.getpid())
        if key not in cls._instances:
            cls._instances[key] = cls(device, int(os.environ.get("GRIT_COMM_CU_MASK", "0")) if cu_mask is None else cu_mask)
        return cls._instances[key]

    def allgather_packed(self, q: torch.Tensor | None, p: torch.Tensor | None) -> GatherHandle:
        """q [nq, H], p [np, H] fp32 (either may be None) -> handle of (q_all [W * nq, H], p_all [W * np, H]), rank-major."""
        ref = q if q is not None else p
        H = ref.shape[1]
        mk = lambda t: None if t is None else torch.empty((self.world * t.shape[0], H), dtype=torch.float32, device=self.device)
        q = None if q is None else q.detach().float().contiguous()
        p = None if p is None else p.detach().float().contiguous()
        q_all, p_all = mk(q), mk(p)
        ptr = lambda t: 0 if t is None else t.data_ptr()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))          # the inputs are produced on the compute stream
        with torch.cuda.device(self.device):
            check(self.lib.grit_comm_allgather_packed(self.handle, ptr(q), 0 if q is None else q.shape[0], ptr(p), 0 if p is None else p.shape[0],
                                                      H, ptr(q_all), ptr(p_all), self.stream.cuda_stream), "grit_comm_allgather_packed")
        ev = torch.cuda.Event()
        ev.record(self.stream)
        # no record_stream(): the handle keeps the inputs alive until wait() has ordered the compute stream behind the collective, after
        # which the allocator may reuse them in compute-stream order (and a CU-masked stream can then be destroyed without the caching
        # allocator still holding it)
        return GatherHandle(q_all, p_all, ev, (q, p), self.device)

    def close(self):
        if getattr(self, "handle", None) is not None:
            torch.cuda.synchronize(self.device)
            self.lib.grit_comm_destroy(self.handle)
            self.handle = None
        if getattr(self, "_raw_stream", None) is not None:
            self.lib.grit_stream_destroy(self._raw_stream)
            self._raw_stream = None
        key = (self.device.index, os.getpid())
        if self._instances.get(key) is self:
            self._instances.pop(key)
