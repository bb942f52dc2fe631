"""Multi-GPU encode: one process per GPU (torchrun), documents sharded in contiguous slices, no data-path collective.

The reference parallelises embedding-mode inference with ``torch.nn.DataParallel`` inside one process
(gritlm/gritlm.py:71-75, :106-107: batch_size *= num_gpus).  Here every rank owns one MI355X and its own replica; documents are
independent, so the only communication is the final gather of the [n_docs, H] result (optional)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of rank ``rank``; the first n % world ranks get one extra document."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def encode_sharded(model, sentences: list[str], gather: bool = True, **encode_kwargs) -> np.ndarray:
    """Every rank encodes its slice with ``model.encode``; with ``gather`` all ranks return the full [len(sentences), H] array
    (rank order == document order), otherwise each rank returns only its slice."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model.encode(sentences, **encode_kwargs)
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(len(sentences), world, rank)
    encode_kwargs = dict(encode_kwargs, convert_to_tensor=False)
    local = model.encode(sentences[lo:hi], **encode_kwargs) if hi > lo else None
    if not gather:
        return local
    width = torch.tensor([0 if local is None else local.shape[1]], dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    width = width.to(dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    H = int(width.item())
    sizes = [shard_bounds(len(sentences), world, r) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    buf = torch.zeros((cap, H), dtype=torch.float32, device=dev)
    if local is not None:
        buf[: hi - lo] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32)).to(dev)
    out = torch.empty((world * cap, H), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, cap, H).cpu().numpy()
    return np.concatenate([out[r, : b - a] for r, (a, b) in enumerate(sizes)], axis=0)
