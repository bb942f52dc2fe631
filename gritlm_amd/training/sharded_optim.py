"""AdamW with the optimizer state sharded over the data-parallel ranks (ZeRO stage 1).

The reference trains its 8x7B configuration under FSDP (scripts/configs/config_*_m8x7.yml through accelerate): parameters, gradients
and optimizer state sharded by torch's FSDP wrapper.  The native engine keeps whole bf16 replicas of parameters and gradients (its
kernels read packed weight matrices in place) and shards what dominates the footprint of a replica -- AdamW's two fp32 moments, 8 bytes
per parameter against 2 + 2 for the bf16 weight and gradient: at the Mixtral-8x7B shape 93 GB + 93 GB + 374 GB does not fit one
288 GB GPU, 93 + 93 + 374 / 8 = 233 GB does.

Every parameter has ONE owner rank (whole tensors, dealt largest-first to the least loaded rank: the same table on every rank).
After the gradient average -- unchanged: every rank holds the averaged gradient of every parameter -- a rank runs AdamW on the
parameters it owns and broadcasts their new values; the others receive them.  AdamW is element-wise, so the result is bit-identical to
one AdamW over all parameters on every rank (tests/test_sharded_optim.py compares against exactly that under 2 gloo ranks).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def partition(params, world: int):
    """owner[i] for ``params[i]``: largest tensors first, each to the rank holding the fewest elements so far (ties: lowest rank)."""
    load = [0] * world
    owner = [0] * len(params)
    for i in sorted(range(len(params)), key=lambda j: (-params[j].numel(), j)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += params[i].numel()
    return owner


class ShardedAdamW:
    """``torch.optim.AdamW`` over the parameters this rank owns + a broadcast of the updated values from every owner.

    ``param_groups`` / ``state_dict`` / ``load_state_dict`` are the LOCAL optimizer's (a learning-rate scheduler is attached to
    ``.local``; a checkpoint holds one optimizer file per rank and needs the same world size to resume)."""

    def __init__(self, params, lr, weight_decay, betas, eps, group=None):
        self.params = list(params)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.owner = partition(self.params, self.world)
        mine = [p for p, o in zip(self.params, self.owner) if o == self.rank]
        if not mine:                          # more ranks than tensors: an optimizer needs a parameter; this one never gets a gradient
            self._dummy = torch.nn.Parameter(torch.zeros((), device=self.params[0].device if self.params else "cpu"), requires_grad=True)
            mine = [self._dummy]
        self.local = torch.optim.AdamW(mine, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)

    @property
    def param_groups(self):
        return self.local.param_groups

    def owned_elements(self):
        return sum(p.numel() for p, o in zip(self.params, self.owner) if o == self.rank)

    @torch.no_grad()
    def step(self):
        self.local.step()
        if self.world == 1:
            return
        work = []
        for p, o in zip(self.params, self.owner):
            src = dist.get_global_rank(self.group, o) if self.group is not None else o
            work.append(dist.broadcast(p.data, src=src, group=self.group, async_op=True))
        for w in work:
            w.wait()

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        return {"local": self.local.state_dict(), "world": self.world, "rank": self.rank,
                "owned": [i for i, o in enumerate(self.owner) if o == self.rank]}

    def load_state_dict(self, sd):
        if sd.get("world") != self.world or sd.get("rank") != self.rank:
            raise ValueError(f"sharded optimizer state of rank {sd.get('rank')} / world {sd.get('world')} cannot resume rank {self.rank} / "
                             f"world {self.world}: the shards follow the world size")
        if sd.get("owned") != [i for i, o in enumerate(self.owner) if o == self.rank]:
            raise ValueError("sharded optimizer state was saved for a different parameter list")
        self.local.load_state_dict(sd["local"])
