"""CPU, world_size 2, gloo: the N>1 path of the contrastive step -- packed all-gather ordering, local-shard-only gradients,
identical global loss on every rank (reference fixture from a real 2-rank gloo run of the reference loss), gradient averaging."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gritlm_amd.training.model import DistributedContrastiveLoss, packed_all_gather
        from gritlm_amd.training.gradcache import sync_gradients
        g = np.load(os.path.join(GOLDEN, "infonce_dist2.npz"))
        q, p, tau = g["q"], g["p"], float(g["tau"])
        bq, bp = q.shape[0] // world, p.shape[0] // world
        tq = torch.from_numpy(q[rank * bq:(rank + 1) * bq].copy()).requires_grad_()
        tp = torch.from_numpy(p[rank * bp:(rank + 1) * bp].copy()).requires_grad_()
        loss = DistributedContrastiveLoss(tau, True)(tq, tp)
        loss.backward()
        qa, pa = packed_all_gather(tq.detach(), tp.detach(), world)
        # gradient averaging over ranks
        lin = torch.nn.Linear(3, 2, bias=False)
        lin.weight.grad = torch.full((2, 3), float(rank + 1))
        sync_gradients(lin)
        ret[rank] = dict(loss=loss.item(), dq=tq.grad.numpy(), dp=tp.grad.numpy(), q_all=qa.numpy(), p_all=pa.numpy(),
                         avg=lin.weight.grad.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_contrastive_loss_matches_reference_gloo_run():
    g = np.load(os.path.join(GOLDEN, "infonce_dist2.npz"))
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert abs(ret[0]["loss"] - ret[1]["loss"]) < 1e-6                      # every rank computes the same global loss
    for r in range(world):
        assert abs(ret[r]["loss"] - float(g[f"loss_rank{r}"])) < 1e-5
        np.testing.assert_allclose(ret[r]["dq"], g[f"dq_rank{r}"], atol=1e-6)
        np.testing.assert_allclose(ret[r]["dp"], g[f"dp_rank{r}"], atol=1e-6)
        np.testing.assert_array_equal(ret[r]["q_all"], g["q"])              # rank order == torch.cat order
        np.testing.assert_array_equal(ret[r]["p_all"], g["p"])
        np.testing.assert_allclose(ret[r]["avg"], 1.5)
