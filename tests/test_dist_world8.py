"""CPU, world_size 8, gloo: the partition of the contrastive step at the world size BASELINE configs[2] names, without the hardware
(VERDICT r05 #6).  Rank r owns global query rows [r B, (r + 1) B) and passage rows [r B G, (r + 1) B G); the gathered matrices must come out
in rank-major (= ``torch.cat``) order because the targets are ``arange(W B) * G`` (gritlm/training/model.py:45-46, :57-58); every rank
computes the SAME global loss and differentiates only its own rows (:49-60); the averaged weight gradients times W are the gradients of
the global-batch loss.  Checked against oracle.distributed_infonce for EVERY rank and against a single-process run of the global batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

WORLD, B_LOC, GROUP, H, TAU = 8, 3, 2, 32, 0.02


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _reps(seed=77):
    rng = np.random.default_rng(seed)
    nrm = lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    return nrm(rng.standard_normal((WORLD * B_LOC, H))), nrm(rng.standard_normal((WORLD * B_LOC * GROUP, H)))


def _loss_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gritlm_amd.training.gradcache import ChunkGather
        from gritlm_amd.training.model import DistributedContrastiveLoss, packed_all_gather
        q, p = _reps()
        bq, bp = B_LOC, B_LOC * GROUP
        tq = torch.from_numpy(q[rank * bq:(rank + 1) * bq].copy()).requires_grad_()
        tp = torch.from_numpy(p[rank * bp:(rank + 1) * bp].copy()).requires_grad_()
        fn = DistributedContrastiveLoss(TAU, True)
        loss = fn(tq, tp)
        loss.backward()
        qa, pa = packed_all_gather(tq.detach(), tp.detach(), world)
        # the chunk-wise exchange GradCache pass 1 rides on: groups of 2 rows + a remainder group, several adds per group
        cg = ChunkGather(bp, H, torch.float32, "cpu", group_rows=4)
        for s in range(0, bp, 3):
            cg.add(tp.detach()[s:s + 3])
        p_chunked = cg.finish()
        # ... and the loss on an exchange that already happened (GradCacheStep): same value, same local-row gradients
        tq2, tp2 = tq.detach().clone().requires_grad_(), tp.detach().clone().requires_grad_()
        loss2 = fn.with_gathered(tq2, tp2, qa, p_chunked)
        loss2.backward()
        ret[rank] = dict(loss=loss.item(), dq=tq.grad.numpy(), dp=tp.grad.numpy(), q_all=qa.numpy(), p_all=pa.numpy(), p_chunked=p_chunked.numpy(),
                         gathers=cg.calls, loss2=loss2.item(), dq2=tq2.grad.numpy(), dp2=tp2.grad.numpy())
    finally:
        dist.destroy_process_group()


def test_eight_rank_contrastive_loss_partition_matches_the_oracle_on_every_rank():
    import gritlm_oracle as O
    q, p = _reps()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_loss_worker, args=(WORLD, _free_port(), ret), nprocs=WORLD, join=True)
    q_sh = [q[r * B_LOC:(r + 1) * B_LOC] for r in range(WORLD)]
    p_sh = [p[r * B_LOC * GROUP:(r + 1) * B_LOC * GROUP] for r in range(WORLD)]
    losses = [ret[r]["loss"] for r in range(WORLD)]
    assert max(losses) - min(losses) < 1e-6                                   # ONE global loss, on every rank
    for r in range(WORLD):
        l_ref, dq_ref, dp_ref = O.distributed_infonce(q_sh, p_sh, TAU, r)
        assert abs(ret[r]["loss"] - l_ref) < 2e-5 and abs(ret[r]["loss2"] - l_ref) < 2e-5
        np.testing.assert_allclose(ret[r]["dq"], dq_ref, atol=2e-6)           # the rank's OWN rows only (the other shards are constants)
        np.testing.assert_allclose(ret[r]["dp"], dp_ref, atol=2e-6)
        np.testing.assert_allclose(ret[r]["dq2"], dq_ref, atol=2e-6)
        np.testing.assert_allclose(ret[r]["dp2"], dp_ref, atol=2e-6)
        np.testing.assert_array_equal(ret[r]["q_all"], q)                     # rank-major order == torch.cat order: target = arange(W B) * G
        np.testing.assert_array_equal(ret[r]["p_all"], p)
        np.testing.assert_array_equal(ret[r]["p_chunked"], p)                 # the chunk-wise gathers stitch back into the same order
        assert ret[r]["gathers"] == 2                                         # 6 local rows in groups of 4: one full group + the remainder
    # the target of global query i is passage i * G: the oracle's loss on a rank-PERMUTED gather differs (the order is not a formality)
    perm = np.concatenate([p_sh[r] for r in (1, 0, 2, 3, 4, 5, 6, 7)])
    assert abs(O.infonce(q, perm, TAU)[0] - losses[0]) > 1e-3


def _gc_worker(rank, world, port, model_dir, ids, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=TAU, negatives_cross_device=True, device="cpu")
        m.model.train()
        qi, qm, pi, pm = ids
        sq, sp = slice(rank * 2, (rank + 1) * 2), slice(rank * 2 * GROUP, (rank + 1) * 2 * GROUP)
        q = {"input_ids": torch.from_numpy(qi[sq]), "attention_mask": torch.from_numpy(qm[sq])}
        p = {"input_ids": torch.from_numpy(pi[sp]), "attention_mask": torch.from_numpy(pm[sp])}
        loss = GradCacheStep(m, chunk_size=2)(q, p)
        sd = dict(m.model.named_parameters())
        ret[rank] = dict(loss=loss.item(), grads={n: sd[n].grad.numpy().copy() for n in
                                                  ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight", "norm.weight")})
    finally:
        dist.destroy_process_group()


def test_eight_rank_gradcache_step_equals_the_global_batch(tmp_path):
    """8 ranks x (2 queries + 4 passages) through GradCacheStep (chunk-wise gathers under pass 1, loss on the gathered batch, local-shard
    backward, gradient averaging) == ONE process on the global batch of 16 queries + 32 passages: same loss (oracle InfoNCE on the global
    reps), W x averaged gradients == the global-batch gradients, replicas in lock-step."""
    import gritlm_oracle as O
    import synth
    from gritlm_amd.training import GritLMTrainModel
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    cfg = synth.CONFIGS["tiny"]
    qi, qm = synth.make_batch(cfg, 2 * WORLD, 16, seed=31, min_len=5)
    pi, pm = synth.make_batch(cfg, 2 * WORLD * GROUP, 24, seed=32, min_len=7)
    # single process, global batch, direct forward + backward (no GradCache, no collectives)
    torch.set_num_threads(4)
    m = GritLMTrainModel(model_name_or_path=d, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc", temperature=TAU,
                         negatives_cross_device=False, device="cpu")
    m.model.train()
    out = m(query={"input_ids": torch.from_numpy(qi), "attention_mask": torch.from_numpy(qm)},
            passage={"input_ids": torch.from_numpy(pi), "attention_mask": torch.from_numpy(pm)})
    out.loss.backward()
    l_oracle = O.infonce(out.q_reps.detach().numpy(), out.p_reps.detach().numpy(), TAU)[0]
    assert abs(float(out.loss.detach()) - l_oracle) < 2e-5
    ref_g = {n: t.grad.numpy().copy() for n, t in m.model.named_parameters()}
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_gc_worker, args=(WORLD, _free_port(), d, (qi, qm, pi, pm), ret), nprocs=WORLD, join=True)
    for r in range(WORLD):
        assert abs(ret[r]["loss"] - l_oracle) < 2e-4
        for n, got in ret[r]["grads"].items():
            np.testing.assert_allclose(WORLD * got, ref_g[n], atol=3e-3 * np.abs(ref_g[n]).max())
            np.testing.assert_array_equal(got, ret[0]["grads"][n])            # replicas stay in lock-step
