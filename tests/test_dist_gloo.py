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


def _gc_worker(rank, world, port, model_dir, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=True, device="cpu")
        m.model.train()
        B, G = g["q_ids"].shape[0], int(g["group"])
        bq = B // world
        sl_q, sl_p = slice(rank * bq, (rank + 1) * bq), slice(rank * bq * G, (rank + 1) * bq * G)
        q = {"input_ids": torch.from_numpy(g["q_ids"][sl_q]), "attention_mask": torch.from_numpy(g["q_mask"][sl_q])}
        p = {"input_ids": torch.from_numpy(g["p_ids"][sl_p]), "attention_mask": torch.from_numpy(g["p_mask"][sl_p])}
        loss = GradCacheStep(m, chunk_size=2)(q, p)
        sd = dict(m.model.named_parameters())
        ret[rank] = dict(loss=loss.item(), grads={n: sd[n].grad.numpy().copy() for n in
                                                  ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight", "norm.weight")})
    finally:
        dist.destroy_process_group()


def test_two_rank_gradcache_step_equals_reference_global_batch(tmp_path):
    """2 ranks x half of the reference batch, chunked gather overlapped with pass 1, gradient averaging:
    loss == the reference's global-batch loss, W * averaged grads == the reference's global-batch grads."""
    import synth
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_gc_worker, args=(world, _free_port(), d, ret), nprocs=world, join=True)
    for r in range(world):
        assert abs(ret[r]["loss"] - float(g["loss_gradcache"])) < 2e-4
        for n, got in ret[r]["grads"].items():
            ref = g["grad_gradcache/" + n]
            np.testing.assert_allclose(world * got, ref, atol=3e-3 * np.abs(ref).max())
    for n in ret[0]["grads"]:
        np.testing.assert_array_equal(ret[0]["grads"][n], ret[1]["grads"][n])      # replicas stay in lock-step


def _gc_uneven_worker(rank, world, port, model_dir, ret):
    """Rank 1's passages are padded 8 columns further than rank 0's (a collator pads every rank's batch to ITS longest text), and the
    pass-1 token cap sits between the two lengths: rank 0 runs pass 1 in calls of 4 rows, rank 1 in calls of 2."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from gritlm_amd.training import GradCacheStep, GritLMTrainModel
        import gritlm_amd.training.gradcache as gcmod
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=0.02, negatives_cross_device=True, device="cpu")
        m.model.train()
        B, G = g["q_ids"].shape[0], int(g["group"])
        bq = B // world
        sl_q, sl_p = slice(rank * bq, (rank + 1) * bq), slice(rank * bq * G, (rank + 1) * bq * G)
        q = {"input_ids": torch.from_numpy(g["q_ids"][sl_q]), "attention_mask": torch.from_numpy(g["q_mask"][sl_q])}
        p_ids, p_mask = torch.from_numpy(g["p_ids"][sl_p]), torch.from_numpy(g["p_mask"][sl_p])
        seq = p_ids.shape[1]
        os.environ["GRIT_GRADCACHE_PASS1_TOKENS"] = str(4 * seq)
        if rank == 1:
            p_ids = torch.cat([p_ids, torch.zeros((p_ids.shape[0], 8), dtype=p_ids.dtype)], dim=1)
            p_mask = torch.cat([p_mask, torch.zeros((p_mask.shape[0], 8), dtype=p_mask.dtype)], dim=1)
        p = {"input_ids": p_ids, "attention_mask": p_mask}
        step = GradCacheStep(m, chunk_size=2, pass1_chunk_size=4)
        rows = step._pass1_rows(p)
        sizes = []
        orig = gcmod.ChunkGather._gather
        def spy(self, reps):
            sizes.append(int(reps.shape[0]))
            return orig(self, reps)
        gcmod.ChunkGather._gather = spy
        loss = step(q, p)
        ret[rank] = dict(loss=loss.item(), pass1_rows=rows, gather_sizes=sizes)
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_schedule_does_not_depend_on_a_ranks_padding(tmp_path):
    """The cross-rank gathers are a function of the configuration and the per-rank batch size only: ranks whose pass-1 calls differ in
    size (different padded lengths under the token cap) still issue the same collectives, and the loss is the reference's."""
    import synth
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_gc_uneven_worker, args=(world, _free_port(), d, ret), nprocs=world, join=True)
    assert ret[0]["pass1_rows"] == 4 and ret[1]["pass1_rows"] == 2          # the premise: the ranks' model calls differ
    assert ret[0]["gather_sizes"] == ret[1]["gather_sizes"], (ret[0]["gather_sizes"], ret[1]["gather_sizes"])
    for r in range(world):
        assert abs(ret[r]["loss"] - float(g["loss_gradcache"])) < 2e-4


def _enc_worker(rank, world, port, model_dir, sents, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from gritlm_amd import GritLM
        from gritlm_amd.distributed import encode_sharded
        m = GritLM(model_dir, pooling_method="mean", attn="bbcc", device="cpu")
        ret[rank] = encode_sharded(m, sents, batch_size=4, max_length=32)
    finally:
        dist.destroy_process_group()


def test_sharded_encode_equals_single_process(tmp_path):
    """Encode shards by contiguous document slices (replicas, no data-path collective); the gathered result is the single-process result."""
    import synth
    from gritlm_amd import GritLM
    from gritlm_amd.distributed import shard_bounds
    assert [shard_bounds(7, 3, r) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    sents = synth.make_sentences(7, seed=9, max_words=25)
    ref = GritLM(d, pooling_method="mean", attn="bbcc", device="cpu").encode(sents, batch_size=4, max_length=32)
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_enc_worker, args=(2, _free_port(), d, sents, ret), nprocs=2, join=True)
    for r in range(2):
        np.testing.assert_allclose(ret[r], ref, atol=2e-6)


def _unified_worker(rank, world, port, model_dir, data_dir, out_dir, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gritlm_amd.training import run
    loss = run.main(["--model_name_or_path", model_dir, "--train_data", data_dir, "--output_dir", out_dir, "--mode", "unified",
                     "--per_device_train_batch_size", "2", "--gradient_accumulation_steps", "2", "--negatives_cross_device",
                     "--train_group_size", "4", "--pooling_method", "mean", "--max_steps", "2", "--learning_rate", "1e-3",
                     "--query_max_len", "16", "--passage_max_len", "24", "--generative_max_len", "32", "--report_to", "none", "--use_cpu"])
    ret[rank] = dict(loss=float(loss), loss_gen=float(run.main.last_loss_gen))
    if rank == 0:
        ret["saved"] = sorted(f for f in os.listdir(out_dir) if f.endswith(".safetensors"))


def test_two_rank_unified_cli_keeps_replicas_in_lockstep(tmp_path):
    """--mode unified under 2 gloo ranks: generative step (its gradients, lm_head included, stay local until the embedding step's
    averaging), cross-device negatives in the GradCache embedding step; both ranks finish with finite losses and the same embedding loss
    (it is a function of the gathered global batch)."""
    import json
    import synth
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    os.makedirs(tmp_path / "data")
    W = synth.WORDS
    rows = [{"query": " ".join(W[i:i + 5]), "pos": [" ".join(W[i + 1:i + 9])], "neg": [" ".join(W[j:j + 7]) for j in range(i + 20, i + 24)]}
            for i in range(0, 48, 2)]
    open(tmp_path / "data" / "emb.jsonl", "w").write("\n".join(json.dumps(r) for r in rows))
    open(tmp_path / "data" / "gen.jsonl", "w").write("\n".join(json.dumps({"text": [" ".join(W[i:i + 4]), " ".join(W[i + 30:i + 40])]}) for i in range(24)))
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_unified_worker, args=(2, _free_port(), d, str(tmp_path / "data"), str(tmp_path / "out"), ret), nprocs=2, join=True)
    assert all(np.isfinite(ret[r]["loss"]) and np.isfinite(ret[r]["loss_gen"]) for r in range(2))
    assert abs(ret[0]["loss"] - ret[1]["loss"]) < 1e-5
    assert ret["saved"]
