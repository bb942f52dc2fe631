"""ZeRO-1 optimizer sharding (gritlm_amd/training/sharded_optim.py) under 2 gloo ranks: the parameters after three steps are BIT-EQUAL
to one torch.optim.AdamW over all parameters fed the same (averaged) gradients; the ownership table is balanced and identical on every
rank; through the CLI (--shard_optimizer): both ranks end with the same weights as the unsharded 2-rank run, checkpoint + resume
reproduce the uninterrupted run."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


SHAPES = [(64, 32), (32,), (16, 16), (128, 8), (8,), (40, 10)]


def _make_params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in SHAPES]


def _grads(step, world):
    """the per-rank gradients of a step (deterministic) and their average"""
    per_rank = []
    for r in range(world):
        g = torch.Generator().manual_seed(1000 + 17 * step + r)
        per_rank.append([torch.randn(s, generator=g) for s in SHAPES])
    avg = [sum(per_rank[r][i] for r in range(world)) / world for i in range(len(SHAPES))]
    return per_rank, avg


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gritlm_amd.training.sharded_optim import ShardedAdamW
        params = _make_params(3)
        opt = ShardedAdamW(params, lr=1e-2, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-8)
        sched = torch.optim.lr_scheduler.LambdaLR(opt.local, lambda s: 1.0 / (1 + s))
        for step in range(3):
            per_rank, _ = _grads(step, world)
            for p, g in zip(params, per_rank[rank]):
                p.grad = g.clone()
            for p in params:                                     # the trainer's gradient average (sync_gradients)
                dist.all_reduce(p.grad); p.grad /= world
            opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
        state_elems = sum(v.numel() for st in opt.local.state.values() for v in st.values() if torch.is_tensor(v) and v.dim() > 0)
        ret[rank] = dict(params=[p.detach().clone() for p in params], owner=list(opt.owner), owned=opt.owned_elements(), state_elems=state_elems,
                         sd_keys=sorted(opt.state_dict().keys()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_adamw_is_bit_equal_to_plain_adamw():
    world = 2
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    ref = _make_params(3)
    opt = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1 + s))
    for step in range(3):
        per_rank, _ = _grads(step, world)
        for i, p in enumerate(ref):
            p.grad = (per_rank[0][i] + per_rank[1][i]) / world          # the same summation order as the all-reduce of two ranks
        opt.step(); sched.step(); opt.zero_grad()
    total = sum(p.numel() for p in ref)
    assert ret[0]["owner"] == ret[1]["owner"] and set(ret[0]["owner"]) == {0, 1}
    assert ret[0]["owned"] + ret[1]["owned"] == total and abs(ret[0]["owned"] - ret[1]["owned"]) <= max(p.numel() for p in ref)
    for r in range(world):
        assert ret[r]["state_elems"] == 2 * ret[r]["owned"]            # exp_avg + exp_avg_sq of the owned parameters only
        for got, want in zip(ret[r]["params"], ref):
            assert torch.equal(got, want.detach())


def test_partition_is_deterministic_and_balanced():
    from gritlm_amd.training.sharded_optim import partition
    ps = [torch.empty(s) for s in SHAPES * 3]
    own = partition(ps, 4)
    assert own == partition(ps, 4)
    load = [sum(p.numel() for p, o in zip(ps, own) if o == r) for r in range(4)]
    assert max(load) - min(load) <= max(p.numel() for p in ps)
    assert partition(ps, 1) == [0] * len(ps)


def _cli_worker(rank, world, port, argv, ret, tag):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gritlm_amd.training import run
    ret[(tag, rank)] = float(run.main(argv))


def _weights(d):
    from safetensors.torch import load_file
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".safetensors"):
            out.update(load_file(os.path.join(d, f)))
    return out


def test_two_rank_cli_with_sharded_optimizer_equals_the_unsharded_run_and_resumes(tmp_path):
    import synth
    d = synth.build_mistral_dir(str(tmp_path / "m32"), "tiny", 0, "float32")
    W = synth.WORDS
    rows = [{"query": " ".join(W[i:i + 5]), "pos": [" ".join(W[i + 1:i + 9])], "neg": [" ".join(W[j:j + 7]) for j in range(i + 20, i + 24)]}
            for i in range(0, 64, 2)]
    data = str(tmp_path / "emb.jsonl")
    open(data, "w").write("\n".join(json.dumps(r) for r in rows))

    def argv(out, *extra):
        return ["--model_name_or_path", d, "--train_data", data, "--output_dir", str(tmp_path / out), "--per_device_train_batch_size", "2",
                "--train_group_size", "2", "--pooling_method", "mean", "--learning_rate", "1e-3", "--query_max_len", "16",
                "--passage_max_len", "24", "--report_to", "none", "--use_cpu", "--negatives_cross_device", "--save_safetensors", "true",
                "--save_strategy", "steps", "--save_steps", "2", "--max_steps", "4", *extra]

    mgr = mp.Manager(); ret = mgr.dict()
    for tag, extra in (("plain", ()), ("sharded", ("--shard_optimizer",))):
        mp.spawn(_cli_worker, args=(2, _free_port(), argv(tag, *extra), ret, tag), nprocs=2, join=True)
    ck = str(tmp_path / "sharded" / "checkpoint-2")
    assert {"optimizer_shard_0.pt", "optimizer_shard_1.pt", "scheduler.pt", "trainer_state.json"} <= set(os.listdir(ck))
    assert "optimizer.pt" not in os.listdir(ck) and "optimizer.pt" in os.listdir(str(tmp_path / "plain" / "checkpoint-2"))
    mp.spawn(_cli_worker, args=(2, _free_port(), argv("resumed", "--shard_optimizer", "--resume_from_checkpoint", ck), ret, "resumed"),
             nprocs=2, join=True)
    wp, ws, wr = _weights(str(tmp_path / "plain")), _weights(str(tmp_path / "sharded")), _weights(str(tmp_path / "resumed"))
    for k in wp:
        assert torch.equal(wp[k], ws[k]), k                             # element-wise AdamW: sharding the state changes no bit
        assert torch.allclose(ws[k].float(), wr[k].float(), rtol=0, atol=1e-6), k
    assert ret[("plain", 0)] == ret[("sharded", 0)]
    # a sharded checkpoint does not resume an unsharded run (and says why)
    with pytest.raises(Exception):
        mp.spawn(_cli_worker, args=(2, _free_port(), argv("bad", "--resume_from_checkpoint", ck), ret, "bad"), nprocs=2, join=True)
