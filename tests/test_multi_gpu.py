"""The N > 1 paths on REAL devices: `-m gpu` tests that need at least two GPUs in the box (skipped otherwise -- the builder's box has one;
the driver's 8-GPU node runs them), so that the first multi-GPU bench run is not also the first multi-GPU execution.

  * 2-rank RCCL contrastive loss and GradCache step through torch.distributed AND through the C ABI (GRIT_NATIVE_COMM=1, grit_comm_*:
    grouped ncclAllGather on a CU-masked side stream), against the fixtures the REFERENCE produced on a 2-rank gloo run
    (tests/golden/infonce_dist2.npz: loss + per-rank dq / dp) and on its global batch (gradcache_tiny.npz: loss + weight gradients);
  * OverlappedGradSync (bucketed all-reduce under the last chunk's backward) == a blocking all-reduce of the same gradients;
  * `--shard_optimizer` (ZeRO-1 over RCCL broadcasts) ends with the weights of the unsharded run;
  * in-process multi-GPU encode (one GritLM, one engine replica per device) across two REAL devices == one engine, bit for bit;
  * `bench.py --gpus 2` (RCCL, one rank per GPU) prints its one JSON line.

Every worker takes the device kind as an argument: the SAME code runs here on `cpu` + gloo (the tests without the gpu marker at the
bottom), so the harness itself -- slicing, fixtures, comparisons -- is exercised in the build container."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

two_gpus = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs in one box")
GRAD_NAMES = ("layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight", "norm.weight")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port, kind, native_comm=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.pop("GRIT_NATIVE_COMM", None)
    if native_comm:
        os.environ["GRIT_NATIVE_COMM"] = "1"
    if kind == "cuda":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        return torch.device("cuda", rank)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cpu")


def _train_model(model_dir, dev, cross=True):
    from gritlm_amd.training import GritLMTrainModel
    kw = dict(torch_dtype=torch.bfloat16) if dev.type == "cuda" else {}
    m = GritLMTrainModel(model_name_or_path=model_dir, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc", temperature=0.02,
                         negatives_cross_device=cross, device=str(dev), **kw)
    if dev.type == "cuda":
        m.enable_native()
    m.model.train()
    return m


def _half_batch(g, rank, world, dev):
    B, G = g["q_ids"].shape[0], int(g["group"])
    bq = B // world
    sq, sp = slice(rank * bq, (rank + 1) * bq), slice(rank * bq * G, (rank + 1) * bq * G)
    t = lambda a: torch.from_numpy(a).to(dev)
    return ({"input_ids": t(g["q_ids"][sq]), "attention_mask": t(g["q_mask"][sq])}, {"input_ids": t(g["p_ids"][sp]), "attention_mask": t(g["p_mask"][sp])})


def _backbone_grads(m):
    bb = m._backbone() if hasattr(m, "_backbone") else m.model
    sd = dict(bb.named_parameters())
    return {n: sd[n].grad.detach().float().cpu().numpy().copy() for n in GRAD_NAMES}


# ------------------------------------------------------------------------------------------------ workers
def _loss_worker(rank, world, port, kind, native_comm, ret):
    dev = _init(rank, world, port, kind, native_comm)
    try:
        from gritlm_amd.training.model import DistributedContrastiveLoss, packed_all_gather
        g = np.load(os.path.join(GOLDEN, "infonce_dist2.npz"))
        q, p, tau = g["q"], g["p"], float(g["tau"])
        bq, bp = q.shape[0] // world, p.shape[0] // world
        tq = torch.from_numpy(q[rank * bq:(rank + 1) * bq].copy()).to(dev).requires_grad_()
        tp = torch.from_numpy(p[rank * bp:(rank + 1) * bp].copy()).to(dev).requires_grad_()
        loss = DistributedContrastiveLoss(tau, True)(tq, tp)
        loss.backward()
        qa, pa = packed_all_gather(tq.detach(), tp.detach(), world)
        ret[rank] = dict(loss=float(loss.item()), dq=tq.grad.cpu().numpy(), dp=tp.grad.cpu().numpy(), q_all=qa.cpu().numpy(), p_all=pa.cpu().numpy(),
                         backend=dist.get_backend())
    finally:
        dist.destroy_process_group()


def _gradcache_worker(rank, world, port, kind, native_comm, model_dir, ret):
    dev = _init(rank, world, port, kind, native_comm)
    try:
        from gritlm_amd.training import GradCacheStep
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = _train_model(model_dir, dev)
        q, p = _half_batch(g, rank, world, dev)
        loss = GradCacheStep(m, chunk_size=2)(q, p, sync=True)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        ret[rank] = dict(loss=float(loss.item()), grads=_backbone_grads(m), backend=dist.get_backend())
    finally:
        dist.destroy_process_group()


def _overlap_worker(rank, world, port, kind, model_dir, ret):
    """the same half batch twice on fresh gradients: (a) OverlappedGradSync inside the step, (b) no sync in the step, then ONE blocking
    all-reduce(mean) per parameter"""
    dev = _init(rank, world, port, kind)
    try:
        from gritlm_amd.training import GradCacheStep
        g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
        m = _train_model(model_dir, dev, cross=False)
        q, p = _half_batch(g, rank, world, dev)
        GradCacheStep(m, chunk_size=2)(q, p, sync=True)
        a = _backbone_grads(m)
        m.zero_grad(set_to_none=True)
        GradCacheStep(m, chunk_size=2)(q, p, sync=False)
        bb = m._backbone() if hasattr(m, "_backbone") else m.model
        for prm in bb.parameters():
            if prm.grad is not None:
                buf = prm.grad.float()
                dist.all_reduce(buf)
                prm.grad.copy_((buf / world).to(prm.grad.dtype))
        ret[rank] = dict(overlapped=a, blocking=_backbone_grads(m))
    finally:
        dist.destroy_process_group()


def _cli_worker(rank, world, port, argv, ret, tag):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.set_num_threads(2)
    from gritlm_amd.training import run
    ret[(tag, rank)] = float(run.main(argv))


# ------------------------------------------------------------------------------------------------ comparisons (shared by both device kinds)
def _assert_loss_matches_reference_run(ret, world, atol):
    g = np.load(os.path.join(GOLDEN, "infonce_dist2.npz"))
    assert abs(ret[0]["loss"] - ret[1]["loss"]) < 1e-6
    for r in range(world):
        assert abs(ret[r]["loss"] - float(g[f"loss_rank{r}"])) < 10 * atol
        np.testing.assert_allclose(ret[r]["dq"], g[f"dq_rank{r}"], atol=atol)
        np.testing.assert_allclose(ret[r]["dp"], g[f"dp_rank{r}"], atol=atol)
        np.testing.assert_array_equal(ret[r]["q_all"], g["q"])              # rank order == torch.cat order (targets are arange(B) * G)
        np.testing.assert_array_equal(ret[r]["p_all"], g["p"])


def _assert_gradcache_matches_reference(ret, world, loss_rel, grad_rel):
    g = np.load(os.path.join(GOLDEN, "gradcache_tiny.npz"))
    ref_loss = float(g["loss_gradcache"])
    for r in range(world):
        assert abs(ret[r]["loss"] - ref_loss) < loss_rel * abs(ref_loss), (ret[r]["loss"], ref_loss)
        for n, got in ret[r]["grads"].items():
            ref = g["grad_gradcache/" + n]
            rel = np.linalg.norm(world * got - ref) / np.linalg.norm(ref)       # ranks hold the MEAN of the per-rank gradients
            assert rel < grad_rel, (n, rel)
    for n in ret[0]["grads"]:
        np.testing.assert_array_equal(ret[0]["grads"][n], ret[1]["grads"][n])      # replicas stay in lock-step, bit for bit


def _run_loss(kind, native_comm=False):
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_loss_worker, args=(2, _free_port(), kind, native_comm, ret), nprocs=2, join=True)
    return ret


def _run_gradcache(kind, tmp_path, native_comm=False):
    import synth
    d = synth.build_mistral_dir(str(tmp_path / ("m16" if kind == "cuda" else "m32")), "tiny", 0, "bfloat16" if kind == "cuda" else "float32")
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_gradcache_worker, args=(2, _free_port(), kind, native_comm, d, ret), nprocs=2, join=True)
    return ret


def _run_overlap(kind, tmp_path):
    import synth
    d = synth.build_mistral_dir(str(tmp_path / ("m16" if kind == "cuda" else "m32")), "tiny", 0, "bfloat16" if kind == "cuda" else "float32")
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), kind, d, ret), nprocs=2, join=True)
    return ret


def _assert_overlap_equals_blocking(ret, rel_tol):
    for r in range(2):
        for n in GRAD_NAMES:
            a, b = ret[r]["overlapped"][n], ret[r]["blocking"][n]
            rel = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-20)
            assert rel <= rel_tol, (n, rel)                    # bf16 gradients: the bucketed sum rounds once more than the fp32 blocking mean
    for n in GRAD_NAMES:
        np.testing.assert_array_equal(ret[0]["overlapped"][n], ret[1]["overlapped"][n])


def _cli_argv(tmp_path, model_dir, data, out, *extra, cpu):
    return ["--model_name_or_path", model_dir, "--train_data", data, "--output_dir", str(tmp_path / out), "--per_device_train_batch_size", "2",
            "--train_group_size", "2", "--pooling_method", "mean", "--learning_rate", "1e-3", "--query_max_len", "16", "--passage_max_len", "24",
            "--report_to", "none", "--negatives_cross_device", "--save_safetensors", "true", "--max_steps", "3",
            *(("--use_cpu",) if cpu else ("--bf16",)), *extra]


def _weights(d):
    from safetensors.torch import load_file
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".safetensors"):
            out.update(load_file(os.path.join(d, f)))
    return out


def _run_sharded_cli(tmp_path, cpu):
    import synth
    d = synth.build_mistral_dir(str(tmp_path / "m"), "tiny", 0, "float32" if cpu else "bfloat16")
    W = synth.WORDS
    rows = [{"query": " ".join(W[i:i + 5]), "pos": [" ".join(W[i + 1:i + 9])], "neg": [" ".join(W[j:j + 7]) for j in range(i + 20, i + 24)]}
            for i in range(0, 64, 2)]
    data = str(tmp_path / "emb.jsonl")
    open(data, "w").write("\n".join(json.dumps(r) for r in rows))
    mgr = mp.Manager(); ret = mgr.dict()
    for tag, extra in (("plain", ()), ("sharded", ("--shard_optimizer",))):
        mp.spawn(_cli_worker, args=(2, _free_port(), _cli_argv(tmp_path, d, data, tag, *extra, cpu=cpu), ret, tag), nprocs=2, join=True)
    wp, ws = _weights(str(tmp_path / "plain")), _weights(str(tmp_path / "sharded"))
    assert wp.keys() == ws.keys() and len(wp) > 0
    for k in wp:
        assert torch.equal(wp[k], ws[k]), k            # element-wise AdamW: sharding the optimizer state changes no bit of the weights
    assert ret[("plain", 0)] == ret[("sharded", 0)] and ret[("plain", 0)] == ret[("plain", 1)]


# ------------------------------------------------------------------------------------------------ two real GPUs (RCCL)
@pytest.mark.gpu
@two_gpus
@pytest.mark.parametrize("native_comm", [False, True], ids=["torch_distributed", "grit_comm_c_abi"])
def test_two_rank_rccl_contrastive_loss_matches_reference_gloo_run(native_comm):
    ret = _run_loss("cuda", native_comm)
    assert ret[0]["backend"] == "nccl"
    _assert_loss_matches_reference_run(ret, 2, atol=2e-5)            # fp32 reps; the exact-f32 MFMA InfoNCE kernel vs the reference's fp32 matmul


@pytest.mark.gpu
@two_gpus
@pytest.mark.parametrize("native_comm", [False, True], ids=["torch_distributed", "grit_comm_c_abi"])
def test_two_rank_rccl_gradcache_step_equals_reference_global_batch(tmp_path, native_comm):
    ret = _run_gradcache("cuda", tmp_path, native_comm)
    assert ret[0]["backend"] == "nccl"
    _assert_gradcache_matches_reference(ret, 2, loss_rel=2e-3, grad_rel=6e-2)      # bf16 training engine vs the reference's fp32 step (the bounds of rccl_world1_step)


@pytest.mark.gpu
@two_gpus
def test_overlapped_grad_sync_equals_blocking_allreduce_on_rccl(tmp_path):
    _assert_overlap_equals_blocking(_run_overlap("cuda", tmp_path), rel_tol=8e-3)


@pytest.mark.gpu
@two_gpus
def test_sharded_optimizer_on_rccl_equals_the_unsharded_run(tmp_path):
    _run_sharded_cli(tmp_path, cpu=False)


@pytest.mark.gpu
@two_gpus
def test_in_process_encode_across_two_real_devices_is_bit_identical(tmp_path):
    import synth
    from gritlm_amd import GritLM
    d16 = synth.build_mistral_dir(str(tmp_path / "m16"), "tiny", 0, "bfloat16")
    sents = synth.make_sentences(23, seed=9, min_words=2, max_words=40)
    instr = "Represent the sentence: "
    one = GritLM(d16, mode="embedding", pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, devices=["cuda:0"])
    two = GritLM(d16, mode="embedding", pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, devices=["cuda:0", "cuda:1"])
    assert one.num_gpus == 1 and two.num_gpus == 2 and [e.device.index for e in two.engines] == [0, 1]
    base = one.encode(sents, batch_size=8, max_length=64, instruction=instr)
    got = two.encode(sents, batch_size=4, max_length=64, instruction=instr)          # 4 x 2 replicas: the same batches of 8
    np.testing.assert_array_equal(base, got)
    f16 = GritLM(d16, mode="embedding", pooling_method="mean", attn="bbcc", device="cuda", torch_dtype=torch.bfloat16, devices=["cuda:0", "cuda:1"],
                 precision="f16_operands")                                            # per-device overflow flags, per-device fp16 weight copies
    g16 = f16.encode(sents, batch_size=4, max_length=64, instruction=instr)
    assert np.all(1 - np.sum(g16 * base, axis=1) < 1e-4)


@pytest.mark.gpu
@two_gpus
def test_bench_two_gpus_prints_one_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2", "--pairs", "8",
                        "--chunk", "4", "--contrastive-steps", "1", "--no-ragged"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["collectives"]["backend"] == "nccl"
    c = d["contrastive"]
    assert "error" not in c and c["n_gpus"] == 2 and np.isfinite(c["loss"]) and c["value"] > 0


# ------------------------------------------------------------------------------------------------ the same harness on cpu + gloo (runs here)
def test_harness_two_rank_loss_on_gloo():
    _assert_loss_matches_reference_run(_run_loss("cpu"), 2, atol=1e-6)


def test_harness_two_rank_gradcache_on_gloo(tmp_path):
    _assert_gradcache_matches_reference(_run_gradcache("cpu", tmp_path), 2, loss_rel=1e-4, grad_rel=3e-3)


def test_harness_overlap_equals_blocking_on_gloo(tmp_path):
    _assert_overlap_equals_blocking(_run_overlap("cpu", tmp_path), rel_tol=1e-6)


def test_harness_sharded_cli_on_gloo(tmp_path):
    _run_sharded_cli(tmp_path, cpu=True)
