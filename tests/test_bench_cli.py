"""bench.py's launch plumbing, executed without a GPU (VERDICT r02 "next" #1).

``python bench.py --gpus N`` with N > 1 and NO launcher environment must start its own N ranks (the round-2 file exited with rc 2
there) and rank 0 must print exactly one JSON line with ``n_gpus == N`` and ``collectives.ranks == N``.  ``--dry-cpu`` swaps the HIP
engine for a tiny Hugging Face model on the host and RCCL for gloo, so the whole control flow around the kernels -- self-launch through
``torch.distributed.run``, process group, barriers, max-over-ranks timing, the cross-rank GradCache step (chunk-wise gathers, loss on
the gathered batch, gradient averaging), the deadline guard, process-group teardown -- runs here with 2 ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "GRIT_BENCH_FORCE_DIST", "GRIT_DIST_WORLD1")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-cpu", "--layers", "1", "--steps", "2", "--warmup", "1",
                        "--pairs", "4", "--chunk", "2", *flags], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # ONE line on stdout: library chatter (gloo, RCCL) goes to stderr
    return json.loads(lines[0]), r


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    line, r = _run("--gpus", "2")
    assert "starting 2 ranks" in r.stderr
    assert line["n_gpus"] == 2 and line["collectives"]["ranks"] == 2 and line["collectives"]["backend"] == "gloo"
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert "INVALID" in line                                   # a dry run must never pass for a measurement
    c = line["contrastive"]
    assert "error" not in c, c
    assert c["global_batch"] == 8 and c["n_gpus"] == 2
    assert c["per_step_ms"]["gather_collectives"] == 2 + 4      # one gather per GradCache chunk: 4 q rows / 2 + 8 p rows / 2
    assert c["grad_norm_spread_over_ranks"] < 1e-6 * max(1.0, c["grad_norm_after_averaging"])     # replicas hold the same averaged gradient
    assert c["loss"] == c["loss"] and c["loss"] > 0


def test_single_process_dry_run_has_no_collectives():
    line, r = _run("--gpus", "1")
    assert "starting" not in r.stderr
    assert line["n_gpus"] == 1 and "collectives" not in line and "INVALID" in line
    assert "error" not in line["contrastive"]


def test_under_a_launcher_environment_bench_does_not_relaunch():
    """The driver's form: ``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2``."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-cpu", "--layers", "1", "--steps",
                        "1", "--warmup", "0", "--no-contrastive"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting 2 ranks" not in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["collectives"]["ranks"] == 2


def test_gpus_8_dry_run_is_the_eight_rank_job_the_driver_will_launch():
    """``python bench.py --gpus 8``: eight ranks, one line, ``collectives.ranks == 8``, the cross-rank GradCache step on a
    global batch of 8 x 4 pairs with one gather per chunk and identical averaged gradients on every rank -- so that the first run on an
    8-GPU node is a re-run of something that has executed."""
    line, r = _run("--gpus", "8")
    assert "starting 8 ranks" in r.stderr
    assert line["n_gpus"] == 8 and line["collectives"]["ranks"] == 8 and line["collectives"]["encode_data_path_collectives"] == 0
    assert line["config"]["parallelism"].startswith("replicas x8") and line["scaling"] == "weak" and "INVALID" in line
    c = line["contrastive"]
    assert "error" not in c, c
    assert c["global_batch"] == 32 and c["n_gpus"] == 8
    assert c["per_step_ms"]["gather_collectives"] == 2 + 4
    assert c["grad_norm_spread_over_ranks"] < 1e-6 * max(1.0, c["grad_norm_after_averaging"])


def test_summary_is_compact_and_covers_every_leg():
    """bench.compact_summary (VERDICT r05 #2b): built from a full line it stays under 1500 characters, carries both headline metrics, the
    north-star policy's rate, and the parity datum of every leg; missing legs are dropped, never a crash."""
    sys.path.insert(0, ROOT)
    import bench
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    line.pop("summary")
    s = bench.compact_summary(line)
    assert len(json.dumps(s)) <= 1500
    assert s["encode_docs_per_s"] > 0 and s["ns_policy"] in ("f16_stream", "f16_operands") and s["ns_timed_steps"] == line["steps"]
    assert s["contrastive"]["loss_abs_err"] < 1e-3 and s["contrastive"]["pass1"] == line["contrastive"]["pass1_precision"]
    assert "f16_e2e_1mcos" in s["mixtral"] and "ref_bf16_dataflow_e2e_1mcos" in s["mixtral"] and s["rag"]["decode_frac"] > 0
    assert list(json.loads(json.dumps({**line, "summary": s})))[-1] == "summary"
    bare = bench.compact_summary({"value": 1.0, "n_gpus": 8, "roofline": {"frac": 0.5}})
    assert bare["encode_docs_per_s"] == 1.0 and "mixtral" not in bare and "contrastive" not in bare
