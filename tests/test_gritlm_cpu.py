"""CPU plumbing parity: gritlm_amd.GritLM reproduces the REFERENCE GritLM.encode() outputs (tests/golden/gritlm_encode.npz)
on the host path -- BASELINE.json configs[0] (GPT-Neo / SGPT shape, weightedmean, 32 docs @ seq128) and a tiny Mistral."""
import os

import numpy as np
import pytest
import torch

import synth
from gritlm_amd import GritLM


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "gritlm_encode.npz"))


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    td = tmp_path_factory.mktemp("models")
    return dict(neo=synth.build_gptneo_dir(str(td / "neo"), 0), m32=synth.build_mistral_dir(str(td / "m32"), "tiny", 0, "float32"))


def test_config1_sgpt_weightedmean_plumbing(gold, dirs):
    sents = [str(s) for s in gold["sentences"]]
    m = GritLM(dirs["neo"], pooling_method="weightedmean", attn=None, device="cpu")
    assert m.engine is None and m.embedding_attr == "transformer"
    e = m.encode(sents, batch_size=8, max_length=128)
    assert e.dtype == np.float32 and e.shape == (32, 64)
    np.testing.assert_allclose(e, gold["neo_weightedmean"], atol=2e-6)
    m.pooling_method = "lasttoken"
    np.testing.assert_allclose(m.encode(sents[:8], batch_size=8, max_length=128), gold["neo_lasttoken"], atol=2e-6)


def test_mistral_cpu_matches_reference_including_instruction_masking(gold, dirs):
    sents = [str(s) for s in gold["sentences"]]
    instr = str(gold["instruction"]) + " "
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")
    np.testing.assert_allclose(m.encode(sents[:12], batch_size=5, max_length=64, instruction=instr), gold["mistral_fp32_mean_instr"], atol=5e-6)
    np.testing.assert_allclose(m.encode(sents[:12], batch_size=5, max_length=64), gold["mistral_fp32_mean"], atol=5e-6)
    np.testing.assert_allclose(m.encode(sents[:4], batch_size=5, max_length=64, instruction=instr, embed_instruction=True),
                               gold["mistral_fp32_mean_embed_instr"], atol=5e-6)
    one = m.encode(sents[0], max_length=64)
    assert one.shape == (256,)                                   # str in -> 1-D out (gritlm.py:169-170)
    t = m.encode(sents[:3], max_length=64, convert_to_tensor=True)
    assert isinstance(t, torch.Tensor) and t.dtype == torch.float32
    tb = m.encode(sents[:3], max_length=64, convert_to_tensor=True, recast=True)
    assert tb.dtype == m.model.dtype
    c = GritLM(dirs["m32"], pooling_method="weightedmean", attn="cccc", device="cpu")
    np.testing.assert_allclose(c.encode(sents[:6], batch_size=6, max_length=64), gold["mistral_fp32_wmean_causal"], atol=5e-6)
    corp = m.encode_corpus([{"title": "w1", "text": "w2 w3"}, {"text": "w4"}], max_length=16)
    np.testing.assert_allclose(corp, m.encode(["w1 w2 w3", "w4"], max_length=16), atol=1e-7)


def test_no_sentences_raises_like_the_reference(dirs):
    """gritlm.py:162-164: np.concatenate([]) -> ValueError; torch.cat([]) raises too (ValueError or RuntimeError by torch version)."""
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")
    with pytest.raises(ValueError, match="at least one array"):
        m.encode([], max_length=8)
    with pytest.raises((ValueError, RuntimeError)):
        m.encode([], max_length=8, convert_to_tensor=True)


def test_constructor_contract(dirs):
    with pytest.raises(ValueError, match="Mixed attention"):
        GritLM(dirs["m32"], attn="bbcb", device="cpu")
    m = GritLM(dirs["m32"], pooling_method="weighted_mean", device="cpu")     # README.md:38 typo -> NotImplementedError at pool
    with pytest.raises(NotImplementedError):
        m.encode(["w1 w2"], max_length=8)
    with pytest.raises(RuntimeError, match="native=True"):
        GritLM(dirs["m32"], device="cpu", native=True)
    for a in ("model", "tokenizer", "device", "generate", "projection", "pooling_method", "normalized", "attn", "embed_eos", "num_gpus"):
        assert hasattr(m, a), a


def test_pooling_mutates_mask_like_the_reference(golden_dir, dirs):
    g = np.load(os.path.join(golden_dir, "pooling.npz"))
    m = GritLM(dirs["m32"], device="cpu")
    for method in ("mean", "weightedmean", "cls", "lasttoken"):
        m.pooling_method = method
        mask = torch.from_numpy(g["mask"].copy())
        out = m.pooling(torch.from_numpy(g["hidden"]), mask)
        np.testing.assert_allclose(out.numpy(), g[f"pool_{method}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(mask.numpy(), g[f"mask_after_{method}"])


# ---------------------------------------------------------------------------------------------- in-process multi-GPU encode (host logic)
def test_row_chunks_are_dataparallel_scatter_chunks():
    """gritlm/gritlm.py:69-75: the reference scatters a batch with nn.DataParallel = torch.chunk along dim 0."""
    for n_rows in (0, 1, 2, 5, 7, 8, 9, 16, 255, 256, 2048):
        for parts in (1, 2, 3, 4, 8):
            want = [int(c.numel()) for c in torch.arange(n_rows).chunk(parts)] if n_rows else []
            got = GritLM._row_chunks(n_rows, parts)
            assert [b - a for a, b in got] == want, (n_rows, parts, got, want)
            assert all(got[i][1] == got[i + 1][0] for i in range(len(got) - 1)) and (not got or (got[0][0] == 0 and got[-1][1] == n_rows))


class _FakeEngine:
    """Stands in for one engine replica: 'embeds' a row as (sum of its real token ids, number of real tokens, rows in the call, replica id);
    records what it was handed."""

    def __init__(self, tag):
        self.device, self.tag, self.calls = torch.device("cpu"), tag, []

    def encode_pooled(self, ids, mask, method, normalize, instr_len=None):
        assert ids.device.type == "cpu" and mask.device.type == "cpu", "the tokenizer's HOST tensors go to the replicas"
        self.calls.append((tuple(ids.shape), None if instr_len is None else int(instr_len[0])))
        m = mask.to(torch.float32)
        return torch.stack([(ids * mask).sum(1).float(), m.sum(1), torch.full((ids.shape[0],), float(ids.shape[0])),
                            torch.full((ids.shape[0],), float(self.tag))], dim=1)


def test_two_replica_split_keeps_row_order_and_scales_the_batch(dirs):
    """One process, one GritLM, two engine replicas (the reference: nn.DataParallel over 2 GPUs, batch_size x 2): every
    (batch_size x 2) batch is tokenised ONCE (one padding length), its rows are dealt in contiguous halves, results come back in
    sentence order; the instruction length reaches every replica."""
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")
    sents = synth.make_sentences(11, seed=5, min_words=3, max_words=30)
    m.engine, m.engines, m.num_gpus = _FakeEngine(0), [], 1
    one = m.encode(sents, batch_size=4, max_length=64)
    e0, e1 = _FakeEngine(0), _FakeEngine(1)
    m.engine, m.engines, m.num_gpus = e0, [e0, e1], 2
    two = m.encode(sents, batch_size=2, max_length=64, instruction="Represent: ")          # 2 x 2 GPUs = batches of 4 sentences
    assert [c[0][0] for c in e0.calls] == [2, 2, 2] and [c[0][0] for c in e1.calls] == [2, 2, 1]       # 4 + 4 + 3 rows -> (2,2) (2,2) (2,1)
    assert all(c[1] == e0.calls[0][1] and c[1] > 0 for c in e0.calls + e1.calls)
    assert e0.calls[0][0][1] == e1.calls[0][0][1], "both halves of a batch share the batch's padding length"
    np.testing.assert_array_equal(two[:, 3], np.array([0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1], dtype=np.float32))
    m.engine, m.engines, m.num_gpus = e0, [e0, e1], 2
    plain = m.encode(sents, batch_size=2, max_length=64)
    np.testing.assert_array_equal(plain[:, :2], one[:, :2])                    # same rows, same order, whichever replica computed them


class _ReplicaEngine:
    """engine stand-in for `_parallelize`: records which devices it was replicated onto"""

    def __init__(self, device):
        self.device = torch.device(device)
        self.made = []

    def replica(self, d):
        self.made.append(torch.device(d))
        return _ReplicaEngine(d)


def test_parallelize_stays_on_one_device_inside_a_distributed_launch(dirs, monkeypatch):
    """in a one-process-per-GPU launch every rank sees every GPU; auto-replication there would put world_size copies
    of the model on each GPU.  `_parallelize` must do nothing when LOCAL_RANK / RANK is set (or torch.distributed is initialised) unless
    the caller names `devices=` explicitly; device lists are normalised ('cuda' == the current device, duplicates dropped)."""
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    m.engine = _ReplicaEngine("cuda:0")
    monkeypatch.setenv("LOCAL_RANK", "1")
    m.engines, m.num_gpus = [], 1
    m._parallelize(None)
    assert m.engines == [] and m.num_gpus == 1 and m.engine.made == []
    m._parallelize(["cuda:0", "cuda:2"])                       # explicit request: honoured even under a launcher
    assert m.num_gpus == 2 and m.engine.made == [torch.device("cuda:2")] and m.engines[0] is m.engine
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.delenv("RANK", raising=False)
    m.engine = _ReplicaEngine("cuda:0")
    m.engines, m.num_gpus = [], 1
    m._parallelize(None)                                       # plain single-process use: one replica per visible GPU, as the reference
    assert m.num_gpus == 4 and m.engine.made == [torch.device("cuda", i) for i in (1, 2, 3)]
    m.engine = _ReplicaEngine("cuda")                          # index None == current device: no second copy on the same GPU
    m.engines, m.num_gpus = [], 1
    m._parallelize(["cuda", "cuda:0", "cuda:1", "cuda:1"])
    assert m.num_gpus == 2 and m.engine.made == [torch.device("cuda:1")]


def test_parallelize_replicates_in_a_single_task_slurm_job(dirs, monkeypatch):
    """SLURM sets SLURM_LOCALID=0 in EVERY step, including the reference's own single-task 8-GPU evaluation job
    (scripts/eval_mteb.sh: ntasks-per-node=1, gres=gpu:8, plain `python`), where the reference wraps the model in nn.DataParallel over all
    GPUs (gritlm/gritlm.py:72-75).  SLURM_LOCALID / RANK alone must therefore NOT switch replication off; more than one task does."""
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    for k in ("LOCAL_RANK", "RANK", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SLURM_LOCALID", "0"); monkeypatch.setenv("SLURM_NTASKS", "1"); monkeypatch.setenv("SLURM_PROCID", "0")
    m.engine = _ReplicaEngine("cuda:0")
    m.engines, m.num_gpus = [], 1
    m._parallelize(None)
    assert m.num_gpus == 8 and m.engine.made == [torch.device("cuda", i) for i in range(1, 8)]
    monkeypatch.setenv("SLURM_NTASKS", "8")                     # srun with one task per GPU: every task keeps to its own device
    m.engine = _ReplicaEngine("cuda:0")
    m.engines, m.num_gpus = [], 1
    m._parallelize(None)
    assert m.num_gpus == 1 and m.engine.made == []
    monkeypatch.setenv("SLURM_NTASKS", "1"); monkeypatch.setenv("WORLD_SIZE", "2")      # a launcher that exports WORLD_SIZE only
    m._parallelize(None)
    assert m.num_gpus == 1 and m.engine.made == []


def test_precision_auto_ladder_is_host_logic(dirs):
    """precision='auto' (round 6): validated, and GritLM.set_precision walks gritlm_amd.encoder.AUTO_LADDER restricted to what the engine's
    model kind supports (dense, bidirectional or causal: all four rungs; sparse-MoE: f16_operands then bf16)."""
    from gritlm_amd.encoder import AUTO_LADDER
    assert AUTO_LADDER == ("f16_stream", "f16_operands", "fp32_residual", "bf16")
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu", precision="auto")
    assert m._precision == "auto" and m.engine is None and m.precision == "auto"

    class _Eng:
        def __init__(self, sup):
            self.sup, self.precision, self.device = sup, "bf16", "cuda:0"

        def supported_precisions(self):
            return self.sup

        def set_precision(self, p):
            assert p in self.sup
            self.precision = p
    for sup, first in ((AUTO_LADDER, "f16_stream"), (("f16_operands", "bf16"), "f16_operands"), (("fp32_residual", "bf16"), "fp32_residual")):
        m.engine, m.engines = _Eng(sup), []
        m.set_precision("auto")
        assert m._auto and m._ladder == [r for r in AUTO_LADDER if r in sup] and m.precision == first
    m.set_precision("bf16")
    assert not m._auto and m.precision == "bf16"
    with pytest.raises(ValueError, match="precision"):
        m.set_precision("fp8")


def test_precision_keyword_is_validated(dirs):
    """`precision=` (extension): one of gritlm_amd.encoder.PRECISIONS; `residual_fp32=True` stays an alias of 'fp32_residual'."""
    from gritlm_amd.encoder import PRECISIONS
    assert PRECISIONS == ("bf16", "fp32_residual", "f16_operands", "f16_stream")
    with pytest.raises(ValueError, match="precision"):
        GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu", precision="fp8")
    assert GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu", residual_fp32=True)._precision == "fp32_residual"
    assert GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu", precision="f16_operands")._precision == "f16_operands"
    assert GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu")._precision == "bf16"


class _LadderEngine(_FakeEngine):
    """_FakeEngine with a precision policy and a sticky overflow flag: the flag is raised by every encode under a policy in `overflows_under`"""

    def __init__(self, tag, overflows_under=()):
        super().__init__(tag)
        self.precision, self.flag, self.overflows_under, self.cleared = "bf16", False, tuple(overflows_under), 0
        self.policies_run = []

    def supported_precisions(self):
        return ("f16_stream", "f16_operands", "fp32_residual", "bf16")

    def set_precision(self, p):
        self.precision = p

    def encode_pooled(self, ids, mask, method, normalize, instr_len=None):
        self.policies_run.append(self.precision)
        if self.precision in self.overflows_under:
            self.flag = True
        return super().encode_pooled(ids, mask, method, normalize, instr_len)

    def f16_overflowed(self, clear=True):
        f = self.flag
        if clear:
            self.flag = False
            self.cleared += 1
        return f


def test_auto_ladder_reruns_the_call_one_rung_down_and_never_raises(dirs):
    """precision='auto' (host logic; the kernels' side is tests/gpu_checks.py::check_gritlm_f16_auto_ladder): a flag set during the call
    re-runs the WHOLE call one rung down -- twice if the next rung overflows too --, the rung is sticky, every replica's flag is read and
    cleared (also a stale one at the start of a call), and an explicit fp16 policy raises naming the flagged devices."""
    from gritlm_amd._lib import GritHipError
    m = GritLM(dirs["m32"], pooling_method="mean", attn="bbcc", device="cpu", precision="auto")
    sents = synth.make_sentences(7, seed=6, min_words=3, max_words=20)
    e0, e1 = _LadderEngine(0, overflows_under=("f16_stream", "f16_operands")), _LadderEngine(1)
    m.engine, m.engines, m.num_gpus = e0, [e0, e1], 2
    m.set_precision("auto")
    assert m.precision == "f16_stream" and e1.precision == "f16_stream"
    e1.flag = True                                           # a stale flag from an earlier, unchecked forward on the second replica
    out = m.encode(sents, batch_size=2, max_length=32)
    assert out.shape[0] == 7 and m.precision == "fp32_residual" and e1.precision == "fp32_residual"      # two rungs down, all replicas together
    assert e0.policies_run[:2] == ["f16_stream", "f16_stream"] and "f16_operands" in e0.policies_run and e0.policies_run[-1] == "fp32_residual"
    n_calls = len(e0.policies_run)
    out2 = m.encode(sents, batch_size=2, max_length=32)      # sticky: straight on the rung, one pass
    assert len(e0.policies_run) == n_calls + (n_calls // 3) and set(e0.policies_run[n_calls:]) == {"fp32_residual"}
    np.testing.assert_array_equal(out[:, :2], out2[:, :2])
    # explicit policy: raises, names the device, leaves no flag behind
    m.set_precision("f16_stream")
    assert not m._auto and e0.precision == e1.precision == "f16_stream"
    with pytest.raises(GritHipError, match="fp16 range"):
        m.encode(sents, batch_size=2, max_length=32)
    assert not e0.flag and not e1.flag
    m.set_precision("bf16")
    assert m.encode(sents, batch_size=2, max_length=32).shape[0] == 7


def test_native_decoder_follows_the_engine_policy_host_logic():
    """MistralDecoder (round 6): the decode arithmetic follows the engine's precision policy -- fp16 operands under the fp16 policies --,
    `decoder.precision` pins it, invalid pins / on_overflow values raise before any kernel is launched, and GritLM(precision="auto") makes
    the bf16 repeat the decoder's overflow behaviour (the ladder's last rung)."""
    from types import SimpleNamespace
    from gritlm_amd.decoder import MistralDecoder
    eng = SimpleNamespace(cfg=SimpleNamespace(num_local_experts=0, num_attention_heads=2, num_key_value_heads=1, head_dim=128, hidden_size=256,
                                              intermediate_size=512, num_hidden_layers=1, rms_norm_eps=1e-5, rope_theta=1e4),
                          device=torch.device("cpu"), precision="bf16", window_keys=0)
    dec = MistralDecoder(eng, torch.zeros((8, 256)))
    assert dec.on_overflow == "raise" and dec.last_precision is None
    for pol, want in (("bf16", False), ("fp32_residual", False), ("f16_operands", True), ("f16_stream", True)):
        eng.precision = pol
        assert dec._f16() is want
    dec.precision = "bf16"
    assert dec._f16() is False
    eng.precision, dec.precision = "bf16", "f16"
    assert dec._f16() is True
    dec.precision = "fp8"
    with pytest.raises(ValueError):
        dec._f16()
    dec.precision = "bf16"
    with pytest.raises(ValueError):
        dec.generate(torch.zeros((1, 2), dtype=torch.long), 2, on_overflow="ignore")
    moe = MistralDecoder(SimpleNamespace(cfg=SimpleNamespace(num_local_experts=8), device=torch.device("cpu"), precision="bf16"), torch.zeros((8, 256)))
    assert moe.moe and not moe.prompt_chunk and not dec.moe          # sparse-MoE engines decode too (round 6); their prompt rides token by token
    # GritLM.native_decoder(): "auto" -> on_overflow "bf16"
    m = GritLM.__new__(GritLM)
    torch.nn.Module.__init__(m)
    m.engine, m.model, m._precision, m._decoder = eng, SimpleNamespace(lm_head=SimpleNamespace(weight=torch.zeros((8, 256)))), "auto", None
    assert m.native_decoder().on_overflow == "bf16"
    m._precision = "f16_stream"
    assert m.native_decoder().on_overflow == "raise"


def test_env_pass1_precision_lands_on_the_policy_the_model_kind_takes(monkeypatch):
    """GRIT_PASS1_PRECISION is a process-wide default: 'f16_stream' on a sparse-MoE engine (which routes on the fp32 stream and refuses it)
    lands on 'f16_operands' with a line on stdout; the same value as an explicit argument is refused as it stands."""
    from types import SimpleNamespace
    from gritlm_amd._lib import GritHipError
    from gritlm_amd.training.gradcache import GradCacheStep

    class _MoeEngine:
        nograd_precision = "bf16"

        def set_nograd_precision(self, p):
            if p == "f16_stream":
                raise GritHipError("the sparse-MoE block routes on the fp32 residual stream")
            self.nograd_precision = p
    m = SimpleNamespace(train_engine=_MoeEngine())
    monkeypatch.setenv("GRIT_PASS1_PRECISION", "f16_stream")
    assert GradCacheStep(m, 2, pass1_chunk_size=2).precision == "f16_operands"
    with pytest.raises(GritHipError):
        GradCacheStep(m, 2, pass1_chunk_size=2, precision="f16_stream")
    monkeypatch.setenv("GRIT_PASS1_PRECISION", "bf16")
    assert GradCacheStep(m, 2, pass1_chunk_size=2).precision == "bf16"
