"""Compile-time guard (no GPU): hipcc's resource-usage remarks for the two MFMA kernels.

A register spill inside these kernels is not a small cost: the reload is a scratch load, scratch loads share the vmcnt counter with the
hand-placed LDS-DMA stream, and hipcc waits for them with ``s_waitcnt vmcnt(0)`` -- one spilled accumulator fragment in the persistent
GEMM's first K-tile drained the DMA queue once per tile and cost the K = 14336 launches the whole gain of the persistent form
(NOTEBOOK.md §4, "what the compiler left in the persistent loop").  So: zero VGPR spills and zero scratch in every instantiation, and the
occupancy the launch geometry counts on (two waves per SIMD)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _resources(src):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(ROOT, "gritlm_amd", "csrc", src),
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)(?: \[-Rpass|$)", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            out.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,needle,count", [("gemm_bf16.hip", "gemm_bf16_nt_k", 28), ("attention.hip", "attn_bidir_fwd_k", 8)])
def test_mfma_kernels_do_not_spill(src, needle, count):
    """(counts: 8 epilogues x {per-tile, persistent} bf16 GEMMs + the 6 forward epilogues x 2 of the fp16-operand policies
    -- STORE, ROPE, RESIDUAL, SWIGLU, SWIGLU_STACKED (round 6: GradCache pass 1), RESIDUAL_F32; 4 bf16 attention
    forwards + the 4 fp16 ones: bidirectional and -- round 6 -- causal, padded and packed)"""
    ks = [k for k in _resources(src) if needle in k["name"]]
    assert len(ks) == count, [k["name"] for k in ks]
    for k in ks:
        assert int(k["VGPRs Spill"]) == 0 and int(k["ScratchSize [bytes/lane]"]) == 0, k
        assert int(k["VGPRs"]) + int(k["AGPRs"]) <= 256 and int(k["Occupancy [waves/SIMD]"]) >= 2, k


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("inst,mfma", [("Li1ELb1ELb0E", "v_mfma_f32_16x16x32_bf16"), ("Li7ELb1ELb1E", "v_mfma_f32_16x16x32_f16")])
def test_gemm_k_loop_hands_the_matrix_pipe_over_one_product_early(inst, mfma):
    """ISA-level guard of the round-4 hand-over (csrc/gemm_bf16.hip, GRIT_GEMM_BAR_EARLY = 1): in the persistent RESIDUAL instantiation
    every MFMA segment of the K loop is 15 products, s_barrier, ONE product, s_setprio 0 -- hipcc must neither move products across the
    barrier nor merge segments (a barrier behind the last product costs 1-2.6 % per shape, two products early 6-8 %:
    profiles/r04_gemm_barrier_ab.log) -- and no s_waitcnt vmcnt(0) (a drain of the LDS-DMA queue) sits between two segments of a K-tile."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out,
                            os.path.join(ROOT, "gritlm_amd", "csrc", "gemm_bf16.hip")], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = open(out).read().split("\n")
    # (also for the fp16-operand instantiation of round 5 -- persistent RESIDUAL_F32 -- whose K loop must be the same stream, one opcode changed)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN4grit14gemm_bf16_nt_kI" + inst + r"\w+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = [m.group(1) + (" " + m.group(2) if m.group(1) in ("s_setprio", "s_waitcnt") else "")
           for m in (re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l) for l in lines[start:end]) if m]
    # segments: maximal runs between an `s_setprio 1` and the next `s_setprio 0`
    segs, cur = [], None
    for op in ops:
        if op.startswith("s_setprio 1"):
            cur = []
        elif op.startswith("s_setprio 0") and cur is not None:
            segs.append(cur); cur = None
        elif cur is not None:
            cur.append(op)
    assert all(o.startswith(mfma) for sg in segs for o in sg if o.startswith("v_mfma")), "wrong MFMA opcode in the K loop"
    full = [s for s in segs if sum(o.startswith("v_mfma") for o in s) == 16]
    assert len(full) >= 16, (len(segs), len(full))                  # 4 segments per K-tile form, several forms (first / middle / last K-tile, two buffers)
    for s in full:
        assert s.count("s_barrier") == 1, s
        b = s.index("s_barrier")
        assert sum(o.startswith("v_mfma") for o in s[:b]) == 15 and sum(o.startswith("v_mfma") for o in s[b:]) == 1, s
        assert not any(o.startswith("s_waitcnt") and "vmcnt(0)" in o for o in s), s
