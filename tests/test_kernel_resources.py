"""Compile-time guard (no GPU): hipcc's resource-usage remarks for the two MFMA kernels.

A register spill inside these kernels is not a small cost: the reload is a scratch load, scratch loads share the vmcnt counter with the
hand-placed LDS-DMA stream, and hipcc waits for them with ``s_waitcnt vmcnt(0)`` -- one spilled accumulator fragment in the persistent
GEMM's first K-tile drained the DMA queue once per tile and cost the K = 14336 launches the whole gain of the persistent form
("what the compiler left in the persistent loop").  So: zero VGPR spills and zero scratch in every instantiation, and the
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
def test_w64_attention_uses_the_whole_register_file_without_spilling():
    """The W64 attention forward runs ONE wave per SIMD on purpose (64 query rows per wave: oacc 128 + scores 64 in AGPRs, Q in AGPRs, the
    softmax's copy of the scores, P and the fragments in arch VGPRs): <= 512 registers, no spill, no scratch -- a scratch reload's vmcnt(0)
    would drain the LDS-DMA queue in the middle of the tile loop."""
    ks = [k for k in _resources("attention.hip") if "attn_fwd_w64_k" in k["name"]]
    assert len(ks) == 2, [k["name"] for k in ks]
    for k in ks:
        assert int(k["VGPRs Spill"]) == 0 and int(k["ScratchSize [bytes/lane]"]) == 0, k
        assert int(k["VGPRs"]) <= 256 and int(k["AGPRs"]) <= 256 and int(k["Occupancy [waves/SIMD]"]) >= 1, k


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,needle,count", [("gemm_bf16.hip", "gemm_bf16_nt_k", 16), ("attention.hip", "attn_bidir_fwd_k", 4)])
def test_mfma_kernels_do_not_spill(src, needle, count):
    ks = [k for k in _resources(src) if needle in k["name"]]
    assert len(ks) == count, [k["name"] for k in ks]
    for k in ks:
        assert int(k["VGPRs Spill"]) == 0 and int(k["ScratchSize [bytes/lane]"]) == 0, k
        assert int(k["VGPRs"]) + int(k["AGPRs"]) <= 256 and int(k["Occupancy [waves/SIMD]"]) >= 2, k
