#!/usr/bin/env python3
"""Freeze every number the GPU checks take from the reference's OWN bf16 runs (the `*_bf16` arrays of the fixtures in this directory)
into tests/golden/numeric_bounds.json.

Why: those arrays come from the reference's bf16 CPU run and depend on the generating host's bf16 GEMM path (the fp32 arrays do not:
they regenerate bit-identically anywhere).  A check of the form `ours < 1.25 x reference-bf16-error` therefore moved 5-10 % with the
CPU that produced the fixture.  The checks now read the yardsticks from the JSON written here -- numbers, versioned with the host that
produced them -- and tests/test_oracle_golden.py::test_numeric_bounds_match_committed_fixtures fails if a re-generated fixture drifts
from them (so a drift is a visible decision, not a silently moved tolerance).

    python tests/golden/make_bounds.py          # needs only the committed .npz files (no reference, no GPU)
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TRAIN_NAMES = ["layers.0.self_attn.q_proj.weight", "layers.1.mlp.down_proj.weight", "norm.weight", "layers.0.input_layernorm.weight",
               "layers.1.self_attn.v_proj.weight", "embed_tokens.weight"]          # tests/gpu_checks.py::_NAMES


def compute() -> dict:
    Y = {}
    ld = lambda n: np.load(os.path.join(HERE, n))
    rel_all = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    omc = lambda a, b: float(np.max(1 - np.sum(a * b, axis=1)))
    for cfg in ("tiny", "gqa", "moe-tiny", "moe-gqa"):
        g = ld(f"encoder_{cfg}.npz")
        valid = g["attention_mask"].astype(bool)
        Y[f"encoder_{cfg}/rel_refbf16_vs_fp32"] = float(np.linalg.norm((g["last_hidden_state_bf16"] - g["last_hidden_state"])[valid]) /
                                                        np.linalg.norm(g["last_hidden_state"][valid]))
        for m in ("mean", "weightedmean"):
            r, rb = g[f"emb_{m}"], g[f"emb_{m}_bf16"]
            Y[f"encoder_{cfg}/pair_delta_of_bf16ref_{m}"] = float(np.max(np.abs(rb @ rb.T - r @ r.T)))
            Y[f"encoder_{cfg}/1-cos_of_bf16ref_{m}"] = omc(rb, r)
        if "routing" in g.files:
            Y[f"encoder_{cfg}/routing_agree_of_bf16_ref"] = float((np.sort(g["routing_bf16"], axis=-1) == np.sort(g["routing"], axis=-1)).all(-1)[:, valid].mean())
    g = ld("encoder_7b-l1.npz")
    Y["encoder_7b-l1/rel_refbf16_vs_fp32"] = rel_all(g["probe_hidden_bf16"], g["probe_hidden"])
    g = ld("gritlm_encode.npz")
    for key in ("mean_instr", "mean"):
        r32, r16 = g[f"mistral_fp32_{key}"], g[f"mistral_bf16_{key}"]
        Y[f"gritlm_encode/pair_delta_of_bf16ref_{key}"] = float(np.max(np.abs(r16 @ r16.T - r32 @ r32.T)))
    g = ld("gradcache_tiny.npz")
    for key in ("direct", "gradcache"):
        Y[f"gradcache_tiny/loss_gap_bf16_{key}"] = abs(float(g[f"loss_{key}_bf16"]) - float(g[f"loss_{key}"]))
        for n in TRAIN_NAMES:
            ref, ref16 = g[f"grad_{key}/" + n], g[f"grad_{key}_bf16/" + n]
            Y[f"gradcache_tiny/grad_rel_bf16_{key}/{n}"] = float(np.linalg.norm(ref16 - ref) / (np.linalg.norm(ref) + 1e-20))
    for fx in ("train_7b-l1", "train_7b-d8"):              # (d8: eight distinct layers, 16 queries -- round 6)
        g = ld(fx + ".npz")
        Y[f"{fx}/loss_gap_bf16"] = abs(float(g["loss_bf16"]) - float(g["loss"]))
        for k in g.files:
            if k.startswith("gnorm/"):
                n = k[len("gnorm/"):]
                ref_n, ref_n16 = float(g["gnorm/" + n]), float(g["gnorm_bf16/" + n])
                Y[f"{fx}/gnorm_rel_bf16/{n}"] = abs(ref_n16 - ref_n) / (ref_n + 1e-20)
                ref, ref16 = g["probe/" + n], g["probe_bf16/" + n]
                Y[f"{fx}/probe_rel_bf16/{n}"] = float(np.linalg.norm(ref16 - ref) / (np.linalg.norm(ref) + 1e-20))
    return Y


def host() -> str:
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cpu


if __name__ == "__main__":
    out = {"what": "yardsticks derived from the reference's own bf16 runs stored in the fixtures of this directory, frozen (see make_bounds.py)",
           "fixtures_generated_on": "the build container's host (rounds 1-4; same image and CPU model as recorded here)",
           "frozen_on": host(), "values": compute()}
    json.dump(out, open(os.path.join(HERE, "numeric_bounds.json"), "w"), indent=1, sort_keys=True)
    print(len(out["values"]), "values ->", os.path.join(HERE, "numeric_bounds.json"))
