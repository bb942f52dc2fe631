#!/usr/bin/env python3
"""Per-parameter signed gradient-norm error of the native step at the 7B layer shape vs the reference's fp32 run (and the reference's own
bf16 run): is the error noise (norm error ~ 0) or a systematic scale?   python tools/dbg/train_7b_norm_probe.py  (GPU)"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import synth
from gritlm_amd.training import GritLMTrainModel
DEV = "cuda"
g = np.load(os.path.join(ROOT, "tests", "golden", "train_7b-l1.npz"))
q = {"input_ids": torch.from_numpy(g["q_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["q_mask"]).to(DEV)}
p = {"input_ids": torch.from_numpy(g["p_ids"]).to(DEV), "attention_mask": torch.from_numpy(g["p_mask"]).to(DEV)}
with tempfile.TemporaryDirectory() as td:
    d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "7b-l1", 0, "bfloat16")
    for packed in (True, False):
        m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=float(g["tau"]), negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
        m.enable_native()
        m.native_packed = packed
        o = m(query=dict(q), passage=dict(p))
        o.loss.backward()
        print(f"packed={packed} loss {float(o.loss):.6f} (fp32 ref {float(g['loss']):.6f}, bf16 ref {float(g['loss_bf16']):.6f})")
        for n, t in m._backbone().named_parameters():
            got = t.grad.float()
            r, r16 = float(g["gnorm/" + n]), float(g["gnorm_bf16/" + n])
            ref = g["probe/" + n]
            if n == "embed_tokens.weight":
                gp = got[torch.from_numpy(g["probe_rows/" + n]).to(DEV)].cpu().numpy()
            elif got.dim() == 2:
                gp = got[:8].cpu().numpy()
            else:
                gp = got.cpu().numpy()
            # projection of the error on the reference direction: systematic scale = <gp, ref>/<ref, ref> - 1
            sc = float((gp * ref).sum() / (ref * ref).sum()) - 1
            print(f"  {n:45s} norm {float(got.norm()):.5f} ref {r:.5f} signed_rel {(float(got.norm()) - r) / r:+.2e} (ref bf16 {(r16 - r) / r:+.2e})  "
                  f"probe rel_l2 {np.linalg.norm(gp - ref) / np.linalg.norm(ref):.2e} probe_scale {sc:+.2e}")
        del m
        torch.cuda.empty_cache()
