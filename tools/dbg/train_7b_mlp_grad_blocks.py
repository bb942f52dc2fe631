#!/usr/bin/env python3
"""Native step vs the Hugging Face module (fp32, same GPU, same weights) at the 7B layer shape: where in the MLP weight gradients does the
excess gradient norm sit?  Prints the worst 64-wide slices along the intermediate dimension.  (GPU)"""
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
grads = {}
with tempfile.TemporaryDirectory() as td:
    d16 = synth.build_mistral_dir(os.path.join(td, "m16"), "7b-l1", 0, "bfloat16")
    d32 = synth.build_mistral_dir(os.path.join(td, "m32"), "7b-l1", 0, "float32")
    for tag, d, dt, native in (("ref", d32, torch.float32, False), ("hip", d16, torch.bfloat16, True)):
        m = GritLMTrainModel(model_name_or_path=d, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc",
                             temperature=float(g["tau"]), negatives_cross_device=False, device="cuda", torch_dtype=dt)
        if native:
            m.enable_native()
        else:
            m.model.to(DEV)
        o = m(query=dict(q), passage=dict(p))
        o.loss.backward()
        print(tag, "loss", float(o.loss))
        grads[tag] = {n: t.grad.float().clone() for n, t in m._backbone().named_parameters() if "mlp" in n or "o_proj" in n}
        del m
        torch.cuda.empty_cache()
for n, r in grads["ref"].items():
    h = grads["hip"][n]
    fp = float(g["gnorm/" + n])
    print(f"{n}: ref(gpu fp32) norm {float(r.norm()):.5f} fixture {fp:.5f} hip {float(h.norm()):.5f}  rel_l2 {float((h - r).norm() / r.norm()):.3e}")
    axis = 0 if r.shape[0] >= r.shape[1] else 1          # the intermediate dimension
    e = (h - r).pow(2).sum(dim=1 - axis); b = r.pow(2).sum(dim=1 - axis)
    eb, bb = e.view(-1, 64).sum(1), b.view(-1, 64).sum(1)
    rel = (eb / bb).sqrt()
    top = torch.topk(rel, 6)
    print("   worst 64-slices along dim", axis, [(int(i), f"{float(v):.3f}") for v, i in zip(top.values, top.indices)], " median", f"{float(rel.median()):.4f}")
    # energy ratio per slice hip/ref
    hb = h.pow(2).sum(dim=1 - axis).view(-1, 64).sum(1)
    ratio = (hb / bb).sqrt()
    top = torch.topk(ratio, 6)
    print("   largest norm ratios hip/ref per slice", [(int(i), f"{float(v):.3f}") for v, i in zip(top.values, top.indices)], " median", f"{float(ratio.median()):.4f}")
