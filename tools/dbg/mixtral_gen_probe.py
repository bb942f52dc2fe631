import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
import gpu_checks as G, synth
from gritlm_amd.training import GritLMTrainModel
from transformers import MixtralForCausalLM
DEV = G.DEV
g = np.load("tests/golden/generative_moe-tiny.npz")
I = 512
with tempfile.TemporaryDirectory() as td:
    d16 = synth.build_mixtral_dir(os.path.join(td, "m16"), "moe-tiny", 0, "bfloat16")
    hf = MixtralForCausalLM.from_pretrained(d16, torch_dtype=torch.float32).to(DEV).train()
ids = torch.from_numpy(g["input_ids"]).to(DEV); mask = torch.from_numpy(g["attention_mask"]).to(DEV); labels = torch.from_numpy(g["labels"]).to(DEV)
logits = hf(input_ids=ids, attention_mask=mask).logits.float()
sl, sb = logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1)
loss = torch.nn.functional.cross_entropy(sl, sb, reduction="sum", ignore_index=-100) / ids.shape[0] * 0.25
loss.backward()
print("stock HF fp32 loss", loss.item(), "fixture no-aux loss", float(g["loss_noaux"]))
sd = dict(hf.named_parameters())
def stock(n):
    if "block_sparse_moe" not in n:
        return sd[n].grad.float().cpu().numpy()
    pre, rest = n.split(".block_sparse_moe.")
    if rest == "gate.weight":
        return sd[pre + ".mlp.gate.weight"].grad.float().cpu().numpy()
    _, e, w, _ = rest.split("."); e = int(e)
    if w == "w2":
        return sd[pre + ".mlp.experts.down_proj"].grad[e].float().cpu().numpy()
    gu = sd[pre + ".mlp.experts.gate_up_proj"].grad[e]
    return (gu[:I] if w == "w1" else gu[I:]).float().cpu().numpy()
for k in g.files:
    if k.startswith("grad_noaux/"):
        n = k[len("grad_noaux/"):]
        ref = g[k]
        print(f"  stock-HF-fp32 vs fixture(no aux) {n:55s} rel {np.linalg.norm(stock(n)-ref)/np.linalg.norm(ref):.3e}")

# ---- per-layer d loss / d hidden_states: native (recorded rmsnorm_bwd outputs) vs stock HF fp32
hf.zero_grad()
o = hf(input_ids=ids, attention_mask=mask, output_hidden_states=True)
hs = o.hidden_states
for h in hs:
    h.retain_grad()
logits = o.logits.float()
sl = logits[:, :-1].reshape(-1, logits.shape[-1])
loss = torch.nn.functional.cross_entropy(sl, sb, reduction="sum", ignore_index=-100) / ids.shape[0] * 0.25
loss.backward()
keep = mask.reshape(-1) != 0
ref = [h.grad.reshape(-1, h.shape[-1])[keep].float() for h in hs]      # hs[0] embed out, hs[1] after L0, hs[2] after L1 (pre final norm)

with tempfile.TemporaryDirectory() as td:
    d16 = synth.build_mixtral_dir(os.path.join(td, "mixtral-tiny"), "moe-tiny", 0, "bfloat16")
    m = GritLMTrainModel(model_name_or_path=d16, mode="unified", pooling_method="mean", normalized=True, attn="bbcc", temperature=0.02,
                         negatives_cross_device=False, loss_gen_type="token", loss_gen_factor=0.25, device="cuda", torch_dtype=torch.bfloat16)
    m.model.config.router_aux_loss_coef = 0.0
    m.enable_native()
from gritlm_amd import ops
rec = []
orig = ops.rmsnorm_bwd
def spy(*a, **k):
    r = orig(*a, **k)
    rec.append(r.float().clone())
    return r
ops.rmsnorm_bwd = spy
import gritlm_amd.training.engine as E
o2 = m(generative={"input_ids": ids, "attention_mask": mask, "labels": labels})
o2.loss.backward()
ops.rmsnorm_bwd = orig
rel = lambda a, b: float((a - b).norm() / b.norm())
print("native loss", float(o2.loss_gen.item()))
print("recorded", len(rec), [tuple(r.shape) for r in rec])
print("g(hs2) final-norm bwd :", rel(rec[0], ref[2]))
print("g(hs1) after layer 1  :", rel(rec[2], ref[1]))
print("g(hs0) after layer 0  :", rel(rec[4], ref[0]))
# per-token error profile of g(hs1)
e = ((rec[2] - ref[1]).norm(dim=1) / ref[1].norm(dim=1))
print("per-token rel err g(hs1): max", float(e.max()), "median", float(e.median()), "n>0.1:", int((e > 0.1).sum()), "of", e.numel())
bad = (e > 0.1).nonzero().squeeze(-1).tolist()
print("bad tokens", bad[:40])
cu = np.concatenate([[0], np.cumsum(g["attention_mask"].sum(1))])
print("cu", cu.tolist())

# which tokens route differently from the reference's fp32 run?
eng = m.train_engine
eng._router_log = []
with torch.no_grad():
    eng.forward(ids, mask, save=False, packed=True, causal=True)
log, eng._router_log = eng._router_log, None
keepn = g["attention_mask"].reshape(-1) != 0
for li, (lg, ex) in enumerate(log):
    refr = np.sort(g["routing"][li][keepn], axis=-1)
    mine = np.sort(ex.cpu().numpy(), axis=-1)
    flips = np.nonzero(~(mine == refr).all(-1))[0]
    print("layer", li, "flipped tokens", flips.tolist())
    p = torch.softmax(lg, -1)
    top3 = torch.topk(p, 3, dim=-1)[0]
    for t in flips.tolist():
        print("    token", t, "top-3 probs", [round(float(v), 4) for v in top3[t]])
e0 = ((rec[4] - ref[0]).norm(dim=1) / ref[0].norm(dim=1))
print("g(hs0) bad tokens", (e0 > 0.1).nonzero().squeeze(-1).tolist()[:60])
em = ((rec[1] - 0).norm(dim=1))
