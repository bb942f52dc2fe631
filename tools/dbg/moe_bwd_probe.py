import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
import gpu_checks as G, synth
from gritlm_amd.training import GritLMTrainModel
DEV = G.DEV
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as td:
    d16 = synth.build_mixtral_dir(os.path.join(td, "m16"), "moe-tiny", 0, "bfloat16")
    m = GritLMTrainModel(model_name_or_path=d16, mode="embedding", pooling_method="mean", normalized=True, attn="bbcc", temperature=0.02,
                         negatives_cross_device=False, device="cuda", torch_dtype=torch.bfloat16)
    m.enable_native()
eng = m.train_engine
L = eng.layers[0]
T, H, I, E = 200, eng.cfg.hidden_size, eng.cfg.intermediate_size, eng.cfg.num_local_experts
x2 = (torch.randn((T, H), device=DEV) * 1.0).to(torch.bfloat16)
h_mid = torch.zeros((T, H), device=DEV, dtype=torch.bfloat16)
dh = (torch.randn((T, H), device=DEV) * 1.0).to(torch.bfloat16)
eng.prepare_grads()
buf = eng._layer_buffers(T, with_gu=True)
h_out = torch.empty((T, H), device=DEV, dtype=torch.bfloat16)
sv = eng._mlp_fwd(L, h_mid, x2, buf, True, h_out)
sv["x2"] = x2
for p in (L.wgu, L.wdown, L.wgate):
    p.grad.zero_()
dx2 = eng._mlp_bwd(0, L, sv, dh)
# torch fp32 reference with the same routing
xf = x2.float().requires_grad_(True)
wgu, wdn, wg = (p.detach().float().requires_grad_(True) for p in (L.wgu, L.wdown, L.wgate))
p_ = torch.softmax(xf @ wg.t(), dim=-1)
ex = sv["experts"].long()
sel = torch.gather(p_, 1, ex)
w = sel / sel.sum(-1, keepdim=True)
out = torch.zeros((T, H), device=DEV)
for k in range(2):
    for e in range(E):
        idx = (ex[:, k] == e).nonzero().squeeze(-1)
        if idx.numel():
            xe = xf[idx]
            gu = xe @ wgu[e].t()
            act = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
            out = out.index_add(0, idx, (act @ wdn[e].t()) * w[idx, k:k + 1])
print("fwd rel", float((h_out.float() - out).norm() / out.norm()))
(out * dh.float()).sum().backward()
rel = lambda a, b: float((a.float() - b).norm() / (b.norm() + 1e-20))
print("dx2", rel(dx2, xf.grad), "dWg", rel(L.wgate.grad, wg.grad), "dW13", rel(L.wgu.grad, wgu.grad), "dW2", rel(L.wdown.grad, wdn.grad))
for e in range(E):
    print("  expert", e, "n", int(sv["counts"][e]), "dW13", rel(L.wgu.grad[e], wgu.grad[e]), "dW2", rel(L.wdown.grad[e], wdn.grad[e]))
# split dx2 into the expert path and the router path
