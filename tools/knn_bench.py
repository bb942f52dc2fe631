#!/usr/bin/env python3
"""Index search timing: Q queries against N fp32 embeddings of width 4096 (rag/index.py:97-104).  python tools/knn_bench.py [--n 1000000]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd.rag import DenseIndex  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000); ap.add_argument("--q", type=int, default=32); ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
dev = "cuda"
emb = torch.nn.functional.normalize(torch.randn((a.n, 4096), device=dev), dim=-1)
idx = DenseIndex.from_rows(emb)
q = torch.nn.functional.normalize(torch.randn((a.q, 4096), device=dev), dim=-1)
s, i = idx.search_knn(q, a.k); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    s, i = idx.search_knn(q, a.k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    rs, ri = torch.topk(q @ emb.t(), a.k, dim=1)
torch.cuda.synchronize()
dt_ref = (time.perf_counter() - t0) / 3
print(json.dumps({"metric": "index search ms", "n_docs": a.n, "queries": a.q, "k": a.k, "ms": dt * 1e3, "embedding_gb": a.n * 4096 * 4 / 1e9,
                  "hbm_roofline_ms": a.n * 4096 * 4 / 8e12 * 1e3, "torch_matmul_topk_ms": dt_ref * 1e3, "same_top1": float((i[:, 0] == ri[:, 0]).float().mean())}))
