#!/usr/bin/env python3
"""Golden files for gritlm_amd.rag.DistributedIndex / load_passages, produced by the REFERENCE's rag/index.py (imported from
/root/reference, CPU): a 3-shard saved index (the reference's own file format), the passages its loader reads from a JSONL file
(section -> title merge, an empty line), and the result of its search_knn on fixed queries.
    python tests/golden/make_rag_index_golden.py          (writes tests/golden/rag_index/)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from rag.index import DistributedIndex, load_passages  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rag_index")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(20240924)
N, DIM, Q, K = 11, 16, 4, 3
lines = []
for i in range(N):
    item = {"id": str(i), "title": f"Title {i}", "text": " ".join(f"w{(i * 7 + j) % 23}" for j in range(5 + i % 4))}
    if i % 3 == 1:
        item["section"] = f"Section {i}"
    if i == 5:
        item["section"] = ""                     # empty section: the title stays as it is
    lines.append(json.dumps(item))
lines.insert(4, "")                              # an empty line: the reference appends None for it
with open(os.path.join(OUT, "passages.jsonl"), "w") as f:
    f.write("\n".join(lines) + "\n")
loaded = load_passages([os.path.join(OUT, "passages.jsonl")])
loaded_max7 = load_passages([os.path.join(OUT, "passages.jsonl")], maxload=7)
passages = [p for p in loaded if p is not None]
index = DistributedIndex(dtype=torch.float32)
index.init_embeddings(passages, DIM)
emb = rng.standard_normal((DIM, len(passages))).astype(np.float32)
index.embeddings[:, :] = torch.from_numpy(emb)
for stale in os.listdir(OUT):
    if stale.endswith(".pt"):
        os.remove(os.path.join(OUT, stale))
index.save_index(OUT, 3)
queries = rng.standard_normal((Q, DIM)).astype(np.float32)
docs, scores = index.search_knn(torch.from_numpy(queries), K)
json.dump({"loaded": loaded, "loaded_maxload7": loaded_max7, "n_passages": len(passages), "dim": DIM, "shards": 3,
           "queries": queries.tolist(), "topk": K, "docs_ids": [[d["id"] for d in row] for row in docs], "scores": scores,
           "embeddings": emb.tolist()}, open(os.path.join(OUT, "expected.json"), "w"), indent=0)
print("wrote", sorted(os.listdir(OUT)))
