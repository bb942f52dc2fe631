#!/usr/bin/env python3
"""Host-inclusive encode throughput: GritLM.encode() from Python strings (WordLevel tokenizer on the host, H2D, native engine at the
GritLM-7B shape with random weights, fused pool/normalise, one D2H) -- the number a caller of the drop-in API sees."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from gritlm_amd import GritLM  # noqa: E402
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine  # noqa: E402
from transformers import AutoTokenizer  # noqa: E402

layers = int(os.environ.get("LAYERS", 32))
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=8,
                    vocab_size=32000)
m = GritLM.__new__(GritLM)
torch.nn.Module.__init__(m)
with tempfile.TemporaryDirectory() as td:
    synth.make_tokenizer(td)
    m.tokenizer = AutoTokenizer.from_pretrained(td, padding_side="right")
m.engine = MistralEncoderEngine.random_init(cfg, "cuda", seed=0)
m.model = torch.nn.Module(); m.model.dtype = torch.bfloat16
m.embedding_attr, m.projection, m.normalized, m.pooling_method, m.attn = None, None, True, "mean", "bbcc"
m.device, m.num_gpus, m.embed_eos = "cuda", 1, ""

for name, (lo, hi) in {"full (every doc truncated to 512 tokens)": (600, 700), "ragged (U{64..512} tokens)": (63, 511)}.items():
    docs = synth.make_sentences(1024, seed=3, min_words=lo, max_words=hi)
    m.encode(docs[:256], batch_size=256, max_length=512)            # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = m.encode(docs, batch_size=256, max_length=512)
    dt = time.perf_counter() - t0
    assert e.shape == (1024, 4096) and np.isfinite(e).all()
    print(f"{name}: {len(docs) / dt:.1f} docs/s host-inclusive ({layers} layers)", flush=True)
