#!/usr/bin/env python3
"""Precision of the encode path as a function of depth (VERDICT r03 "next round" #1 b/d/e).

For depth d in 1, 2, 4, 8, 16, 32 (the first d decoder layers + the final norm + mean pooling + L2 normalise) and 32 documents x 512
tokens, `1 - cos` of the pooled embeddings against the REFERENCE-EQUIVALENT MODULE IN FP32 ON THIS GPU, loaded from the same weights
(oracle/torch_reference.py: the stock transformers module, pinned bit-for-bit on the reference-generated fixtures), for

  engine_bf16_residual   the HIP engine, residual stream in bf16 (the reference's bf16 arithmetic; the default until round 4)
  engine_fp32_residual   the HIP engine, residual stream in fp32 (GRIT_EPI_RESIDUAL_F32 / grit_rmsnorm_fwd_f32in)
  engine_f16_operands    the HIP engine, fp32 residual stream + fp16 MFMA operands (round 5: the policy held to the north-star's 1e-4)
  stock_bf16_reference_mask   the stock module in bf16 the way the reference drives SDPA: NO mask for an all-valid batch
  stock_bf16_explicit_mask    the stock module in bf16 with the explicit 4-D additive mask (what rounds 1-3 used as the yardstick)

on two models: `bench` = bench.py's model (7B shape, DISTINCT N(0, 0.02) weights per layer, bench.py's own batch) and `fixture` = the
repeated-layer model of tests/golden/encoder_7b-depth32.npz (reference-generated; its `full` / `ragged` cases are also compared with the
fixture's own fp32 and bf16 embeddings).  Then the cost: docs/s of both engine modes at 256 x 512.

    python tools/depth_parity.py [--out gpurun_out/depth_parity.json] [--docs 32] [--no-timing]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import torch_reference as TR  # noqa: E402
from gritlm_amd import ops  # noqa: E402
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine  # noqa: E402

DEPTHS = (1, 2, 4, 8, 16, 32)


def omc(a: torch.Tensor, b: torch.Tensor) -> dict:
    d = 1.0 - torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=1)
    return {"max": float(d.max()), "mean": float(d.mean())}


@torch.no_grad()
def engine_emb(eng, ids, mask, depth, policy):
    layers = eng.layers
    eng.layers = layers[:depth]
    eng.set_precision({False: "bf16", True: "fp32_residual"}.get(policy, policy))
    try:
        return ops.pool_norm(eng.forward(ids, mask, borrow=True), mask, "mean", True).clone()
    finally:
        eng.layers = layers


def curve(eng, hf_sd, cfgd, ids, mask, dev, depths=DEPTHS, chunk=8):
    """1 - cos against the fp32 stock module for the four implementations, per depth."""
    out = {k: {} for k in ("engine_bf16_residual", "engine_fp32_residual", "engine_f16_operands", "engine_f16_stream", "stock_bf16_reference_mask",
                           "stock_bf16_explicit_mask")}
    f32m = TR.build_model(cfgd, torch.float32, dev, state_dict=hf_sd)
    ref = {}
    for d in depths:                       # fp32 reference, `chunk` documents at a time (fp32 sdpa on the explicit mask is memory-hungry)
        ref[d] = torch.cat([TR.encode(f32m, ids[i:i + chunk], mask[i:i + chunk], layers=d) for i in range(0, ids.shape[0], chunk)])
    del f32m
    torch.cuda.empty_cache()
    b16 = TR.build_model(cfgd, torch.bfloat16, dev, state_dict=hf_sd)
    for d in depths:
        for rule, key in (("reference", "stock_bf16_reference_mask"), ("explicit", "stock_bf16_explicit_mask")):
            e = torch.cat([TR.encode(b16, ids[i:i + chunk], mask[i:i + chunk], mask_rule=rule, layers=d) for i in range(0, ids.shape[0], chunk)])
            out[key][str(d)] = omc(e, ref[d])
    del b16
    torch.cuda.empty_cache()
    for d in depths:
        out["engine_bf16_residual"][str(d)] = omc(engine_emb(eng, ids, mask, d, False), ref[d])
        out["engine_fp32_residual"][str(d)] = omc(engine_emb(eng, ids, mask, d, True), ref[d])
        out["engine_f16_operands"][str(d)] = omc(engine_emb(eng, ids, mask, d, "f16_operands"), ref[d])
        out["engine_f16_stream"][str(d)] = omc(engine_emb(eng, ids, mask, d, "f16_stream"), ref[d])
    eng.set_precision("bf16")
    return out, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "depth_parity.json"))
    ap.add_argument("--docs", type=int, default=32)
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--no-bench-model", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = {"what": __doc__.split("\n\n")[0], "docs": args.docs, "seq": 512, "gpu": torch.cuda.get_device_name(0)}

    # ---- the fixture's model (reference-generated embeddings: tests/golden/encoder_7b-depth32.npz)
    import bench
    g = np.load(os.path.join(ROOT, "tests", "golden", "encoder_7b-depth32.npz"))
    cfg, w, ids_np, mask_np = bench.oracle_full_depth_case(sample_docs=args.docs, seq=512, layers=32)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    eng = MistralEncoderEngine.from_state_dict(EncoderConfig.from_dict(cfg), sd, dev)
    ids, mask = torch.from_numpy(ids_np).to(dev), torch.from_numpy(mask_np).to(dev)
    t0 = time.perf_counter()
    res["fixture_model"], _ = curve(eng, sd, cfg, ids, mask, dev)
    res["fixture_model"]["seconds"] = time.perf_counter() - t0
    fx = {}
    for tag in ("full", "ragged"):
        fi, fm = torch.from_numpy(g[f"{tag}_input_ids"]).to(dev), torch.from_numpy(g[f"{tag}_attention_mask"]).to(dev)
        r32, rb = torch.from_numpy(g[f"{tag}_emb"]).to(dev), torch.from_numpy(g[f"{tag}_emb_bf16"]).to(dev)
        fx[tag] = {"reference_bf16_cpu_vs_reference_fp32": omc(rb, r32)}
        for name, hp in (("engine_bf16_residual", "bf16"), ("engine_fp32_residual", "fp32_residual"), ("engine_f16_operands", "f16_operands"), ("engine_f16_stream", "f16_stream")):
            eng.set_precision(hp)
            e_pad = eng.encode_pooled(fi, fm, "mean", True, packed=False)
            e_pack = eng.encode_pooled(fi, fm, "mean", True, packed=True)
            fx[tag][name + "_vs_reference_fp32"] = omc(e_pad, r32)
            fx[tag][name + "_vs_reference_bf16"] = omc(e_pad, rb)
            fx[tag][name + "_packed_identical"] = bool(torch.equal(e_pad, e_pack))
    res["fixture_cases_vs_reference_generated_embeddings"] = fx
    del eng
    torch.cuda.empty_cache()
    print(json.dumps(res["fixture_cases_vs_reference_generated_embeddings"], indent=1), flush=True)

    # ---- bench.py's model and batch
    if not args.no_bench_model:
        cfgb = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
                             vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0)
        eng = MistralEncoderEngine.random_init(cfgb, dev, seed=0)
        gen = torch.Generator(device=dev).manual_seed(1234)
        ids_all = torch.randint(3, cfgb.vocab_size, (256, 512), generator=gen, device=dev, dtype=torch.int64)
        mask_all = torch.ones((256, 512), dtype=torch.int64, device=dev)
        ids, mask = ids_all[:args.docs].contiguous(), mask_all[:args.docs].contiguous()
        t0 = time.perf_counter()
        res["bench_model"], _ = curve(eng, eng.to_hf_state_dict(), dict(TR.SHAPE_7B, num_hidden_layers=32), ids, mask, dev)
        res["bench_model"]["seconds"] = time.perf_counter() - t0
        print(json.dumps({k: v.get("32") for k, v in res["bench_model"].items() if isinstance(v, dict)}, indent=1), flush=True)
        if not args.no_timing:
            tim = {}
            for name, hp in (("bf16_residual", "bf16"), ("fp32_residual", "fp32_residual"), ("f16_operands", "f16_operands"), ("f16_stream", "f16_stream"),
                             ("bf16_residual_again", "bf16"), ("fp32_residual_again", "fp32_residual"), ("f16_operands_again", "f16_operands"),
                             ("f16_stream_again", "f16_stream")):
                eng.set_precision(hp)
                for _ in range(2):
                    ops.pool_norm(eng.forward(ids_all, mask_all, borrow=True), mask_all, "mean", True)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    ops.pool_norm(eng.forward(ids_all, mask_all, borrow=True), mask_all, "mean", True)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                tim[name] = {"docs_per_s_median": 256 / sorted(ts)[len(ts) // 2], "docs_per_s_best": 256 / min(ts)}
            res["encode_256x512_docs_per_s"] = tim
            print(json.dumps(tim, indent=1), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
