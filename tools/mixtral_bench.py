#!/usr/bin/env python3
"""Mixtral-8x7B-shape (GritLM-8x7B) bidirectional encode on one MI355X: docs/s at 256 docs x 512 tokens, random-init weights
generated on the device (90 GB bf16).  SURVEY §8 f1.  python tools/mixtral_bench.py [--layers 32] [--docs 256] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gritlm_amd import ops  # noqa: E402
from gritlm_amd.encoder import EncoderConfig, MistralEncoderEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--docs", type=int, default=256)
ap.add_argument("--seq", type=int, default=512)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--parity-docs", type=int, default=4, help="documents of the timed batch re-computed in fp32 (layer-streamed) for the parity datum; 0 = skip")
ap.add_argument("--yardstick-docs", type=int, default=2, help="of those, documents also run through the reference's data flow in bf16 (the end-to-end yardstick)")
ap.add_argument("--no-f16", action="store_true", help="skip the f16_operands policy (timing and parity): bf16 only")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = EncoderConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=a.layers, num_attention_heads=32, num_key_value_heads=8,
                    vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_local_experts=8, num_experts_per_tok=2)
t0 = time.perf_counter()
eng = MistralEncoderEngine.random_init(cfg, dev, seed=0)
torch.cuda.synchronize()
t_init = time.perf_counter() - t0
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(3, cfg.vocab_size, (a.docs, a.seq), generator=gen, device=dev)
mask = torch.ones((a.docs, a.seq), dtype=torch.int64, device=dev)
for _ in range(a.warmup):
    eng.encode_pooled(ids, mask, "mean", True)
torch.cuda.synchronize()
timer = ops.KernelTimer()
ops.set_timer(timer)
eng.record_routing = []
t0 = time.perf_counter()
for _ in range(a.steps):
    e = eng.encode_pooled(ids, mask, "mean", True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ops.set_timer(None)
ks = timer.summary()
counts = torch.stack([torch.bincount(r.reshape(-1).long(), minlength=8) for r in eng.record_routing[:a.layers]]).float()
eng.record_routing = None
# ---- the policy that meets the north-star tolerance (round 6): precision="f16_operands" on the sparse-MoE engine -- fp32 residual stream,
#      routing decided in fp32 on the stream itself, expert GEMMs on fp16 copies of w1|w3 / w2 (93 GB more of HBM), fp32 combine --
#      timed like the default: the same batch, the same number of steps, HIP events at the step boundaries on the launch stream
f16_rate = None
if not a.no_f16:
    eng.set_precision("f16_operands")
    for _ in range(max(a.warmup, 1)):                   # (the first call converts the weights)
        eng.encode_pooled(ids, mask, "mean", True)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True)]
    marks[0].record()
    for _ in range(a.steps):
        e16 = eng.encode_pooled(ids, mask, "mean", True)
        marks.append(torch.cuda.Event(enable_timing=True)); marks[-1].record()
    torch.cuda.synchronize()
    ms16 = marks[0].elapsed_time(marks[-1]) / a.steps
    f16_rate = {"precision": "f16_operands", "docs_per_s": a.docs / ms16 * 1e3, "ms_per_step": ms16, "steps": a.steps,
                "overflow_flag": bool(ops.f16_overflow_flag(dev)), "finite": bool(torch.isfinite(e16).all()),
                "weights_subnormal_in_fp16_frac": eng.f16_weight_stats["subnormal"] / max(eng.f16_weight_stats["total"], 1)}
    eng.set_precision("bf16")
    eng._ws.clear()
    torch.cuda.empty_cache()
# ---- PARITY at the leg's own shape (VERDICT r04 #2c, r05 #1c), on the ENGINE'S OWN weights widened layer by layer (the 46.7 B-parameter
#      model is 187 GB in fp32: it never fits next to the engine, one layer's 5.6 GB does), against the reference's bidirectional Mixtral
#      restated in FP32 (oracle/torch_reference.py::mixtral_encode_fp32, pinned on the reference-generated Mixtral fixtures), for BOTH
#      policies of the engine.
#      (1) TEACHER-FORCED, per layer: the fp32 run's residual stream entering layers 0, L/2 and L-1 is handed to the engine
#      (`inputs_embeds`, `layer_range`), and the stream leaving the layer is compared token by token: routing agreement, per-row relative
#      error on tokens that took the same experts.  bf16 (the reference's arithmetic type): agreement >= 0.97, every token whose router
#      logits separate the 2nd from the 3rd choice by more than 4.0 agrees (the logits of this random-init router have a standard
#      deviation of ~30 and bf16 noise in x moves them by ~0.5), row error median < 2e-2.  f16_operands: agreement >= 0.995 (the routing
#      itself is exact fp32 arithmetic on the stream; what reaches it is the fp16-operand error of the attention block in front of it:
#      1.2e-3 of the stream at layer 0, where the stream IS that block's output, 2.5e-4 deeper -- times a logit scale of 30), every token
#      with a logit gap above 0.2 agrees, row error median < 2e-3.
#      (2) END TO END, all layers free-running: 1 - cos of the pooled embeddings, REPORTED for both policies with the north-star's 1e-4 as a
#      flag (not a bound: a token that re-routes once re-decides every later layer, so the free-running figure measures how many tokens
#      re-routed, not the arithmetic -- 16-bit operands of any kind leave some).  For bf16 it stands beside a same-run YARDSTICK -- the reference's data flow run in bf16 by plain torch on the same weights and documents
#      (mixtral_encode_fp32(dtype=bfloat16): what the reference's own bf16 run does to this model): top-2 routing is a discontinuous function
#      of x, a token that takes another expert once carries a perturbed state and re-decides its routing in every later layer, so ANY bf16
#      run and the fp32 run of a random-init 32-layer MoE decorrelate token by token.
parity = None
if a.parity_docs > 0:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import torch_reference as TR
    from gritlm_amd import _lib
    n = min(a.parity_docs, a.docs)
    blk = _lib.load().grit_swiglu_block()
    pols = ("bf16",) if a.no_f16 else ("bf16", "f16_operands")
    e_eng = {}
    for pol in pols:
        eng.set_precision(pol)
        e_eng[pol] = eng.encode_pooled(ids[:n].contiguous(), mask[:n].contiguous(), "mean", True).double()
    torch.cuda.synchronize()
    eng._ws.clear()
    torch.cuda.empty_cache()
    t0p = time.perf_counter()
    tl = tuple(sorted({0, a.layers // 2, a.layers - 1}))
    wl = lambda li: TR.mixtral_layer_weights_from_engine(eng, li, blk)
    GAP = {"bf16": 4.0, "f16_operands": 0.2}
    refs, yard = [], []
    per_layer = {pol: {li: {"agree": [], "clear_agree": [], "clear_n": 0, "row_rel": [], "upd_rel": []} for li in tl} for pol in pols}
    margs = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.rms_norm_eps, cfg.rope_theta)
    with torch.no_grad():
        for i in range(n):
            trace = {}
            refs.append(TR.mixtral_encode_fp32(wl, a.layers, eng.embed, eng.norm, ids[i:i + 1], mask[i:i + 1], *margs, trace_layers=tl, trace=trace))
            if i < a.yardstick_docs:
                yard.append(TR.mixtral_encode_fp32(wl, a.layers, eng.embed, eng.norm, ids[i:i + 1], mask[i:i + 1], *margs, dtype=torch.bfloat16))
            for pol in pols:
                eng.set_precision(pol)
                for li in tl:
                    x_in, x_out, sel, margin = trace[li]
                    eng.record_routing = []
                    got = eng.forward(None, mask[i:i + 1], inputs_embeds=x_in, layer_range=(li, li + 1), final_norm=False).float()
                    r_eng = eng.record_routing[0].view(1, a.seq, 2).sort(-1)[0].to(sel.device)
                    eng.record_routing = None
                    same = (r_eng == sel).all(-1)[0]
                    clear = margin[0] > GAP[pol]
                    d = per_layer[pol][li]
                    d["agree"].append(float(same.float().mean())); d["clear_n"] += int(clear.sum())
                    d["clear_agree"].append(bool(same[clear].all()))
                    xo, xi, g = x_out[0][same], x_in[0][same], got[0][same]
                    d["row_rel"].append(((g - xo).norm(dim=1) / xo.norm(dim=1)).cpu())
                    d["upd_rel"].append(((g - xo).norm(dim=1) / (xo - xi).norm(dim=1)).cpu())
            del trace
    eng.set_precision("bf16")
    e_ref = torch.cat(refs).double()
    torch.cuda.synchronize()
    omc_of = lambda e: 1.0 - (e * e_ref[:e.shape[0]]).sum(1) / (e.norm(dim=1) * e_ref[:e.shape[0]].norm(dim=1))
    BOUNDS = {"bf16": {"routing_agree_min": 0.97, "logit_gap_all_agree_above": 4.0, "row_rel_median_max": 2.0e-2},
              "f16_operands": {"routing_agree_min": 0.995, "logit_gap_all_agree_above": 0.2, "row_rel_median_max": 2.0e-3}}
    by_pol, all_ok = {}, True
    for pol in pols:
        tf, ok = {}, True
        for li, d in per_layer[pol].items():
            rr, ur = torch.cat(d["row_rel"]), torch.cat(d["upd_rel"])
            tf[str(li)] = {"routing_agree": min(d["agree"]), "clear_gap_tokens": d["clear_n"], "clear_gap_all_agree": all(d["clear_agree"]),
                           "row_rel_median": float(rr.median()), "row_rel_p90": float(rr.quantile(0.9)), "row_rel_max": float(rr.max()),
                           "update_rel_median": float(ur.median())}
            ok = ok and tf[str(li)]["routing_agree"] >= BOUNDS[pol]["routing_agree_min"] and tf[str(li)]["clear_gap_all_agree"] \
                and tf[str(li)]["row_rel_median"] < BOUNDS[pol]["row_rel_median_max"]
        omc = omc_of(e_eng[pol])
        e2e = {"max_one_minus_cos": float(omc.max()), "mean_one_minus_cos": float(omc.mean())}
        if pol == "f16_operands":
            e2e["north_star_1e-4_met"] = bool(e2e["max_one_minus_cos"] < 1e-4)
        by_pol[pol] = {"teacher_forced_per_layer": tf, "end_to_end": e2e, "bounds": BOUNDS[pol], "within_bound": bool(ok and torch.isfinite(e_eng[pol]).all())}
        all_ok = all_ok and by_pol[pol]["within_bound"]
    yard_omc = omc_of(torch.cat(yard).double()) if yard else None
    parity = {"what": f"the first {n} documents of the timed batch ({a.seq} tokens each): HIP engine under each precision policy vs the reference's "
                      "bidirectional Mixtral restated in FP32 on the engine's own weights (oracle/torch_reference.py, pinned on the reference-generated "
                      "fixtures).  teacher_forced_per_layer: the fp32 run's residual stream entering the layer is given to the engine, the stream "
                      "leaving it compared per token (row_rel = |engine - fp32| / |fp32| on tokens that took the same experts; update_rel relates the "
                      "same difference to the layer's own update).  end_to_end: all layers free-running, 1 - cos of the pooled embeddings: bounded by "
                      "the north-star's 1e-4 for f16_operands; for bf16 reported beside the same-run yardstick (the reference's data flow in bf16 by "
                      "plain torch, same weights and documents): any bf16 run of a random-init 32-layer top-2 MoE decorrelates from the fp32 run "
                      "token by token once tokens re-route",
              "docs": n, "policies": by_pol,
              "reference_dataflow_in_bf16_end_to_end": None if yard_omc is None else
              {"docs": len(yard), "max_one_minus_cos": float(yard_omc.max()), "mean_one_minus_cos": float(yard_omc.mean()),
               "what": "mixtral_encode_fp32(dtype=bfloat16): the reference's own bf16 arithmetic on this model, vs its fp32 run"},
              "north_star_policy": "f16_operands" if "f16_operands" in by_pol else None,
              "north_star_met": bool(by_pol.get("f16_operands", {}).get("within_bound", False)
                                     and by_pol.get("f16_operands", {}).get("end_to_end", {}).get("north_star_1e-4_met", False)),
              "north_star_note": "teacher-forced per layer the f16_operands policy is within its bounds (within_bound); the free-running end-to-end "
                                 "1 - cos meets 1e-4 only if no token re-routes anywhere in the stack (north_star_1e-4_met)",
              # (kept for readers of the round-5 layout: the default policy's data)
              "teacher_forced_per_layer": by_pol["bf16"]["teacher_forced_per_layer"], "end_to_end": by_pol["bf16"]["end_to_end"],
              "within_bound": bool(all_ok), "fp32_reference_seconds": time.perf_counter() - t0p}
    del refs, e_ref
    torch.cuda.empty_cache()
docs_per_s = a.docs * a.steps / dt
flops_doc = eng.flops_per_token(a.seq) * a.seq
# generation from one cached document on the native sparse-MoE decoder (round 6): ms per token, median slope of three (32, 96)-token run
# pairs, against the roofline of the weights a token actually streams (attention + router + TWO experts per layer + lm_head)
native_decode = None
try:
    from gritlm_amd.decoder import MistralDecoder
    eng.set_precision("bf16")
    lm_head = (torch.randn((32000, cfg.hidden_size), device=dev) * 0.02).to(torch.bfloat16)
    dec = MistralDecoder(eng, lm_head)
    one = torch.randint(3, 32000, (1, a.seq), device=dev)
    _, kv1 = eng.forward(one, torch.ones_like(one), return_kv=True)
    q4 = torch.randint(3, 32000, (1, 4), device=dev)
    dec.generate(q4, 8, past_key_values=kv1)
    slopes = []
    for _ in range(3):
        ts = []
        for n_new in (32, 96):
            torch.cuda.synchronize(); t0d = time.perf_counter()
            dec.generate(q4, n_new, past_key_values=kv1)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0d)
        slopes.append((ts[1] - ts[0]) / 64 * 1e3)
    ms_tok = sorted(slopes)[1]
    wb = (sum(L.wqkv.numel() + L.wo.numel() + L.wgate.numel() + 2 * (L.w13[0].numel() + L.w2[0].numel()) for L in eng.layers) + lm_head.numel()) * 2
    native_decode = {"what": "greedy generation from one cached document, native sparse-MoE decode step (router on the device, two experts' "
                             "GEMVs per row, one HIP graph), bf16", "ms_per_token": ms_tok, "ms_per_token_runs": slopes,
                     "weight_gb_streamed_per_token": wb / 1e9, "hbm_roofline_ms_per_token": wb / 8e12 * 1e3,
                     "frac_of_weight_streaming_roofline": wb / 8e12 * 1e3 / ms_tok}
    del dec, kv1, lm_head
except Exception as ex:  # noqa: BLE001 -- the leg's encode numbers must survive a decode failure
    native_decode = {"error": repr(ex)[:300]}
print(json.dumps({
    "metric": f"encoded docs/sec @ seq{a.seq} (Mixtral-8x7B shape, sparse MoE top-2)", "value": docs_per_s, "unit": "docs/s", "n_gpus": 1,
    "steps": a.steps, "ms_per_step": dt / a.steps * 1e3, "dtype": "bf16", "data": "synthetic, random-init weights",
    "config": {"workload": f"Mixtral-8x7B shape, {a.layers}L, 8 experts top-2, batch {a.docs} x seq{a.seq}, mean pool + normalise"},
    "model_flops_utilisation": docs_per_s * flops_doc / 2.5e15,
    "roofline": {"bound": "mfma", "kernel": "gemm_bf16_nt_k (grouped launches: expert GEMMs)", "peak": 2500.0, "unit": "TFLOP/s",
                 "achieved": ks["gemm_bf16_nt_grouped"]["work"] / (ks["gemm_bf16_nt_grouped"]["total_ms"] * 1e-3) / 1e12,
                 "frac": ks["gemm_bf16_nt_grouped"]["work"] / (ks["gemm_bf16_nt_grouped"]["total_ms"] * 1e-3) / 1e12 / 2500.0,
                 "whole_step_frac": docs_per_s * flops_doc / 2.5e15},
    "tokens_per_s": docs_per_s * a.seq, "weights_init_s": t_init, "hbm_allocated_gb": torch.cuda.max_memory_allocated() / 1e9,
    "north_star_policy": f16_rate, "parity": parity, "native_decode": native_decode,
    "expert_load_max_over_mean": float((counts.max(dim=1)[0] / counts.mean(dim=1)).mean()), "finite": bool(torch.isfinite(e).all()),
    "kernels": {k: {"launches": v["launches"], "total_ms": round(v["total_ms"], 2), "tflops": v["work"] / (v["total_ms"] * 1e-3) / 1e12}
                for k, v in ks.items()}}))
