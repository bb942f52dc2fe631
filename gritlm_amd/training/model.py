"""Train-time model and contrastive loss: drop-in for gritlm/training/model.py (reference lines cited inline).

* ``DistributedContrastiveLoss`` -- same constructor / call contract (:25-64).  On a HIP device the gathered fp32
  representations go through ONE fused kernel call (similarity on the exact-f32 matrix pipe + cross-entropy +
  gradients for the rank's own rows); the cross-rank exchange is ONE packed all-gather (q||p) over RCCL instead of
  two list-API gathers + cats.  Only the local shard carries grad, exactly as after ``_dist_gather_tensor`` (:49-60).
* ``GritLMTrainModel`` -- ``encode`` / ``forward`` with the reference signature (:134-222); the embedding tower runs on
  ``MistralTrainEngine`` when the backbone is a bf16 Mistral on a HIP device with bidirectional attention.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor
from transformers.utils import ModelOutput

from ..gritlm import GritLM

logger = logging.getLogger(__name__)


@dataclass
class GritLMTrainOutput(ModelOutput):
    q_reps: Optional[Tensor] = None
    p_reps: Optional[Tensor] = None
    loss: Optional[Tensor] = None
    loss_emb: Optional[Tensor] = None
    loss_gen: Optional[Tensor] = None


def packed_all_gather(q: Tensor, p: Tensor, world_size: int):
    """One collective for both towers: every rank contributes [Bq + Bp, H]; returns (q_all [W*Bq,H], p_all [W*Bp,H])
    in rank order (= the ``torch.cat`` order of the reference, :57-58, which the targets ``arange(B) * G`` rely on)."""
    bq, bp = q.shape[0], p.shape[0]
    if q.is_cuda and q.dtype == torch.float32 and p.dtype == torch.float32:
        from .. import comm as _comm
        if _comm.enabled():       # RCCL through the C ABI: one grouped gather straight into the rank-major matrices, no packing copy
            return _comm.NativeComm.get(q.device).allgather_packed(q, p).wait()
    packed = torch.cat([q, p], dim=0).contiguous()
    flat = torch.empty((world_size * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(flat, packed)           # concatenated layout: works on RCCL and gloo alike
    out = flat.view((world_size,) + tuple(packed.shape))
    return out[:, :bq].reshape(world_size * bq, -1).contiguous(), out[:, bq:].reshape(world_size * bp, -1).contiguous()


class _InfoNCEFn(torch.autograd.Function):
    """loss(q_local, p_local) with the gathered negatives as constants; backward = the cached row gradients."""

    @staticmethod
    def forward(ctx, q_local, p_local, q_all, p_all, temperature, q_off, p_off):
        from .. import ops
        loss, dq, dp = ops.infonce(q_all, p_all, temperature, q_off, q_local.shape[0], p_off, p_local.shape[0], want_grad=True)
        ctx.save_for_backward(dq, dp)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        dq, dp = ctx.saved_tensors
        return g * dq, g * dp, None, None, None, None, None


class DistributedContrastiveLoss:
    def __init__(self, temperature: float, negatives_cross_device: bool):
        self.cross_entropy = torch.nn.CrossEntropyLoss(reduction="mean")
        self.temperature = temperature
        self.negatives_cross_device = negatives_cross_device
        if self.negatives_cross_device:
            if not dist.is_initialized():
                raise ValueError("Cannot do negatives_cross_device without distributed training")
            self.rank = dist.get_rank()
            self.world_size = dist.get_world_size()

    def __call__(self, q_reps: Tensor, p_reps: Tensor) -> Tensor:
        native = q_reps.is_cuda and q_reps.dim() == 2 and p_reps.dim() == 2
        if native:
            q32, p32 = q_reps.float().contiguous(), p_reps.float().contiguous()
            q_off = p_off = 0
            q_all, p_all = q32, p32
            if self.negatives_cross_device:
                with torch.no_grad():
                    q_all, p_all = packed_all_gather(q32.detach(), p32.detach(), self.world_size)
                q_off, p_off = self.rank * q32.shape[0], self.rank * p32.shape[0]
            return _InfoNCEFn.apply(q32, p32, q_all.detach(), p_all.detach(), float(self.temperature), q_off, p_off)
        # host tensors (gloo / CPU plumbing tests): same arithmetic with torch ops
        if self.negatives_cross_device:
            q_reps = self._dist_gather_tensor(q_reps)
            p_reps = self._dist_gather_tensor(p_reps)
        scores = self.compute_similarity(q_reps, p_reps) / self.temperature
        scores = scores.view(q_reps.size(0), -1)
        target = torch.arange(scores.size(0), device=scores.device, dtype=torch.long) * (p_reps.size(0) // q_reps.size(0))
        return self.cross_entropy(scores, target)

    def with_gathered(self, q_local: Tensor, p_local: Tensor, q_all: Tensor, p_all: Tensor) -> Tensor:
        """Loss when the cross-rank exchange already happened (GradCacheStep overlaps it with pass 1):
        ``q_all`` / ``p_all`` are the rank-ordered gathers, the local shards are the only rows carrying grad."""
        bq, bp = q_local.shape[0], p_local.shape[0]
        if q_local.is_cuda:
            return _InfoNCEFn.apply(q_local.float().contiguous(), p_local.float().contiguous(), q_all.detach().float().contiguous(),
                                    p_all.detach().float().contiguous(), float(self.temperature), self.rank * bq, self.rank * bp)
        qs = [q_all[r * bq:(r + 1) * bq].detach() for r in range(self.world_size)]
        ps = [p_all[r * bp:(r + 1) * bp].detach() for r in range(self.world_size)]
        qs[self.rank], ps[self.rank] = q_local, p_local
        q, p = torch.cat(qs, dim=0), torch.cat(ps, dim=0)
        scores = (self.compute_similarity(q, p) / self.temperature).view(q.size(0), -1)
        target = torch.arange(scores.size(0), device=scores.device, dtype=torch.long) * (p.size(0) // q.size(0))
        return self.cross_entropy(scores, target)

    def _dist_gather_tensor(self, t: Optional[Tensor]):
        """all_gather whose local slot keeps the grad-carrying tensor (:49-60)."""
        if t is None:
            return None
        t = t.contiguous()
        flat = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(flat, t.detach())
        out = flat.view((self.world_size,) + tuple(t.shape))
        parts = [out[r] for r in range(self.world_size)]
        parts[self.rank] = t
        return torch.cat(parts, dim=0)

    def compute_similarity(self, q_reps, p_reps):
        return torch.matmul(q_reps, p_reps.transpose(-2, -1))


class NextTokenLoss:
    """Generative objective of unified training (:66-107), torch restatement (CPU / non-native path; the native path computes the
    same quantity in MistralTrainEngine.forward_lm with grit_ce_fwd / grit_ce_bwd)."""

    def __init__(self, vocab_size: int, loss_gen_type: str = "mixed", loss_gen_factor: float = 1.0):
        self.vocab_size, self.loss_gen_factor, self.loss_gen_type = vocab_size, loss_gen_factor, loss_gen_type
        if loss_gen_type == "token":
            self.cross_entropy = torch.nn.CrossEntropyLoss(reduction="sum")
        elif loss_gen_type == "mixed":
            self.cross_entropy = torch.nn.CrossEntropyLoss(reduction="mean")
        else:
            raise ValueError(f"Invalid loss_gen_type: {loss_gen_type}")

    def __call__(self, labels, logits):
        sl = logits[..., :-1, :].contiguous().view(-1, self.vocab_size)
        tl = labels[..., 1:].contiguous().view(-1).to(sl.device)
        loss = self.cross_entropy(sl, tl)
        if self.loss_gen_type == "token":
            loss = loss / labels.size(0)
        return loss * self.loss_gen_factor


class _NativeEncodeFn(torch.autograd.Function):
    """reps = normalize(pool(encoder(ids))) on the HIP engine; backward accumulates the parameter gradients
    directly into the packed ``.grad`` storage (like fused wgrad accumulation) and returns no tensor gradients."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, attention_mask, instr_len, want_grad):
        # `want_grad` is decided by the CALLER (torch.is_grad_enabled() there): inside a Function's forward grad mode is always off and
        # ctx.needs_input_grad only mirrors anchor.requires_grad -- it is True under torch.no_grad() too.  Round 3 found GradCache's
        # no-grad pass 1 saving every activation (and running the SAVE epilogue) because of that.
        eng = model.train_engine
        grad = bool(want_grad)
        reps, state = eng.forward_pooled(input_ids, attention_mask, model.pooling_method, bool(model.normalized), instr_len, save=grad,
                                         packed=getattr(model, "native_packed", True), causal=model.attn[:2] == "cc")
        if grad:
            ctx.model, ctx.state = model, state
        return reps

    @staticmethod
    def backward(ctx, d_reps):
        model = ctx.model
        model.train_engine.backward_pooled(ctx.state, d_reps, on_layer_done=getattr(model, "_on_layer_done", None))
        ctx.state = None
        return None, None, None, None, None, None


class _NativeGenFn(torch.autograd.Function):
    """loss_gen = NextTokenLoss(labels, lm_head(model(ids, causal))) on the HIP engine (causal attention, lm_head GEMM, fused cross
    entropy); backward accumulates backbone + lm_head gradients into ``.grad`` and returns no tensor gradients."""

    @staticmethod
    def forward(ctx, anchor, model, input_ids, attention_mask, labels, want_grad):
        eng = model.train_engine
        grad = bool(want_grad)                    # decided by the caller, see _NativeEncodeFn
        fn = model.gen_loss_fn
        if fn is not None:
            kind, factor, aux = fn.loss_gen_type, fn.loss_gen_factor, 0.0
        else:       # a Mixtral: the reference takes the model's own loss (training/model.py:123-127,185-194) = token-sum cross entropy
            #         / batch * loss_gen_factor + router_aux_loss_coef * load_balancing_loss (modeling_mixtral_gritlm.py:1406-1430)
            factor = model.gen_add_kwargs.get("loss_gen_factor")
            kind, factor, aux = "token", (1.0 if factor is None else factor), float(getattr(model.model.config, "router_aux_loss_coef", 0.0))
        loss, state = eng.forward_lm(input_ids, attention_mask, labels, kind, factor, save=grad,
                                     packed=getattr(model, "native_packed", True), router_aux_coef=aux)
        if grad:
            ctx.model, ctx.state = model, state
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        ctx.model.train_engine.backward_lm(ctx.state, d_loss, on_layer_done=getattr(ctx.model, "_on_layer_done", None))
        ctx.state = None
        return None, None, None, None, None, None


class GritLMTrainModel(GritLM):
    def __init__(
        self,
        temperature: float = 1.0,
        negatives_cross_device: bool = False,
        loss_gen_type: str = "mixed",
        loss_gen_factor: float = None,
        **kwargs,
    ):
        super().__init__(**kwargs, is_inference=False)
        self.emb_loss_fn = DistributedContrastiveLoss(temperature, negatives_cross_device)
        self.gen_add_kwargs = {"return_dict": True}
        if "mixtral" in kwargs["model_name_or_path"].lower():
            logger.info("Using token loss with routing loss for mixtral")
            self.gen_loss_fn = None
            self.gen_add_kwargs["loss_gen_factor"] = loss_gen_factor
            self.gen_add_kwargs["output_router_logits"] = True
        else:
            self.gen_loss_fn = NextTokenLoss(self.model.config.vocab_size, loss_gen_type, loss_gen_factor)
        self.config = self.model.config
        self.train_engine = None

    # ------------------------------------------------------------------ native engine
    def enable_native(self, device=None):
        """Bind the HIP training engine to the backbone (after the model sits on its device in bf16)."""
        from .engine import MistralTrainEngine, MixtralTrainEngine
        dev = torch.device(device if device is not None else self.device)
        cfg = self.model.config
        mtype = getattr(cfg, "model_type", "")
        # ('bb..': the bidirectional embedding attention of GritLM; 'cc..': causal embedding attention, e.g. lasttoken / weightedmean models
        #  -- round 6: the engine's causal attention forward / backward serve the embedding tower as they serve the generative branch)
        if not (dev.type == "cuda" and mtype in ("mistral", "mixtral") and self.attn[:2] in ("bb", "cc") and self.model.dtype == torch.bfloat16):
            raise RuntimeError(f"native training engine needs a bf16 Mistral / Mixtral on a HIP device with 'bb' or 'cc' embedding attention "
                               f"(got device={dev}, model_type={getattr(cfg, 'model_type', None)}, attn={self.attn}, dtype={self.model.dtype})")
        self.model.to(dev)
        # a causal-LM wrapper (mode unified / generative) also hands its lm_head to the engine: generative branch on HIP kernels
        engine_cls = MixtralTrainEngine if mtype == "mixtral" else MistralTrainEngine
        self.train_engine = engine_cls(self._backbone(), cfg, dev, lm_head=getattr(self.model, "lm_head", None))
        return self.train_engine

    def encode(self, features):
        if features is None:
            return None
        if self.train_engine is not None and self.projection is None:
            ids, mask = features["input_ids"], features["attention_mask"]
            il = features.get("instruction_lens")
            if il is not None:
                il = torch.as_tensor(il, dtype=torch.int32, device=self.train_engine.device).contiguous()
                if "mean" not in self.pooling_method:
                    raise NotImplementedError("instruction_lens with non-mean pooling is not on the native path")
            anchor = self.train_engine.embed
            return _NativeEncodeFn.apply(anchor, self, ids, mask, il, torch.is_grad_enabled() and anchor.requires_grad)
        # ---- Hugging Face path (CPU / other architectures): the reference's steps (:134-165)
        attention_mask = features["attention_mask"].clone() if "attention_mask" in features else None
        instruction_lens = features.get("instruction_lens")
        kwargs = {"input_ids": features.get("input_ids"), "attention_mask": attention_mask}
        if self.attn[:2] == "bb":
            kwargs["is_causal"] = False
        out = self._backbone()(**kwargs)[0]
        if self.projection is not None:
            out = self.projection(out)
        if instruction_lens is not None:
            attention_mask = features["attention_mask"].clone()
            for i, l in enumerate(instruction_lens):
                attention_mask[i, :l] = 0
                assert attention_mask[i].sum() > 0, f"All 0: {attention_mask[i]}, l: {l}"
        reps = self.pooling(out, attention_mask)
        if self.normalized:
            in_dtype = reps.dtype
            return torch.nn.functional.normalize(reps, dim=-1).contiguous().to(in_dtype)
        return reps.contiguous()

    def forward(
        self,
        query: Dict[str, torch.Tensor] = None,
        passage: Dict[str, torch.Tensor] = None,
        generative: Dict[str, torch.Tensor] = None,
        q_reps: Optional[torch.Tensor] = None,
        p_reps: Optional[torch.Tensor] = None,
        q_grad: bool = True,
        p_grad: bool = True,
    ):
        """query [b, n]; passage [b*s, m] (s = group size); generative [b, m]."""
        loss_gen = None
        if generative is not None:      # generative first, as in the reference (:185-194)
            native_gen = self.train_engine is not None and self.train_engine.lm_head is not None and self.attn[2:4] == "cc"
            if native_gen and (self.gen_loss_fn is not None or getattr(self.model.config, "model_type", "") == "mixtral"):
                labels = generative.pop("labels")
                anchor = self.train_engine.embed
                loss_gen = _NativeGenFn.apply(anchor, self, generative["input_ids"], generative["attention_mask"], labels,
                                              torch.is_grad_enabled() and anchor.requires_grad)
            elif self.gen_loss_fn is not None:
                loss_gen = self.gen_loss_fn(generative.pop("labels"), self.model(**generative, **self.gen_add_kwargs).logits)
            else:
                loss_gen = self.model(**generative, **self.gen_add_kwargs).loss

        def tower(feats, with_grad):
            if with_grad:
                return self.encode(feats)
            with torch.no_grad():
                return self.encode(feats)

        if (q_reps is None) and (query is not None):
            q_reps = tower(query, q_grad)
        if (p_reps is None) and (passage is not None):
            p_reps = tower(passage, p_grad)
        loss_emb = self.emb_loss_fn(q_reps, p_reps) if (q_reps is not None and p_reps is not None) else None
        loss = sum([x for x in [loss_emb, loss_gen] if x is not None])
        return GritLMTrainOutput(q_reps=q_reps, p_reps=p_reps, loss=loss, loss_emb=loss_emb, loss_gen=loss_gen)

    def gradient_checkpointing_enable(self, *args, **kwargs):
        if self.train_engine is None:
            self.model.gradient_checkpointing_enable(*args, **kwargs)
        else:   # native engine: keep only the layer inputs of a chunk, re-run each layer's forward inside backward
            self.train_engine.recompute = True
