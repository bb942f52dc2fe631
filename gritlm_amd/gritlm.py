"""``GritLM`` -- drop-in for the reference inference wrapper (gritlm/gritlm.py:9-218) with the
embedding hot path (bidirectional Mistral forward, pooling, L2-normalise) on the native MI355X engine.

Call-compatible surface: constructor arguments, ``encode`` / ``encode_queries`` /
``encode_corpus`` / ``pooling`` signatures, attributes ``model, tokenizer, device, generate, projection,
pooling_method, normalized, attn, embed_eos, num_gpus, embedding_attr``, error behaviour
(``ValueError`` for mixed attention strings, ``NotImplementedError`` for unknown pooling methods).

What runs where:
  * CUDA(HIP) device + Mistral / Mixtral backbone in bf16 + 'bb..' (bidirectional) or 'cc..' (causal) embedding attention: tokenise
    on the host, then ``MistralEncoderEngine`` (HIP kernels through the C ABI) + fused pool/normalise kernel, under the precision
    policy of ``precision=`` (``get_cache=True`` included: the engine emits the per-layer K/V).  A missing ``libgritlm_hip.so``
    raises -- there is no silent fallback on this path.  (With a ``projection`` head the engine computes the hidden states; the
    Linear and the pooling after it are the reference's torch ops.)
  * anything else (CPU plumbing config "SGPT-125M weightedmean", non-Mistral backbones, other dtypes): the Hugging Face module
    computes the hidden states exactly as in the reference; pooling still uses the HIP kernel when the hidden states are bf16 on
    the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Union

import numpy as np
import torch
from tqdm import tqdm

from ._lib import POOL_MODES

_VALID_ATTN = ("bbcc", "cccc", "bb", "cc")


def _torch_pool(hidden: torch.Tensor, mask: torch.Tensor, method: str) -> torch.Tensor:
    """Plain-torch pooling for tensors the HIP kernel does not take (CPU / fp32 hidden states).
    Semantics of gritlm/gritlm.py:188-214, including the in-place mask update of 'weightedmean'."""
    rows = torch.arange(hidden.shape[0], device=hidden.device)
    if method == "cls":
        return hidden[:, 0]
    if method == "lasttoken":
        n = mask.shape[1]
        last = (n - 1 - torch.argmax(mask.flip(dims=(1,)), dim=1)).clamp_min(0)
        return hidden[rows, last] * mask[rows, last].unsqueeze(-1).float()
    if method in ("mean", "weightedmean"):
        if method == "weightedmean":
            mask.mul_(mask.cumsum(dim=1))
        w = mask.unsqueeze(-1).float()
        return (hidden * w).sum(dim=1) / mask.sum(dim=1, keepdim=True).float()
    raise NotImplementedError(f"Unknown pooling method: {method}")


class GritLM(torch.nn.Module):
    def __init__(
        self,
        model_name_or_path: str = None,
        mode: str = "unified",            # 'unified' | 'embedding' | 'generative'
        pooling_method: str = "mean",     # 'cls' | 'lasttoken' | 'mean' | 'weightedmean'
        normalized: bool = True,
        projection: int = None,
        is_inference: bool = True,
        embed_eos: str = "",
        attn: str = "bbcc",
        device: str = "cuda" if torch.cuda.is_available() else "cpu",
        **kwargs,                          # forwarded to from_pretrained (torch_dtype, attn_implementation, ...)
    ) -> None:
        super().__init__()
        from transformers import AutoModel, AutoModelForCausalLM, AutoTokenizer

        native = kwargs.pop("native", "auto")   # "auto" | True | False  (extension over the reference)
        # the attention path decides how many keys a causal query sees under config.sliding_window (encoder.sliding_window_keys): taken
        # from the caller's explicit `attn_implementation=` (forwarded to from_pretrained as in the reference) rather than from the
        # PRIVATE config._attn_implementation, whose default differs between transformers releases ('eager' on 4.36-4.x, None on 5.x)
        self._attn_impl = kwargs.get("attn_implementation")
        devices = kwargs.pop("devices", None)   # extension: the GPUs in-process multi-GPU encode uses (default: every visible one)
        # extension: precision policy of the native engine (gritlm_amd.encoder.PRECISIONS): "bf16" = the reference's bf16 arithmetic
        # (default), "fp32_residual", "f16_operands" (fp32 stream + fp16 MFMA operands: 1 - cos < 1e-4 against the reference's fp32 run at
        # depth 32), "f16_stream" (the same with the residual stream in fp16: faster, range-limited).  `residual_fp32=True` (round 4) is kept as an alias of precision="fp32_residual".
        residual_fp32 = bool(kwargs.pop("residual_fp32", False))
        precision = kwargs.pop("precision", None) or ("fp32_residual" if residual_fp32 else "bf16")
        # "auto" (round 6): start on the fastest policy that meets the north-star tolerance ("f16_stream") and step down the ladder
        # f16_stream -> f16_operands -> fp32_residual -> bf16 the first time a kernel flags a value beyond the fp16 range (checkpoints with
        # massive activations): the affected call is re-run on the next rung, logged once; encode() never raises for range.
        from .encoder import PRECISIONS
        if precision not in PRECISIONS + ("auto",):
            raise ValueError(f"precision={precision!r}: one of {PRECISIONS + ('auto',)}")
        if mode == "embedding":
            if any(tag in model_name_or_path for tag in ("gtr", "t5", "instructor")):
                from transformers import T5EncoderModel
                self.model = T5EncoderModel.from_pretrained(model_name_or_path, **kwargs)
            else:
                self.model = AutoModel.from_pretrained(model_name_or_path, trust_remote_code=True, **kwargs)
            self.embedding_attr = None
        else:
            self.model = AutoModelForCausalLM.from_pretrained(model_name_or_path, trust_remote_code=True, **kwargs)
            self.generate = self.model.generate
            if hasattr(self.model, "model"):            # Llama / Mistral
                self.embedding_attr = "model"
            elif hasattr(self.model, "transformer"):    # GPT-Neo / GPT-J
                self.embedding_attr = "transformer"
            else:
                raise ValueError("Could not find attribute to use for embedding: ", self.model)

        self.projection = None
        if projection is not None:
            self.projection = torch.nn.Linear(self.model.config.hidden_size, int(projection), dtype=self.model.dtype)
        self.normalized = normalized
        self.pooling_method = pooling_method
        self.device = device
        self.num_gpus = 1
        self.engines = []            # in-process multi-GPU encode: one engine replica per GPU (set by _parallelize)
        self._precision = precision
        # encode(get_cache=True): False = bf16 K/V (the reference's cache format, for a Hugging Face generate()); True = the K/V of the
        # engine's policy as its attention read them -- fp16 under the fp16 policies -- for native_decoder().generate(past_key_values=...)
        self.native_kv_cache = False
        self.embed_eos = embed_eos
        self.attn = attn
        if (attn is not None) and attn not in _VALID_ATTN:
            raise ValueError(f"Mixed attention no longer supported: {self.attn}. Only bbcc, cccc, bb, cc are supported")
        self.engine = None
        self._native = native

        print(f"Created GritLM: {self.model.dtype} dtype, {pooling_method} pool, {mode} mode, {attn} attn")

        if is_inference:
            # right padding: instruction masking indexes from the left (reference :60-61)
            self.tokenizer = AutoTokenizer.from_pretrained(model_name_or_path, padding_side="right", trust_remote_code=True)
            if not self.tokenizer.pad_token and self.tokenizer.eos_token:
                self.tokenizer.pad_token = self.tokenizer.eos_token
                print("Set pad token to eos token: " + self.tokenizer.pad_token)
            if self.embed_eos:
                assert self.embed_eos in self.tokenizer.vocab, f"EOS token {self.embed_eos} not in vocab"
            self.model.eval()
            if "device_map" not in kwargs and not kwargs.get("load_in_4bit", False) and not kwargs.get("load_in_8bit", False):
                self.model.to(self.device)
            self._maybe_build_engine()
            if "device_map" not in kwargs and not kwargs.get("load_in_4bit", False) and not kwargs.get("load_in_8bit", False):
                # Parallelise embedding models unless a specific device is named, e.g. `cuda:1` (reference :69-75)
                if mode == "embedding" and (not isinstance(self.device, str) or ":" not in self.device):
                    self._parallelize(devices)

    # ------------------------------------------------------------------ native engine
    def _backbone(self):
        return getattr(self.model, self.embedding_attr) if self.embedding_attr else self.model

    def _maybe_build_engine(self):
        """Bind the HIP encoder to the backbone weights when the configuration is one it implements."""
        if self._native is False:
            return
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        cfg = self.model.config
        eligible = (dev.type == "cuda" and getattr(cfg, "model_type", "") in ("mistral", "mixtral") and self.attn is not None
                    and self.attn[:2] in ("bb", "cc") and self.model.dtype == torch.bfloat16)
        if not eligible:
            if dev.type == "cuda" and getattr(cfg, "model_type", "") in ("mistral", "mixtral"):
                # a Mistral on the GPU that silently takes the Hugging Face path is a 1.4x slower encode nobody asked for: say why
                print(f"GritLM: native HIP engine NOT bound (attn={self.attn}, dtype={self.model.dtype}; it implements bf16 weights with "
                      f"'bb..' / 'cc..' embedding attention) -- encode() runs on the Hugging Face module")
            if self._native is True:
                raise RuntimeError("native=True but this configuration is not implemented by the HIP engine "
                                   f"(device={dev}, model_type={getattr(cfg, 'model_type', None)}, attn={self.attn}, dtype={self.model.dtype})")
            return
        from .encoder import EncoderConfig, MistralEncoderEngine   # raises if libgritlm_hip.so is missing
        from . import _lib
        _lib.load()
        ecfg = EncoderConfig.from_hf(cfg)
        ecfg.check_supported()
        self.engine = MistralEncoderEngine.from_state_dict(ecfg, self._backbone().state_dict(), dev)
        self.engine.causal = self.attn[:2] == "cc"       # 'cc..': causal embedding attention (e.g. lasttoken / weightedmean models)
        from .encoder import sliding_window_keys
        impl = getattr(self, "_attn_impl", None) or getattr(cfg, "_attn_implementation", None) or "sdpa"      # the reference's default: sdpa
        self.engine.window_keys = sliding_window_keys(getattr(cfg, "sliding_window", None), impl)
        if self.engine.causal and getattr(cfg, "sliding_window", None):
            print(f"GritLM: causal embedding attention with config.sliding_window={cfg.sliding_window} on the '{impl}' path of the reference: "
                  + (f"a query sees {self.engine.window_keys} keys" if self.engine.window_keys else "no window (the sdpa path applies none)")
                  + "; pass attn_implementation= to choose, or set engine.window_keys")
        self.set_precision(getattr(self, "_precision", "bf16"))

    def set_precision(self, precision: str):
        """Precision policy of the native engine AND of every in-process replica (gritlm_amd.encoder.PRECISIONS, or "auto": the ladder)."""
        from .encoder import AUTO_LADDER, PRECISIONS
        if precision not in PRECISIONS + ("auto",):
            raise ValueError(f"precision={precision!r}: one of {PRECISIONS + ('auto',)}")
        self._precision = precision
        if self.engine is None:
            return self
        self._auto = precision == "auto"
        if self._auto:
            sup = self.engine.supported_precisions()
            self._ladder = [r for r in AUTO_LADDER if r in sup]
            precision = self._ladder[0]
        for eng in (getattr(self, "engines", None) or [self.engine]):
            eng.set_precision(precision)
        return self

    @property
    def precision(self) -> str:
        """The policy the engine(s) currently run (under "auto": the rung the ladder stands on)."""
        return self.engine.precision if self.engine is not None else self._precision

    def _f16_flags(self) -> list:
        """Read AND clear the fp16 overflow flag of every engine's device; returns the devices whose flag was set."""
        if self.engine is None or getattr(self.engine, "precision", None) not in ("f16_operands", "f16_stream"):
            return []
        return [str(e.device) for e in (getattr(self, "engines", None) or [self.engine]) if e.f16_overflowed(clear=True)]

    def _parallelize(self, devices=None):
        """The reference wraps an embedding model in ``nn.DataParallel`` over every visible GPU and multiplies ``batch_size`` by their
        number (gritlm/gritlm.py:69-75, :106-107): ONE process, ONE ``GritLM``, all GPUs (evaluation/eval_mteb.py constructs it that way).
        Here: one ENGINE REPLICA per GPU, all driven from this process -- every (batch_size x num_gpus) batch is tokenised once, its rows
        are dealt to the replicas in DataParallel's order (contiguous chunks of ceil(B / n) rows), each replica's launches are issued
        asynchronously on its own device (host-side batch geometry: no synchronisation between issue and result), and the pooled rows come
        back to the first device for the one concatenation.  No collective, no ``nn.DataParallel`` module replication per call.
        Models the engine does not implement keep the reference's behaviour (``nn.DataParallel`` around the Hugging Face module)."""
        if self.engine is None:
            n = torch.cuda.device_count() if torch.cuda.is_available() and torch.device(self.device).type == "cuda" else 0
            if n > 1:
                self.num_gpus = n
                print(f"----------Using {self.num_gpus} data-parallel GPUs----------")
                self.model = torch.nn.DataParallel(self.model)
            return
        if devices is None:
            # Inside a one-process-per-GPU launch (torchrun / torch.distributed: every rank sees every GPU unless the launcher pins
            # CUDA_VISIBLE_DEVICES) each rank owns ONE device: replicating onto all visible GPUs would put world_size copies of the model
            # on every GPU.  There the engine stays on its own device; pass `devices=[...]` to ask for in-process replicas explicitly.
            import os

            def _n(k):
                try:
                    return int(os.environ.get(k, "1") or 1)
                except ValueError:
                    return 1
            # a one-process-per-GPU launch = an initialised process group, torchrun's LOCAL_RANK, or a launcher that started MORE THAN ONE
            # task (WORLD_SIZE / SLURM_NTASKS / OMPI_COMM_WORLD_SIZE > 1).  SLURM_LOCALID / RANK alone say nothing: SLURM sets SLURM_LOCALID=0
            # in every step, including the reference's own single-task 8-GPU evaluation job (scripts/eval_mteb.sh: ntasks-per-node=1,
            # gres=gpu:8, plain `python`), which must replicate over all 8 GPUs as the reference's nn.DataParallel does
            in_dist = (torch.distributed.is_available() and torch.distributed.is_initialized()) or os.environ.get("LOCAL_RANK") is not None \
                or max(_n("WORLD_SIZE"), _n("SLURM_NTASKS"), _n("OMPI_COMM_WORLD_SIZE")) > 1
            if in_dist:
                if torch.cuda.device_count() > 1:
                    print("GritLM: one-process-per-GPU launch detected (process group / LOCAL_RANK / more than one task): the engine stays "
                          f"on {self.engine.device}; pass devices=[...] for in-process replicas")
                return
            devices = [f"cuda:{i}" for i in range(torch.cuda.device_count())]
        cur = torch.cuda.current_device()
        norm, seen = [], set()
        for d in devices:                        # "cuda" (no index) means the current device; a device named twice is used once
            d = torch.device(d)
            if d.type != "cuda":
                raise ValueError(f"devices={devices}: in-process multi-GPU encode runs on CUDA devices")
            idx = cur if d.index is None else d.index
            if idx not in seen:
                seen.add(idx)
                norm.append(torch.device("cuda", idx))
        devices = norm
        if len(devices) <= 1:
            return
        first = self.engine.device.index if self.engine.device.index is not None else cur
        self.engines = [self.engine if (d.index == first) else self.engine.replica(d) for d in devices]
        self.num_gpus = len(self.engines)
        print(f"----------Using {self.num_gpus} data-parallel GPUs (one native engine replica each)----------")

    @staticmethod
    def _row_chunks(n_rows: int, n_parts: int):
        """Row ranges ``torch.chunk`` / DataParallel's scatter give: contiguous chunks of ceil(n_rows / n_parts) rows, the last one shorter,
        trailing parts dropped when the rows run out."""
        size = -(-n_rows // n_parts) if n_rows else 0
        return [(s, min(s + size, n_rows)) for s in range(0, n_rows, size)] if size else []

    def _encode_native(self, inputs, n_instr, normalize=None):
        """One tokenised batch (HOST tensors) through the native engine(s): pooled (and, by default, normalised) rows [B, H] fp32 on the first
        engine's device."""
        normalize = bool(self.normalized) if normalize is None else bool(normalize)
        engines = getattr(self, "engines", None) or [self.engine]
        ids, mask = inputs["input_ids"], inputs["attention_mask"]
        parts = []
        for eng, (a, b) in zip(engines, self._row_chunks(ids.shape[0], len(engines))):
            il = None if n_instr is None else torch.full((b - a,), n_instr, dtype=torch.int32, device=eng.device)
            parts.append(eng.encode_pooled(ids[a:b], mask[a:b], self.pooling_method, normalize, il))
        if len(parts) == 1:
            return parts[0]
        return torch.cat([p.to(engines[0].device, non_blocking=True) for p in parts], dim=0)

    def native_decoder(self):
        """Greedy decoder on the HIP kernels (gritlm_amd.decoder.MistralDecoder) sharing the engine's weights; use it where the reference
        calls ``model.generate(..., past_key_values=kv_cache)`` on the cache returned by ``encode(get_cache=True)`` (rag/eval.py:296-302)."""
        if self.engine is None or not hasattr(self.model, "lm_head"):
            raise RuntimeError("native_decoder: needs the native engine and a causal-LM checkpoint (mode 'unified' or 'generative')")
        if getattr(self, "_decoder", None) is None:
            from .decoder import MistralDecoder
            self._decoder = MistralDecoder(self.engine, self.model.lm_head.weight)
        # the decoder follows the engine's policy (fp16 operands under the fp16 policies); under "auto" a value beyond the fp16 range
        # repeats the generate() call in bf16 instead of raising, like encode() steps down its ladder
        self._decoder.on_overflow = "bf16" if self._precision == "auto" else "raise"
        return self._decoder

    @torch.no_grad()
    def generate_native(self, input_ids: torch.Tensor, attention_mask: torch.Tensor | None = None, past_key_values=None,
                        max_new_tokens: int = 16, min_new_tokens: int = 0, pad_token_id: int | None = None, eos_token_id: int | None = None,
                        do_sample: bool = False, use_cache: bool = True, **unsupported) -> torch.Tensor:
        """``model.generate(**inputs, past_key_values=kv_cache, ...)`` as rag/eval.py:277-302 calls it, on the native decoder: greedy
        continuation of ``input_ids`` [B, P] (on top of the cached K/V when ``past_key_values`` is given -- then ``attention_mask`` is the
        reference's [B, cache + P] mask, ones over the cache), returned like Hugging Face's ``generate``: the input ids followed by the new
        tokens, positions after a sequence's EOS filled with ``pad_token_id``.  ``min_new_tokens`` is honoured when it equals
        ``max_new_tokens`` (the reference's latency runs: EOS never stops a row) or is 0; sampling, beams and other generation options are
        not native -- ``GritLM.generate`` (the Hugging Face method) stays available for them."""
        if do_sample or unsupported:
            raise NotImplementedError(f"generate_native: greedy decoding only (got do_sample={do_sample}, {sorted(unsupported)}); use GritLM.generate")
        if min_new_tokens not in (0, max_new_tokens):
            raise NotImplementedError("generate_native: min_new_tokens must be 0 or equal to max_new_tokens")
        dec = self.native_decoder()
        ids = input_ids.to(self.engine.device)
        B, P = ids.shape
        mask = attention_mask
        if past_key_values is not None and mask is not None:
            if mask.shape[1] != P:                    # the reference's mask spans cache + inputs: its tail is the inputs' own mask
                mask = mask[:, mask.shape[1] - P:]
        eos = None if min_new_tokens == max_new_tokens else (eos_token_id if eos_token_id is not None else getattr(self.tokenizer, "eos_token_id", None))
        new = dec.generate(ids, int(max_new_tokens), attention_mask=mask, past_key_values=past_key_values, eos_token_id=eos)
        if eos is not None:
            pad = pad_token_id if pad_token_id is not None else eos
            after = ((new == eos).long().cumsum(dim=1) - (new == eos).long()) > 0          # strictly after the first EOS of the row
            new = torch.where(after, torch.full_like(new, pad), new)
            done = (new == eos).any(dim=1)
            if bool(done.all()):                      # Hugging Face stops when every row has finished: trim the common tail of padding
                keep = int(((new == eos).long().cumsum(dim=1) > 0).long().argmax(dim=1).max()) + 1
                new = new[:, :keep]
        return torch.cat([ids, new.to(ids.dtype)], dim=1)

    # ------------------------------------------------------------------ API
    def encode_queries(self, queries: Union[List[str], str], **kwargs) -> np.ndarray:
        """Queries of retrieval / reranking tasks."""
        return self.encode(queries, **kwargs)

    def encode_corpus(self, corpus: Union[List[str], str, List[Dict[str, str]]], **kwargs) -> np.ndarray:
        """Corpus of retrieval tasks; dict documents are flattened to 'title text'."""
        if isinstance(corpus, dict):
            corpus = [corpus]
        if isinstance(corpus, list) and isinstance(corpus[0], dict):
            corpus = [(doc["title"] + " " + doc["text"]) if "title" in doc else doc["text"] for doc in corpus]
        return self.encode(corpus, **kwargs)

    def _hidden_states(self, inputs, get_cache: bool):
        """(last_hidden_state, kv_cache|None) for one tokenised batch."""
        if self.engine is not None and not get_cache:
            return self.engine.forward(inputs["input_ids"], inputs["attention_mask"], borrow=True), None
        if self.engine is not None and get_cache:
            # native pass that also emits the per-layer post-RoPE K / V (doc caching for RAG), packaged as the cache type the
            # installed transformers hands back for use_cache=True
            from transformers import DynamicCache
            # (bf16 K/V -- the reference's cache format, what a Hugging Face generate() continues from -- unless ``native_kv_cache``: then
            # the K/V of the engine's policy as the attention read them, fp16 under the fp16 policies, for the native decoder)
            hidden, kv = self.engine.forward(inputs["input_ids"], inputs["attention_mask"], borrow=True, return_kv=True,
                                             kv_dtype=None if getattr(self, "native_kv_cache", False) else torch.bfloat16)
            cache = DynamicCache()
            for li, (k, v) in enumerate(kv):
                cache.update(k, v, li)
            return hidden, cache
        kw = dict(inputs)
        if (self.attn is not None) and (self.attn[:2] == "bb"):
            kw["is_causal"] = False
        if get_cache:
            kw["use_cache"] = True
        out = self._backbone()(**kw)
        return out[0], (out[1] if get_cache else None)

    @torch.no_grad()
    def encode(
        self,
        sentences: Union[List[str], str],
        batch_size: int = 256,
        max_length: int = 512,
        instruction: str = "",
        embed_instruction: bool = False,
        get_cache: bool = False,
        convert_to_tensor: bool = False,
        recast: bool = False,
        add_special_tokens: bool = True,
        **kwargs,
    ) -> np.ndarray:
        if self.engine is not None and getattr(self.engine, "precision", None) in ("f16_operands", "f16_stream") and not kwargs.get("_ladder_rerun"):
            self._f16_flags()            # sticky per-device flags of an earlier, unchecked forward must not be blamed on this call
        if self.num_gpus > 1:
            batch_size *= self.num_gpus
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]

        n_instr = None
        if instruction and (embed_instruction is False) and ("mean" in self.pooling_method):
            # token count of the instruction tokenised ALONE with the same special-token setting (:146-152)
            n_instr = len(self.tokenizer(instruction, padding=False, truncation=True, max_length=max_length,
                                         add_special_tokens=add_special_tokens)["input_ids"])

        chunks, kv_caches = [], []
        for start in tqdm(range(0, len(sentences), batch_size), desc="Batches", disable=len(sentences) < 256):
            texts = [instruction + s + self.embed_eos for s in sentences[start:start + batch_size]]
            inputs = self.tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=max_length,
                                    add_special_tokens=add_special_tokens)
            if self.engine is not None and not get_cache and self.projection is None and self.pooling_method in POOL_MODES:
                # native fast path: the tokenizer's HOST tensors go straight to the engine(s) -- padding dropped before the first kernel,
                # pool + normalise fused, no device synchronisation per batch (engine.encode_pooled)
                if recast or self.pooling_method == "cls":
                    # the reference's order (gritlm.py:154-158): the pooled rows go back to the model dtype FIRST ('cls' never leaves it,
                    # :188) and are normalised in that dtype
                    emb = self._encode_native(inputs, n_instr, normalize=False).to(self.model.dtype)
                    if self.normalized:
                        emb = torch.nn.functional.normalize(emb, dim=-1).to(emb.dtype)
                else:
                    emb = self._encode_native(inputs, n_instr)
                chunks.append(emb)
                continue
            inputs = inputs.to(self.device)
            hidden, cache = self._hidden_states(inputs, get_cache)
            if get_cache:
                assert len(kv_caches) == 0, "Can only get cache for one batch at a time"
                kv_caches = cache
            if self.projection:
                hidden = self.projection(hidden)
            if n_instr is not None:
                inputs["attention_mask"][:, :n_instr] = 0      # attended to, but not pooled
            emb = self._pool_normalize(hidden, inputs["attention_mask"], recast)
            chunks.append(emb)

        if convert_to_tensor:
            result = torch.cat(chunks, dim=0)             # no sentences: torch.cat([]) raises, as in the reference (:163)
        elif not chunks:
            raise ValueError("need at least one array to concatenate")      # what the reference's np.concatenate([]) raises (:163)
        else:
            # ONE device->host copy for the whole call (the reference syncs per batch, :164)
            result = torch.cat(chunks, dim=0).to(torch.float32).cpu().numpy()
        flagged = self._f16_flags()       # every engine's flag is read and cleared (one 4-byte D2H each, next to the copy above)
        if flagged:
            pol = self.engine.precision
            if getattr(self, "_auto", False):
                # the ladder: this call's embeddings are invalid under `pol`; step down for good and run the call again
                nxt = self._ladder[self._ladder.index(pol) + 1]
                print(f"GritLM: precision='auto': an activation exceeded the fp16 range under '{pol}' on {', '.join(flagged)} -- this model "
                      f"runs under '{nxt}' from now on (this call is re-run)")
                for eng in (getattr(self, "engines", None) or [self.engine]):
                    eng.set_precision(nxt)
                return self.encode(sentences[0] if single else sentences, batch_size=batch_size // max(self.num_gpus, 1), max_length=max_length,
                                   instruction=instruction, embed_instruction=embed_instruction, get_cache=get_cache,
                                   convert_to_tensor=convert_to_tensor, recast=recast, add_special_tokens=add_special_tokens,
                                   _ladder_rerun=True, **{k: v for k, v in kwargs.items() if k != "_ladder_rerun"})
            from ._lib import GritHipError
            raise GritHipError(f"precision='{pol}': an activation exceeded the fp16 range (|v| >= 65520) on {', '.join(flagged)} in this call; the "
                               "embeddings of this call are invalid -- use precision='auto' (steps down by itself), "
                               + ("'f16_operands' (fp32 residual stream), " if pol == "f16_stream" else "") + "'fp32_residual' or 'bf16'")
        if single:
            result = result[0]
        if get_cache:
            return result, kv_caches
        return result

    # ------------------------------------------------------------------ pooling
    def _native_poolable(self, hidden: torch.Tensor, mask: torch.Tensor) -> bool:
        return (hidden.is_cuda and hidden.dtype == torch.bfloat16 and hidden.dim() == 3 and hidden.shape[-1] % 8 == 0
                and mask is not None and self.pooling_method in POOL_MODES)

    def _pool_normalize(self, hidden, mask, recast):
        """pooling + (optional) F.normalize fused in one kernel when possible (reference :154-158)."""
        if self._native_poolable(hidden, mask):
            from . import ops
            m = mask.to(device=hidden.device, dtype=torch.int64).contiguous()
            emb = ops.pool_norm(hidden.contiguous(), m, self.pooling_method, bool(self.normalized))
            if self.pooling_method == "weightedmean":
                mask.mul_(mask.cumsum(dim=1))                 # keep the reference's side effect (:211)
            if recast or self.pooling_method == "cls":        # 'cls' never leaves the hidden dtype (:188)
                emb = emb.to(hidden.dtype)
            return emb
        emb = self.pooling(hidden, mask, recast=recast)
        if self.normalized:
            emb = torch.nn.functional.normalize(emb, dim=-1).to(emb.dtype)
        return emb

    def pooling(self, hidden_state: torch.Tensor, attention_mask: torch.Tensor = None, recast: bool = False) -> torch.Tensor:
        """hidden_state [b, n, d], attention_mask [b, n] -> [b, d] (fp32 unless ``recast`` / 'cls')."""
        hidden_state = hidden_state.to(attention_mask.device)
        if self.pooling_method not in POOL_MODES:
            raise NotImplementedError(f"Unknown pooling method: {self.pooling_method}")
        if self._native_poolable(hidden_state, attention_mask):
            from . import ops
            m = attention_mask.to(torch.int64).contiguous()
            emb = ops.pool_norm(hidden_state.contiguous(), m, self.pooling_method, False)
            if self.pooling_method == "weightedmean":
                attention_mask.mul_(attention_mask.cumsum(dim=1))
            if self.pooling_method == "cls":
                emb = emb.to(hidden_state.dtype)
        else:
            emb = _torch_pool(hidden_state, attention_mask, self.pooling_method)
        return emb.to(hidden_state.dtype) if recast else emb
