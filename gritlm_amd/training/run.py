"""``python -m gritlm_amd.training.run <flags>`` -- entry point of contrastive (embedding-mode) training on the native engine.

Same flags as ``python -m gritlm.training.run`` (gritlm/training/run.py:54; arguments.py) and the same outputs in
``output_dir`` (``dataset_num_samples.json``, ``config.json``, weights under the reference parameter names, tokenizer).
The reference's ``main`` and its GradCacheTrainer (a copy of HF 4.36 Trainer internals) do not run on the installed
transformers, so the step loop is hosted here: GradCache switch (run.py:93-104), per-step
GradCacheStep / direct step, AdamW + linear schedule, data-parallel gradient averaging over RCCL.

    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m gritlm_amd.training.run --model_name_or_path DIR --train_data DATA \\
        --output_dir OUT --bf16 --per_device_train_batch_size 32 --gradient_accumulation_steps 8 --negatives_cross_device \\
        --train_group_size 8 --pooling_method mean --attn bbcc --query_max_len 256 --passage_max_len 2048 --max_steps 1253
"""
from __future__ import annotations

import json
import logging
import os
import time

import torch
import torch.distributed as dist
from transformers import AutoTokenizer, HfArgumentParser, get_scheduler, set_seed

from .arguments import CustomTrainingArguments, DataArguments, ModelArguments
from .data import EmbeddingCollator, EmbeddingDataset, GenerativeCollator, load_embedding_rows, load_generative_rows
from .gradcache import split_inputs
from .gradcache import GradCacheStep, sync_gradients
from .model import GritLMTrainModel

logger = logging.getLogger(__name__)


def main(argv=None):
    model_args, data_args, args = HfArgumentParser((ModelArguments, DataArguments, CustomTrainingArguments)).parse_args_into_dataclasses(argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", level=logging.INFO)
    if args.mode not in ("embedding", "unified", "generative"):
        raise NotImplementedError(args.mode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if world > 1 and not dist.is_initialized():
        if use_cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl" if use_cuda else "gloo")
    device = f"cuda:{local_rank}" if use_cuda else "cpu"
    set_seed(args.seed)

    # GradCache switch, run.py:93-104: accumulation steps become chunks of one large contrastive batch
    gc_chunk = None
    if (args.gradient_accumulation_steps > 1 and args.negatives_cross_device) or (args.no_gen_gas and args.no_emb_gas):
        gc_chunk = args.per_device_train_batch_size
        args.per_device_train_batch_size *= args.gradient_accumulation_steps
        args.gradient_accumulation_steps = 1
        logger.info("Using GradCache with chunk size %d", gc_chunk)

    tok = AutoTokenizer.from_pretrained(model_args.tokenizer_name or model_args.model_name_or_path, padding_side="right")
    if not tok.pad_token and tok.bos_token:
        tok.pad_token = tok.bos_token          # training pads with BOS (run.py:118-120), inference with EOS

    do_emb, do_gen = args.mode in ("embedding", "unified"), args.mode in ("unified", "generative")
    rows = load_embedding_rows(data_args.train_data, data_args.max_example_num_per_dataset) if do_emb else []
    gen_rows = load_generative_rows(data_args.train_data, data_args.max_example_num_per_dataset) if do_gen else []
    if do_gen and not gen_rows:
        raise ValueError(f"--mode {args.mode} needs rows with a 'text' field in {data_args.train_data}")
    if do_gen and isinstance(gen_rows[0], (tuple, list)):       # too long instructions leave nothing to learn from (run.py:166-176)
        gen_rows = [r for r in gen_rows if len(tok.tokenize("<|user|>\n" + r[0] + "\n<|assistant|>\n")) < data_args.generative_max_len]
    os.makedirs(args.output_dir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.output_dir, "dataset_num_samples.json"), "w") as f:
            json.dump({os.path.basename(data_args.train_data.rstrip("/")): len(rows) + len(gen_rows)}, f)
    max_len = max(data_args.query_max_len or 0, data_args.passage_max_len or 0, data_args.generative_max_len or 0)
    ds = EmbeddingDataset(rows, data_args.train_group_size, max_char_len=max_len * 10, seed=args.seed + rank) if do_emb else None
    collate = EmbeddingCollator(tok, data_args.query_max_len, data_args.passage_max_len)
    collate_gen = GenerativeCollator(tok, data_args.generative_max_len or 128, data_args.prefixlm) if do_gen else None
    gen_bs = args.per_device_generative_bs          # smaller generative batch: every (bs // gen_bs)-th sample (data.py:49-54,137-144)

    dtype = torch.bfloat16 if args.bf16 else torch.float32
    model = GritLMTrainModel(model_name_or_path=model_args.model_name_or_path, normalized=model_args.normalized,
                             pooling_method=model_args.pooling_method, negatives_cross_device=args.negatives_cross_device and world > 1,
                             temperature=args.temperature, loss_gen_type=args.loss_gen_type, loss_gen_factor=args.loss_gen_factor, mode=args.mode, projection=model_args.projection, attn=model_args.attn,
                             attn_implementation=model_args.attn_implementation, torch_dtype=dtype, device=device)
    model.model.to(device)
    model.model.train()
    if use_cuda and dtype == torch.bfloat16 and getattr(model.model.config, "model_type", "") == "mistral" and model_args.attn[:2] == "bb":
        model.enable_native(device).cache_transposed_weights = True      # invalidated by weights_updated() after every step
        logger.info("native MI355X engine bound to %s", model_args.model_name_or_path)
    params = [p for p in model.model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=args.learning_rate, weight_decay=args.weight_decay, betas=(args.adam_beta1, args.adam_beta2),
                            eps=args.adam_epsilon)
    bs = args.per_device_train_batch_size
    if gen_bs is not None:
        assert bs >= gen_bs and bs % gen_bs == 0, "Full batch size must be divisible by the generative batch size"
    n_items = max(len(rows), len(gen_rows))         # unified: the longer data set drives the epoch (data.py:33)
    steps_per_epoch = max(n_items // (bs * world), 1)
    total = args.max_steps if args.max_steps > 0 else int(steps_per_epoch * args.num_train_epochs)
    sched = get_scheduler(args.lr_scheduler_type, opt, num_warmup_steps=args.get_warmup_steps(total), num_training_steps=total)
    gc = GradCacheStep(model, gc_chunk) if gc_chunk else None

    gen = torch.Generator().manual_seed(args.seed)
    step, t0 = 0, time.time()
    while step < total:
        order = torch.randperm(n_items, generator=gen).tolist()
        order = order[rank::world]                                  # disjoint shards per rank
        for s in range(0, len(order) - bs + 1, bs):
            idx = order[s:s + bs]
            loss_gen = None
            if do_gen:
                # generative first (gradcache_trainer.py:551-579): it has no collective, the embedding step does
                take = idx if gen_bs is None else idx[::bs // gen_bs]
                gb = collate_gen([gen_rows[i % len(gen_rows)] for i in take])
                gb = {k: v.to(device) for k, v in gb.items()}
                if args.no_gen_gas or gc_chunk is None:
                    loss_gen = model(generative=gb).loss_gen
                    loss_gen.backward()
                    loss_gen = loss_gen.detach()
                else:
                    chunks = split_inputs(gb, gc_chunk)
                    loss_gen = torch.zeros((), device=device)
                    for ch in chunks:
                        lg = model(generative=ch).loss_gen / len(chunks)
                        lg.backward()
                        loss_gen += lg.detach()
                if not do_emb:
                    loss = loss_gen
                    sync_gradients(model)
            if do_emb:
                batch = collate([ds[i % len(ds)] for i in idx])
                q = {k: v.to(device) for k, v in batch["query"].items()}
                p = {k: v.to(device) for k, v in batch["passage"].items()}
            if not do_emb:
                pass
            elif gc is not None:
                loss = gc(q, p)
            elif args.split_emb:
                # two half-steps (gradcache_trainer.py:584-605): queries with grad vs frozen passages, then the converse;
                # both see the same scores, so the two losses agree
                lq = model(query=q, passage=p, p_grad=False).loss
                lq.backward()
                lp = model(query=q, passage=p, q_grad=False).loss
                lp.backward()
                assert torch.allclose(lq.detach(), lp.detach(), rtol=1e-3, atol=1e-4), (float(lq), float(lp))
                loss = lp
                sync_gradients(model)
            else:
                loss = model(query=q, passage=p, q_grad=not args.emb_p_only, p_grad=not args.emb_q_only).loss
                loss.backward()
                sync_gradients(model)
            if args.max_grad_norm and args.max_grad_norm > 0:
                torch.nn.utils.clip_grad_norm_(params, args.max_grad_norm)
            opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
            if model.train_engine is not None:
                model.train_engine.weights_updated()
            step += 1
            if rank == 0 and step % max(args.logging_steps, 1) == 0:
                logger.info("step %d/%d loss %.4f%s lr %.3e %.2f s/it", step, total, float(loss),
                            "" if loss_gen is None else " loss_gen %.4f" % float(loss_gen), sched.get_last_lr()[0], (time.time() - t0) / step)
            if step >= total:
                break
        if len(order) < bs:
            raise ValueError(f"dataset shard ({len(order)} rows) smaller than the per-device batch {bs}")
    if rank == 0:
        model.model.save_pretrained(args.output_dir, safe_serialization=args.save_safetensors)
        tok.save_pretrained(args.output_dir)
    if dist.is_initialized():
        dist.barrier()
    main.last_loss_gen = None if loss_gen is None else float(loss_gen)
    return float(loss.detach()) if torch.is_tensor(loss) else float(loss)


if __name__ == "__main__":
    main()
