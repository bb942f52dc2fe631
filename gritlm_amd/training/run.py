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
import random
import time

import torch
import torch.distributed as dist
from transformers import AutoTokenizer, HfArgumentParser, get_scheduler, set_seed

from .arguments import CustomTrainingArguments, DataArguments, ModelArguments
from .data import EmbeddingCollator, EmbeddingDataset, GenerativeCollator, ItemPicker, deal_to_rank, load_datasets, multi_dataset_order, pick_items
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

    # flags of the reference CLI that this entry point does not implement must fail loudly, not train something else
    unsupported = {"--lora": args.lora, "--qlora": args.qlora, "--split_emb_full": args.split_emb_full,
                   "--deepspeed": bool(getattr(args, "deepspeed", None)), "--fsdp": bool(getattr(args, "fsdp", None))}
    bad = [k for k, v in unsupported.items() if v]
    if bad:
        raise NotImplementedError(f"{', '.join(bad)}: accepted by the reference's gritlm.training.run, not implemented by the MI355X-native "
                                  f"entry point (full-parameter data-parallel training only; --shard_optimizer shards the AdamW state "
                                  f"over the ranks where the reference's configs use FSDP)")

    # GradCache switch, run.py:93-104: accumulation steps become chunks of one large contrastive batch
    gc_chunk = None
    if (args.gradient_accumulation_steps > 1 and args.negatives_cross_device and args.mode in ("embedding", "unified")) \
            or (args.no_gen_gas and args.no_emb_gas):
        gc_chunk = args.per_device_train_batch_size
        args.per_device_train_batch_size *= args.gradient_accumulation_steps
        args.gradient_accumulation_steps = 1
        logger.info("Using GradCache with chunk size %d", gc_chunk)
    elif args.no_gen_gas or args.no_emb_gas:
        raise ValueError("Cannot use no_gen_gas or no_emb_gas without GradCache")           # run.py:105-106
    gas = max(int(args.gradient_accumulation_steps), 1)       # plain accumulation (HF Trainer semantics) when GradCache is not engaged
    if data_args.generative_max_len is None:
        data_args.generative_max_len = data_args.passage_max_len                             # run.py:132-133

    tok = AutoTokenizer.from_pretrained(model_args.tokenizer_name or model_args.model_name_or_path, padding_side="right")
    if not tok.pad_token and tok.bos_token:
        tok.pad_token = tok.bos_token          # training pads with BOS (run.py:118-120), inference with EOS

    do_emb, do_gen = args.mode in ("embedding", "unified"), args.mode in ("unified", "generative")
    # one data set per file (run.py:122-204): subsample / too-long-instruction filter / --num_samples cap per file, rows kept per file
    # -> dataset_num_samples.json; several embedding files -> global batches drawn from ONE file each (CustomRandomSampler)
    num_samples = None
    if data_args.num_samples:
        with open(data_args.num_samples) as f:
            num_samples = json.load(f)
    emb_sets, gen_sets, kept = load_datasets(data_args.train_data, args.mode, tok, data_args.query_max_len, data_args.passage_max_len,
                                             data_args.generative_max_len, data_args.max_example_num_per_dataset, num_samples)
    rows = [r for _, rs in emb_sets for r in rs] if do_emb else []
    gen_rows = [t for _, ts in gen_sets for t in ts] if do_gen else []
    ds_lens = [len(rs) for _, rs in emb_sets] if do_emb else []
    if do_emb and not rows:
        raise ValueError(f"--mode {args.mode} needs rows with a 'query' field in {data_args.train_data} (after the instruction-length filter)")
    if do_gen and not gen_rows:
        raise ValueError(f"--mode {args.mode} needs rows with a 'text' field in {data_args.train_data}")
    if len(ds_lens) > 1 and not args.dataloader_drop_last:       # run.py:334
        raise AssertionError("Multiple datasets are only supported with dropping the last incomplete batch, set `--dataloader_drop_last`")
    os.makedirs(args.output_dir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.output_dir, "dataset_num_samples.json"), "w") as f:
            json.dump(kept, f)
    max_len = max(data_args.query_max_len or 0, data_args.passage_max_len or 0, data_args.generative_max_len or 0)
    ds = EmbeddingDataset(rows, data_args.train_group_size, max_char_len=max_len * 10, seed=args.seed + rank) if do_emb else None
    collate = EmbeddingCollator(tok, data_args.query_max_len, data_args.passage_max_len)
    collate_gen = GenerativeCollator(tok, data_args.generative_max_len, data_args.prefixlm) if do_gen else None
    gen_bs = args.per_device_generative_bs          # smaller generative batch: every (bs // gen_bs)-th sample (data.py:49-54,137-144)

    dtype = torch.bfloat16 if args.bf16 else torch.float32
    model = GritLMTrainModel(model_name_or_path=model_args.model_name_or_path, normalized=model_args.normalized,
                             pooling_method=model_args.pooling_method, negatives_cross_device=args.negatives_cross_device and world > 1,
                             temperature=args.temperature, loss_gen_type=args.loss_gen_type, loss_gen_factor=args.loss_gen_factor, mode=args.mode, projection=model_args.projection, attn=model_args.attn,
                             attn_implementation=model_args.attn_implementation, torch_dtype=dtype, device=device)
    model.model.to(device)
    model.model.train()
    if use_cuda and dtype == torch.bfloat16 and getattr(model.model.config, "model_type", "") in ("mistral", "mixtral") and model_args.attn[:2] in ("bb", "cc"):
        model.enable_native(device).cache_transposed_weights = True      # invalidated by weights_updated() after every step
        logger.info("native MI355X engine bound to %s", model_args.model_name_or_path)
    if args.gradient_checkpointing:
        model.gradient_checkpointing_enable()       # native engine: per-layer recompute; Hugging Face path: the module's own
    if model.projection is not None:
        model.projection.to(device)
    seen, params = set(), []
    for p_ in model.parameters():                   # backbone AND the optional --projection head (the reference optimises model.parameters())
        if p_.requires_grad and id(p_) not in seen:
            seen.add(id(p_)); params.append(p_)
    sharded = bool(args.shard_optimizer) and world > 1
    if sharded:        # ZeRO-1: AdamW's moments live on ONE rank per parameter; the owner updates and broadcasts (sharded_optim.py)
        from .sharded_optim import ShardedAdamW
        opt = ShardedAdamW(params, lr=args.learning_rate, weight_decay=args.weight_decay, betas=(args.adam_beta1, args.adam_beta2),
                           eps=args.adam_epsilon)
        logger.info("optimizer state sharded over %d ranks: this rank owns %.1f M of %.1f M parameter elements", world,
                    opt.owned_elements() / 1e6, sum(p_.numel() for p_ in params) / 1e6)
    else:
        opt = torch.optim.AdamW(params, lr=args.learning_rate, weight_decay=args.weight_decay, betas=(args.adam_beta1, args.adam_beta2),
                                eps=args.adam_epsilon)
    bs = args.per_device_train_batch_size
    if gen_bs is not None:
        assert bs >= gen_bs and bs % gen_bs == 0, "Full batch size must be divisible by the generative batch size"
    n_items = max(len(rows), len(gen_rows))         # unified: the longer data set drives the epoch (data.py:33)
    micro_per_epoch = max(n_items // (bs * world), 1)
    steps_per_epoch = max(micro_per_epoch // gas, 1)                 # one optimizer step per `gas` micro-batches
    total = args.max_steps if args.max_steps > 0 else int(steps_per_epoch * args.num_train_epochs)
    sched = get_scheduler(args.lr_scheduler_type, opt.local if sharded else opt, num_warmup_steps=args.get_warmup_steps(total),
                          num_training_steps=total)
    gc = GradCacheStep(model, gc_chunk, precision=getattr(args, "pass1_precision", None)) if gc_chunk else None

    def save_checkpoint(step_):
        """checkpoint-<step>/: weights under the reference parameter names + optimizer / scheduler / step (HF Trainer layout)."""
        ck = os.path.join(args.output_dir, f"checkpoint-{step_}")
        if rank == 0:
            model.model.save_pretrained(ck, safe_serialization=args.save_safetensors)
            if model.projection is not None:
                torch.save(model.projection.state_dict(), os.path.join(ck, "projection.pt"))
            if not sharded:
                torch.save(opt.state_dict(), os.path.join(ck, "optimizer.pt"))
            torch.save(sched.state_dict(), os.path.join(ck, "scheduler.pt"))
            with open(os.path.join(ck, "trainer_state.json"), "w") as f:
                json.dump({"global_step": step_, "micro_batches_done": step_ * gas, "world_size": world}, f)
        if dist.is_initialized():
            dist.barrier()                  # the directory exists on every rank's view before the per-rank files go in
        # every rank's OWN generators (HF Trainer: rng_state_{rank}.pth; one rank: rng_state.pth): the CPU generator and the generator of
        # the device this rank computes on.  (Round 3 wrote rank 0's states only and restored them on every rank.)
        os.makedirs(ck, exist_ok=True)
        cuda_state = torch.cuda.get_rng_state(torch.device(device)) if torch.device(device).type == "cuda" else None
        torch.save({"cpu": torch.get_rng_state(), "cuda": cuda_state},
                   os.path.join(ck, "rng_state.pth" if world == 1 else f"rng_state_{rank}.pth"))
        if sharded:                         # one optimizer file per rank: the moments of the parameters it owns
            torch.save(opt.state_dict(), os.path.join(ck, f"optimizer_shard_{rank}.pt"))
        if dist.is_initialized():
            dist.barrier()

    step, micro_done = 0, 0
    if args.resume_from_checkpoint:
        ck = args.resume_from_checkpoint
        if not isinstance(ck, str) or not os.path.isdir(ck):
            raise ValueError(f"--resume_from_checkpoint expects a checkpoint-<step> directory written by this entry point, got {ck!r}")
        st_path = os.path.join(ck, "trainer_state.json")
        st = json.load(open(st_path)) if os.path.exists(st_path) else {}
        lacking = [k_ for k_ in ("global_step", "micro_batches_done", "world_size") if k_ not in st]
        if lacking:       # e.g. a checkpoint written by the Hugging Face Trainer / the reference: its trainer_state.json has another schema
            raise ValueError(f"{st_path} lacks {lacking}: --resume_from_checkpoint resumes checkpoints written by THIS entry point "
                             "(weights of any Hugging Face checkpoint can be loaded with --model_name_or_path instead)")
        if st["world_size"] != world:
            raise ValueError(f"checkpoint was written with world_size {st['world_size']}, this run has {world} (data order differs)")
        from safetensors.torch import load_file
        wfile = os.path.join(ck, "model.safetensors")
        sd = load_file(wfile) if os.path.exists(wfile) else torch.load(os.path.join(ck, "pytorch_model.bin"), map_location="cpu")
        with torch.no_grad():                       # in place: the native engine's packed views stay bound to the parameters
            own = dict(model.model.named_parameters())
            own.update(dict(model.model.named_buffers()))
            unknown = [k_ for k_ in sd if k_ not in own]
            if unknown:
                logger.warning("checkpoint keys without a counterpart in the model (skipped): %s", unknown[:8])
            missing = [k_ for k_ in dict(model.model.named_parameters()) if k_ not in sd]
            if missing:
                raise ValueError(f"checkpoint {ck} lacks parameters of this model: {missing[:8]}")
            for k_, v_ in sd.items():
                if k_ in own:
                    own[k_].copy_(v_.to(own[k_].dtype))
        if model.projection is not None:
            model.projection.load_state_dict(torch.load(os.path.join(ck, "projection.pt"), map_location=device))
        if sharded:
            shard_path = os.path.join(ck, f"optimizer_shard_{rank}.pt")
            if not os.path.exists(shard_path):
                raise ValueError(f"{ck} holds no optimizer shard for rank {rank}: it was written without --shard_optimizer or by another world size")
            opt.load_state_dict(torch.load(shard_path, map_location=device))
        else:
            if not os.path.exists(os.path.join(ck, "optimizer.pt")):
                raise ValueError(f"{ck} holds no optimizer.pt (written with --shard_optimizer?): resume with the same flag and world size")
            opt.load_state_dict(torch.load(os.path.join(ck, "optimizer.pt"), map_location=device))
        sched.load_state_dict(torch.load(os.path.join(ck, "scheduler.pt")))
        step, micro_done = int(st["global_step"]), int(st["micro_batches_done"])
        rng_path = os.path.join(ck, "rng_state.pth" if world == 1 else f"rng_state_{rank}.pth")
        if os.path.exists(rng_path):
            rng = torch.load(rng_path, map_location="cpu")
            torch.set_rng_state(rng["cpu"])
            cs = rng.get("cuda")
            if torch.is_tensor(cs) and torch.device(device).type == "cuda":
                torch.cuda.set_rng_state(cs, torch.device(device))
            elif isinstance(cs, (list, tuple)) and cs and torch.cuda.is_available() and len(cs) == torch.cuda.device_count():
                torch.cuda.set_rng_state_all(cs)             # a round-3 checkpoint (rank 0's states of every visible device)
        if model.train_engine is not None:
            model.train_engine.weights_updated()
        logger.info("resumed from %s at step %d", ck, step)

    gen = torch.Generator().manual_seed(args.seed)
    t0, step0 = time.time(), step
    micro, skip = 0, micro_done                                      # resume: replay the permutations, skip the consumed micro-batches
    loss = loss_gen = None
    # which row answers a data-set index (data.py:92-97, :132-137): in range -> itself, past the end -> a random row, and with
    # --use_unique_indices the smaller data set of a unified run hands out this rank's share of its indices from a refilled set
    pick_rng = random.Random(args.seed * 1000003 + rank)
    unique = bool(data_args.use_unique_indices) and args.mode == "unified" and len(rows) != len(gen_rows)
    emb_pick = ItemPicker(len(rows), pick_rng, unique and len(rows) < len(gen_rows), rank, world) if do_emb else None
    gen_pick = ItemPicker(len(gen_rows), pick_rng, unique and len(gen_rows) < len(rows), rank, world) if do_gen else None
    while step < total:
        if len(ds_lens) > 1:
            # several embedding data sets: every global batch (bs x gas x world consecutive indices) from one of them where possible; the
            # per-device batches of a global batch are dealt to the ranks in turn (HF Trainer / accelerate: batch j goes to rank j % world)
            g_order = multi_dataset_order(ds_lens, bs * gas * world, gen)
            order = deal_to_rank(g_order, bs, rank, world)
        else:
            order = torch.randperm(n_items, generator=gen).tolist()
            order = order[rank::world]                              # disjoint shards per rank
        for s in range(0, len(order) - bs + 1, bs):
            idx = order[s:s + bs]
            nth = 1 if gen_bs is None else bs // gen_bs             # a smaller generative batch: every nth sample (data.py:49-54, :131)
            picks = [pick_items(emb_pick, gen_pick, i, want_gen=(k % nth == 0)) for k, i in enumerate(idx)]
            eidx, gidx = [e for e, _ in picks], [g for _, g in picks if g is not None]
            if skip > 0:                                            # resume: advance the sampling RNGs exactly as the consumed batches did
                skip -= 1
                if do_emb:
                    for j in eidx:
                        ds[j]
                continue
            micro += 1
            last_micro = micro % gas == 0                           # gradients are averaged over ranks / clipped / applied on this one
            scale = 1.0 / gas
            loss_gen = None
            if do_gen:
                # generative first (gradcache_trainer.py:551-579): it has no collective, the embedding step does
                gb = collate_gen([gen_rows[j] for j in gidx])
                gb = {k: v.to(device) for k, v in gb.items()}
                if args.no_gen_gas or gc_chunk is None:
                    loss_gen = model(generative=gb).loss_gen
                    (loss_gen * scale).backward()                   # 1/gas goes into the gradient only: the LOGGED loss is un-scaled, like
                    loss_gen = loss_gen.detach()                    # the embedding loss beside it
                else:
                    chunks = split_inputs(gb, gc_chunk)
                    loss_gen = torch.zeros((), device=device)
                    for ch in chunks:
                        lg = model(generative=ch).loss_gen / len(chunks)
                        lg.backward()
                        loss_gen += lg.detach()
                if not do_emb:
                    loss = loss_gen
                    if last_micro:
                        sync_gradients(model)
            if do_emb:
                batch = collate([ds[j] for j in eidx])
                q = {k: v.to(device) for k, v in batch["query"].items()}
                p = {k: v.to(device) for k, v in batch["passage"].items()}
            if not do_emb:
                pass
            elif gc is not None:
                loss = gc(q, p)
            elif args.split_emb:
                # two half-steps (gradcache_trainer.py:584-605): queries with grad vs frozen passages, then the converse;
                # both see the same scores, so the two losses agree
                lq = model(query=q, passage=p, p_grad=False).loss * scale
                lq.backward()
                lp = model(query=q, passage=p, q_grad=False).loss * scale
                lp.backward()
                assert torch.allclose(lq.detach(), lp.detach(), rtol=1e-3, atol=1e-4), (float(lq), float(lp))
                loss = lp.detach() / scale
                if last_micro:
                    sync_gradients(model)
            else:
                loss = model(query=q, passage=p, q_grad=not args.emb_p_only, p_grad=not args.emb_q_only).loss * scale
                loss.backward()
                loss = loss.detach() / scale
                if last_micro:
                    sync_gradients(model)
            if not last_micro:
                continue                                            # keep accumulating into .grad (HF Trainer: loss / gas per micro-batch)
            if args.max_grad_norm and args.max_grad_norm > 0:
                torch.nn.utils.clip_grad_norm_(params, args.max_grad_norm)
            opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
            if model.train_engine is not None:
                model.train_engine.weights_updated()
            step += 1
            if rank == 0 and step % max(args.logging_steps, 1) == 0:
                logger.info("step %d/%d loss %.4f%s lr %.3e %.2f s/it", step, total, float(loss),
                            "" if loss_gen is None else " loss_gen %.4f" % float(loss_gen), sched.get_last_lr()[0],
                            (time.time() - t0) / max(step - step0, 1))
            if str(getattr(args.save_strategy, "value", args.save_strategy)) == "steps" and args.save_steps and args.save_steps >= 1 and step % int(args.save_steps) == 0 \
                    and step < total:
                save_checkpoint(step)
            if step >= total:
                break
        if len(order) < bs:
            raise ValueError(f"dataset shard ({len(order)} rows) smaller than the per-device batch {bs}")
    if rank == 0:
        model.model.save_pretrained(args.output_dir, safe_serialization=args.save_safetensors)
        if model.projection is not None:
            torch.save(model.projection.state_dict(), os.path.join(args.output_dir, "projection.pt"))
        tok.save_pretrained(args.output_dir)
    if dist.is_initialized():
        dist.barrier()
    main.last_loss_gen = None if loss_gen is None else float(loss_gen)
    if loss is None:
        raise ValueError("no optimizer step was taken (max_steps / data too small, or the checkpoint already reached max_steps)")
    return float(loss.detach()) if torch.is_tensor(loss) else float(loss)


if __name__ == "__main__":
    main()
