"""Command-line flags of ``python -m gritlm_amd.training.run``: the flag NAMES and defaults of the reference
(gritlm/training/arguments.py:9-154) for everything that reaches the embedding hot path, on top of HF TrainingArguments."""
from dataclasses import dataclass, field
from typing import Optional

from transformers import TrainingArguments


@dataclass
class ModelArguments:
    model_name_or_path: str = field(metadata={"help": "Path to a pretrained model directory"})
    config_name: Optional[str] = field(default=None)
    tokenizer_name: Optional[str] = field(default=None)
    pooling_method: str = field(default="weightedmean", metadata={"help": "cls | lasttoken | mean | weightedmean"})
    normalized: bool = field(default=True)
    attn_implementation: str = field(default="sdpa", metadata={"help": "only used by the Hugging Face (non-native) path"})
    attn: str = field(default="bbcc", metadata={"help": "bidirectional/causal attn for emb inst., emb sample, gen inst., gen sample"})
    projection: int = field(default=None)


@dataclass
class DataArguments:
    train_data: str = field(default=None, metadata={"help": "jsonl file or directory of jsonl files (query / pos / neg)"})
    train_group_size: int = field(default=2, metadata={"help": "passages per query: 1 positive + (n-1) negatives"})
    query_max_len: int = field(default=32)
    passage_max_len: int = field(default=128)
    generative_max_len: int = field(default=None)
    max_example_num_per_dataset: int = field(default=100_000_000)
    num_samples: Optional[str] = field(default=None)
    use_unique_indices: bool = field(default=False)
    prefixlm: bool = field(default=False)


@dataclass
class CustomTrainingArguments(TrainingArguments):
    negatives_cross_device: bool = field(default=False, metadata={"help": "share the negatives across all GPUs"})
    temperature: Optional[float] = field(default=0.02)
    mode: str = field(default="embedding", metadata={"help": "only 'embedding' runs on the native path"})
    per_device_generative_bs: int = field(default=None)
    no_gen_gas: bool = field(default=False)
    no_emb_gas: bool = field(default=False)
    loss_gen_factor: float = field(default=1.0)
    loss_gen_type: str = field(default="mixed")
    lora: bool = field(default=False)
    qlora: bool = field(default=False)
    save_safetensors: bool = field(default=False)
    split_emb: bool = field(default=False)
    split_emb_full: bool = field(default=False)
    emb_q_only: bool = field(default=False)
    emb_p_only: bool = field(default=False)
    pass1_precision: Optional[str] = field(default=None, metadata={"help": "(not a reference flag; native engine + GradCache only) precision "
                                           "policy of GradCache pass 1, the no-grad forward that defines the loss: bf16 | f16_operands | f16_stream "
                                           "(default: GRIT_PASS1_PRECISION or bf16, the reference's arithmetic)"})
    shard_optimizer: bool = field(default=False, metadata={"help": "(not a reference flag) AdamW state sharded over the data-parallel ranks: every "
                                  "parameter has one owner rank that updates and broadcasts it (training/sharded_optim.py); what the "
                                  "reference's FSDP configs buy for the 8x7B model, with whole bf16 replicas kept for the kernels"})
