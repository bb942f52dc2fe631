"""rag/index.py of the reference -> gritlm_amd.rag (same class / function names, same shard files)."""
from gritlm_amd.rag import DenseIndex, DistributedIndex, load_or_initialize_index, load_passages  # noqa: F401
