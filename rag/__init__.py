"""Import alias for the reference's top-level ``rag`` directory: ``from rag.index import DistributedIndex, load_passages,
load_or_initialize_index`` resolves to gritlm_amd.rag (the index search runs on the native kernel)."""
