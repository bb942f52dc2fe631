from gritlm_amd.training.arguments import CustomTrainingArguments, DataArguments, ModelArguments  # noqa: F401
