from gritlm_amd.training.model import DistributedContrastiveLoss, GritLMTrainModel, GritLMTrainOutput, NextTokenLoss  # noqa: F401
