from .gradcache import GradCacheStep, split_inputs, sync_gradients  # noqa: F401
from .model import DistributedContrastiveLoss, GritLMTrainModel, GritLMTrainOutput, NextTokenLoss  # noqa: F401
