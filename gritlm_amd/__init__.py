"""gritlm_amd -- MI355X-native (gfx950) engine for the GritLM embedding-encode / contrastive hot path.

``from gritlm_amd import GritLM`` is call-compatible with ``from gritlm import GritLM`` of the reference."""
__version__ = "0.1.0"


def __getattr__(name):          # lazy: importing the package must not import transformers
    if name == "GritLM":
        from .gritlm import GritLM
        return GritLM
    raise AttributeError(name)
