"""Drop-in alias: ``from gritlm import GritLM`` resolves to the MI355X-native implementation (gritlm_amd)."""
from gritlm_amd import __version__  # noqa: F401
from gritlm_amd.gritlm import GritLM  # noqa: F401
