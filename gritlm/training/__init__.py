"""Drop-in alias of gritlm_amd.training (``python -m gritlm.training.run``)."""
