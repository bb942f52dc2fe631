"""``python -m gritlm.training.run <flags>`` -> gritlm_amd.training.run (same flags as the reference entry point)."""
from gritlm_amd.training.run import main

if __name__ == "__main__":
    main()
