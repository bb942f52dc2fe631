"""Run named checks of tests/gpu_checks.py outside pytest and print their full detail strings, one item per line.

    python tools/run_gpu_check.py check_train_step_7b_layer check_train_recompute
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import gpu_checks as G  # noqa: E402

bad = 0
for name in sys.argv[1:]:
    r = getattr(G, name)()
    bad += not r["ok"]
    print(f"{name}: {'ok' if r['ok'] else 'FAILED'}")
    print("  " + r["detail"].replace(" ", "\n  "))
sys.exit(1 if bad else 0)
