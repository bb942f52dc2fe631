"""Run checks of tests/gpu_checks.py outside pytest and print their full detail strings, one item per line.  Arguments are registry
names of ALL_CHECKS (what `pytest -k` matches, e.g. native_generate_f16) or function names (check_train_recompute: default arguments).

    python tools/run_gpu_check.py gemv_f16 attn_decode_f16 check_train_recompute
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import gpu_checks as G  # noqa: E402

by_name = {n: (fn, kw) for n, fn, kw in G.ALL_CHECKS}
bad = 0
for name in sys.argv[1:]:
    fn, kw = by_name[name] if name in by_name else (getattr(G, name), {})
    try:
        r = fn(**kw)
    except Exception as e:  # noqa: BLE001 -- report and go on to the next check
        import traceback
        traceback.print_exc()
        r = dict(ok=False, detail=f"raised {type(e).__name__}: {e}")
    bad += not r["ok"]
    print(f"{name}: {'ok' if r['ok'] else 'FAILED'}")
    print("  " + r["detail"].replace(" ", "\n  "))
sys.exit(1 if bad else 0)
