#!/usr/bin/env python3
"""Run every GPU parity check, never stop at the first failure, print one table (for a single gpurun call)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("GRIT_DUMP_DIR", os.path.join(ROOT, "gpurun_out", "dumps"))
import torch  # noqa: E402
import gpu_checks  # noqa: E402

def main():
    only = sys.argv[1:]
    nfail = 0
    for name, fn, kw in gpu_checks.ALL_CHECKS:
        if only and not any(o in name for o in only):
            continue
        t0 = time.time()
        try:
            r = fn(**kw)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            r = dict(name=name, ok=False, detail="EXC " + repr(e)[:300])
            traceback.print_exc()
        nfail += 0 if r["ok"] else 1
        print(f"{'PASS' if r['ok'] else 'FAIL'}  {r['name']:<60s} {r['detail']}  ({time.time() - t0:.2f}s)", flush=True)
    print(f"== {nfail} failing checks")
    sys.exit(1 if nfail else 0)


if __name__ == "__main__":
    main()
