#!/usr/bin/env python3
"""Run a chosen subset of tests/gpu_checks.py and print one RESULT line per check: python tools/dbg/run_checks.py name [name ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.chdir(ROOT)
import gpu_checks as G
want = set(sys.argv[1:])
if __name__ == "__main__":
    for name, fn, kw in G.ALL_CHECKS:
        if name in want:
            r = fn(**kw)
            print("RESULT", name, r["ok"], r["detail"], flush=True)
