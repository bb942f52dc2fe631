"""bench.py's deadline guard (N > 1 only: protects the primary JSON line against a rank that stalls inside the contrastive leg's
collectives): the guard thread must print the late line and end the process with exit code 0 while the main thread is blocked;
a guard that is disarmed in time must stay silent."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROG = r"""
import sys, time, importlib.util
spec = importlib.util.spec_from_file_location("bench", sys.argv[1])
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
mode = sys.argv[2]
ev = bench.deadline_guard(0.5, lambda: '{"late": true}')
if mode == "stall":
    time.sleep(30)          # a main thread that never comes back (parked in a collective)
    print("NOT REACHED")
else:
    ev.set()
    time.sleep(1.0)
    print('{"late": false}')
"""


def _run(mode):
    return subprocess.run([sys.executable, "-c", _PROG, os.path.join(ROOT, "bench.py"), mode], capture_output=True, text=True, timeout=120)


def test_guard_emits_the_line_and_exits_cleanly_when_the_main_thread_stalls():
    r = _run("stall")
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.strip().splitlines()[-1] == '{"late": true}'
    assert "NOT REACHED" not in r.stdout


def test_disarmed_guard_stays_silent():
    r = _run("ok")
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.strip().splitlines() == ['{"late": false}']
