"""The C-ABI host shim under AddressSanitizer + UBSan (tools/asan_host_shim.sh): every rejected-argument call of tests/test_abi.py runs
against a host-instrumented build of the library (device code untouched).  No GPU needed."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"), reason="no shared ASAN runtime in this image")
def test_abi_tests_pass_on_the_sanitizer_build():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_shim.sh")], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail


@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"), reason="no shared ASAN runtime in this image")
def test_random_arguments_never_crash_the_shim():
    """tools/fuzz_abi.py on the sanitizer build: null / misaligned pointers and extreme sizes for every entry point come back as error
    codes (round 3 found a division by zero in a size query and signed overflows in front of the range checks this way)."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_shim.sh"), "--fuzz", "7", "300"], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "calls " in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    codes = r.stdout.split("negative return codes", 1)[1].split("}")[0]
    assert set(int(k.split(":")[0].strip(" {")) for k in codes.split(",") if ":" in k) <= {-1, -2, -3}, codes


@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so"), reason="no shared TSAN runtime in this image")
def test_concurrent_callers_are_race_free():
    """tools/tsan_host_shim.sh: 8 threads in the library at once on a ThreadSanitizer build of the host shim (per-thread error text, per-device
    LDS opt-ins, knob statics, the persistent GEMM's counter ring)."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan_host_shim.sh")], capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "unexpected results 0" in r.stdout and "ThreadSanitizer" not in tail, tail
