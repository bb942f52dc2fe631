"""Random-argument fuzz of the C ABI WITHOUT a GPU (run it against the sanitizer build: tools/asan_host_shim.sh --fuzz): every grit_*
entry point gets null / misaligned / valid-looking host pointers and extreme sizes.  Nothing can be launched here (no device), so
the only things exercised are the host shim's validation and size arithmetic: a call must come back with an error code (or a
size), never crash, and the sanitizers must stay silent.     usage: fuzz_abi.py [seed] [calls per entry point]"""
import ctypes as C, random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gritlm_amd import _lib
lib = _lib.load()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
buf = (C.c_char * (1 << 16))()
addr = C.addressof(buf)
skip = {"grit_comm_init", "grit_comm_allgather_packed", "grit_comm_destroy", "grit_comm_unique_id", "grit_stream_create_cu_mask", "grit_stream_destroy"}
ints = [-1, 0, 1, 2, 7, 8, 32, 64, 128, 256, 4096, 14336, 2**31 - 1, -2**31]
longs = [-1, 0, 1, 64, 512, 4096, 131072, 2**40]
floats = [0.0, 1.0, -1.0, 1e-5, float("nan"), float("inf")]
total = 0; codes = {}
for name, (res, args) in _lib._SIGNATURES.items():
    if name in skip or not args: continue
    fn = getattr(lib, name)
    for _ in range(N):
        vals = []
        for a in args:
            if a is C.c_void_p: vals.append(random.choice([None, addr, addr + 16, addr + 1]))
            elif a is C.c_int: vals.append(random.choice(ints))
            elif a is C.c_int64: vals.append(random.choice(longs))
            elif a is C.c_float: vals.append(random.choice(floats))
            else: vals.append(None)
        r = fn(*vals)
        total += 1
        codes[int(r) if r is not None else None] = codes.get(int(r) if r is not None else None, 0) + 1
neg = {k: v for k, v in codes.items() if k is not None and k < 0}
print("calls", total, "negative return codes", dict(sorted(neg.items())), "zero", codes.get(0, 0), "positive", sum(v for k, v in codes.items() if k and k > 0))
