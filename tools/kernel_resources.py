#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage of one .hip file as a table (registers, spills, scratch, LDS per kernel).
usage: python tools/kernel_resources.py gritlm_amd/csrc/gemm_bf16.hip [filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)(?: \[-Rpass|$)", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for c in rows:
    name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name:48s} VGPR {c.get('VGPRs','?'):>4s} AGPR {c.get('AGPRs','?'):>4s} SGPR {c.get('SGPRs','?'):>4s} spillV {c.get('VGPRs Spill','?'):>3s} "
          f"spillS {c.get('SGPRs Spill','?'):>3s} scratch {c.get('ScratchSize [bytes/lane]','?'):>4s} occ {c.get('Occupancy [waves/SIMD]','?'):>2s} LDS {c.get('LDS Size [bytes/block]','?')}")
