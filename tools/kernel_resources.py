#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel of the HIP library, from the compiler's
resource remarks (hipcc -Rpass-analysis=kernel-resource-usage).  The walk shares each SIMD
with the side stream's kernels, so what each of them allocates decides how many fit."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "particle-simulator_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
       "-x", "hip", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/psim_resources.so",
       "psim_capi.cpp"]
text = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("S", r"TotalSGPRs: (\d+)"), ("V", r" VGPRs: (\d+)"), ("scr", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
try:
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except FileNotFoundError:
    names = [r["name"] for r in rows]
for r, nm in sorted(zip(rows, names), key=lambda t: -t[0].get("V", 0)):
    nm = re.sub(r"void psim::be::|psim::be::|psim_kernel<psim::|\(.*", "", nm)
    print("%-58s sgpr=%3d vgpr=%3d scratch=%d occ=%d lds=%d" % (nm[:58], r.get("S", 0), r.get("V", 0), r.get("scr", 0),
                                                              r.get("occ", 0), r.get("lds", 0)))
