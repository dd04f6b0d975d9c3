#!/usr/bin/env python3
"""Print VGPR / spill / scratch / occupancy / LDS per kernel of one .hip file
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: tools/kernel_resources.py influentialrs_amd/csrc/score.hip [--all]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
show_all = "--all" in sys.argv
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o",
                        os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
txt = r.stderr
blocks = re.split(r"remark: Function Name: ", txt)[1:]
def g(b, pat):
    m = re.search(pat, b)
    return int(m.group(1)) if m else -1
for b in blocks:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    v, a = g(b, r"VGPRs: (\d+)"), g(b, r"AGPRs: (\d+)")
    sp, sc = g(b, r"VGPRs Spill: (\d+)"), g(b, r"ScratchSize \[bytes/lane\]: (\d+)")
    occ, lds = g(b, r"Occupancy \[waves/SIMD\]: (\d+)"), g(b, r"LDS Size \[bytes/block\]: (\d+)")
    if show_all or sp > 0 or sc > 0:
        print(f"{name[:70]:70s} V={v:3d} A={a:3d} spill={sp:3d} scratch={sc:4d} occ={occ} lds={lds}")
print(len(blocks), "kernels in", src)
