#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: kernel_resources.py [file.hip] [extra -D flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else os.path.join(ROOT, "melonix_amd", "csrc", "stft_kernels.hip")
extra = [a for a in sys.argv[1:] if a.startswith("-")]
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off", "-c", "-x", "hip",
                      src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
if out.returncode:
    print(out.stderr[-3000:]); sys.exit(1)
blocks = re.split(r"Function Name: ", out.stderr)[1:]
for b in blocks:
    name = b.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"mx::|void |\(mx::StftArgs\)", "", short)
    sc, oc, ld = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>3} scratch {sc:>4} occ {oc} LDS {ld:>6}  {short}")
