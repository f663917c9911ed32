#!/usr/bin/env python3
"""Compile csrc/capi.hip with -Rpass-analysis=kernel-resource-usage and print VGPR/AGPR/spill/LDS/occupancy per kernel.
usage: python tools/kernel_resources.py [filter-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    filters = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                            "-Rpass-analysis=kernel-resource-usage", "-Wno-unused-value", "-o", os.path.join(td, "x.so"),
                            os.path.join(REPO, "leopard_amd", "csrc", "capi.hip")], capture_output=True, text=True)
    txt = r.stderr
    blocks = re.split(r"Function Name: ", txt)[1:]
    for b in blocks:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if filters and not all(f in dem for f in filters):
            continue
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return m.group(1) if m else "?"
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print("%-150s VGPR %3s AGPR %3s spill %2s scratch %3s occ %s LDS %s" % (dem[:150], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"),
                                                                                scratch, occ, lds))


if __name__ == "__main__":
    main()
