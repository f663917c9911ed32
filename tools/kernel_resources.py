#!/usr/bin/env python3
"""Register / LDS / spill figures of the gfx950 kernels inside libleopard_amd.so, from the code object's metadata notes.

    python tools/kernel_resources.py [pattern ...]      (substring filters on the demangled or mangled name; default: every GEMM kernel)

Extracts the .hip_fatbin section, unbundles the gfx950 code object (clang-offload-bundler) and reads `llvm-readelf --notes`."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pats = sys.argv[1:] or ["gemm"]
    so = os.environ.get("LEOPARD_AMD_LIB") or os.path.join(REPO, "leopard_amd", "libleopard_amd.so")
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, f"{d}/fat.bin"], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={d}/k.co"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{d}/k.co"], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        rows.append((re.search(r"\.name:\s+(\S+)", k).group(1), g("vgpr_count"), int(re.match(r"\s*(\d+)", k).group(1)), g("sgpr_count"),
                     g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spill':>6} {'scratch':>8} {'lds':>7}  kernel")
    for r, n in zip(rows, names):
        if any(p in n or p in r[0] for p in pats):
            print(f"{r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:6d} {r[5]:8d} {r[6]:7d}  {n[:230] if n != r[0] else r[0][:230]}")


if __name__ == "__main__":
    main()
