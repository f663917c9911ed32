#!/usr/bin/env python3
"""Which tile geometry suits each C3 linear when the operands are fp8 (the MFMA time halves, the epilogue / prologue do not):
lmi_gemm_fp8 under gemm.config in {0: 128x128 ring, 2: 256x128 3-slot ring, 5: staggered 256x256, 8: 64x128}.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402
from tools.bench_fp8 import DEV, F8, SHAPES, time_fn  # noqa: E402


def main():
    ops = Ops()
    g = torch.Generator(device=DEV).manual_seed(0)
    cfgs = (-1, 0, 2, 5, 8)
    print(f"{'shape':22s} | " + " | ".join(f"cfg {c:2d} ms (TF/s)" for c in cfgs))
    for name, M, N, K, epi, act in SHAPES:
        a8 = torch.randn(M, K, generator=g, device=DEV).clamp(-448, 448).to(F8).view(torch.uint8)
        w8 = (torch.randn(N, K, generator=g, device=DEV) * 0.25).to(F8).view(torch.uint8)
        if epi == _lib.EPI_SWIGLU:
            out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
        elif epi == _lib.EPI_RESIDUAL:
            out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        else:
            out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        cells = []
        for c in cfgs:
            ops.set_option("gemm.config", c)
            fn = lambda: ops.gemm_fp8(a8, w8, out, epilogue=epi, act=act, scale_exp=-6)  # noqa: E731
            for _ in range(3):
                fn()
            t = min(time_fn(fn, 20) for _ in range(3))
            cells.append(f"{t:6.3f} ({2.0 * M * N * K / t / 1e9:5.0f})")
        ops.set_option("gemm.config", -1)
        print(f"{name:22s} | " + " | ".join(f"{c:>16s}" for c in cells))


if __name__ == "__main__":
    main()
