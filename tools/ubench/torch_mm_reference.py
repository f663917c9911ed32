#!/usr/bin/env python3
"""Reference point, not product code: the library GEMM (torch.mm -> hipBLASLt) on the prefill shapes, timed in short bursts
and sustained (>= 150 ms back to back) next to lmi_gemm with a plain store epilogue on the same operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leopard_amd.ops import Ops  # noqa: E402

dev = "cuda:0"
ops = Ops()


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench(M, N, K, dtype):
    a = torch.randn(M, K, device=dev).to(dtype)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dtype)
    out = torch.empty(M, N, device=dev, dtype=dtype)
    lib = lambda: torch.mm(a, w.t(), out=out)
    mine = lambda: ops.gemm(a, w, out)
    row = f"{M}x{N}x{K} {str(dtype)[6:]:9s}"
    for name, fn in (("lib", lib), ("lmi", mine)):
        timed(fn, 3)
        short = timed(fn, 5)
        n_long = max(10, int(200.0 / short))
        long_ = timed(fn, n_long)
        tf = lambda ms: 2 * M * N * K / ms / 1e9
        row += f" | {name}: burst {tf(short):5.0f}  sustained({n_long}) {tf(long_):5.0f} TF/s"
    print(row, flush=True)


for dt in (torch.float16, torch.bfloat16):
    for shp in ((7187, 28672, 4096), (7187, 4096, 14336), (7187, 6144, 4096), (7187, 4096, 4096), (28392, 3456, 1152),
                (28392, 4352, 1152), (28392, 1152, 4352), (28392, 1152, 1152), (8192, 8192, 8192)):
        bench(*shp, dt)
