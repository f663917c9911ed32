#!/usr/bin/env python3
"""Why is lmi_gemm_skinny slower from the nn.Linear layout than from the packed copy?  Times the Llama decode projections at M = 8 from
(a) the packed copy, (b) row-major rows of stride K, (c) row-major rows padded to stride K + 64 / K + 192 elements (breaks power-of-two
row strides: a DRAM-channel / partition effect would show here), each at skinny.coalesce 1 and 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd.ops import Ops  # noqa: E402
from leopard_amd.weights import skinny_pack  # noqa: E402

ops = Ops()
dev = "cuda:0"
M = 8


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, N, K, epi in [("gate/up (SwiGLU)", 28672, 4096, 2), ("down", 4096, 14336, 1), ("q|k|v", 6144, 4096, 0), ("o_proj", 4096, 4096, 1)]:
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.float16).to(dev)
    x = torch.randn(M, K, generator=g).to(torch.float16).to(dev)
    out = torch.zeros(M, N // 2 if epi == 2 else N, dtype=torch.float32 if epi == 1 else torch.float16, device=dev)
    wp = skinny_pack(w)
    row = [f"{name:18s} N={N:6d} K={K:6d}  {N * K * 2 / 1e6:6.1f} MB:"]
    t = timeit(lambda: ops.gemm_skinny(wp, x, out, epi, True))
    row.append(f"packed {t:6.1f} us ({N * K * 2 / t / 1e6:4.2f} TB/s)")
    for pad in (0, 64, 192):
        big = torch.zeros(N, K + pad, dtype=torch.float16, device=dev)
        big[:, :K] = w
        wv = big[:, :K]
        for co in (1, 0):
            ops.set_option("skinny.coalesce", co)
            t = timeit(lambda: ops.gemm_skinny(wv, x, out, epi, False))
            row.append(f"rows+{pad}{'c' if co else 'm'} {t:6.1f}")
        ops.set_option("skinny.coalesce", 1)
        del big
    print("  ".join(row), flush=True)
