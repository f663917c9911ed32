#!/usr/bin/env python3
"""Run one GEMM shape a few times with a forced geometry (for rocprofv3 --pmc passes).
usage: python tools/run_one_gemm.py CFG [M N K epilogue]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd.ops import Ops  # noqa: E402

cfg = int(sys.argv[1])
M, N, K, epi = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (7187, 28672, 4096, 3)
ops = Ops()
ops.set_option("gemm.config", cfg)
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
out = torch.zeros(M, N // 2 if epi == 3 else N, dtype=torch.float32 if epi in (1, 2) else torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm(a, w, out, epilogue=epi)
torch.cuda.synchronize()
