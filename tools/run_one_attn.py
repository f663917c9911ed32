#!/usr/bin/env python3
"""Run the Llama-shape causal attention a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd.ops import Ops  # noqa: E402

ops = Ops()
S, H, KV, D = 7187, 32, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(S, (H + 2 * KV) * D, generator=g).to(torch.float16).cuda()
out = torch.empty(S, H * D, dtype=torch.float16, device="cuda")
cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
for _ in range(3):
    ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:], out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
torch.cuda.synchronize()
