#!/usr/bin/env python3
"""Per-segment cycle accounting of the 64-rows-per-wave attention kernel (attention64.h; Llama shape, causal, S = 7187).

Needs a library built with -DLMI_ATTN_PROF:  python tools/attn64_prof.py tools/_ab/libprof.so
Segments (s_memtime marks; per wave and 64-key tile = two pipelined regions): 1 wait + barrier, 2 QK half of the regions (16 MFMAs + 8
exponential slices + 4 LDS-DMA pieces each), 3 PV half (16 MFMAs + 8 exponential slices + the maxima), 4 region tail (row sums, half-wave
exchange, ballot), 5 masks, 6 settle (flush + rescale)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402

lib = _lib.bind(sys.argv[1])
ops = Ops(lib)
S, H, KV, D = 7187, 32, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(S, (H + 2 * KV) * D, generator=g).to(torch.float16).cuda()
out = torch.empty(S, H * D, dtype=torch.float16, device="cuda")
cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
nq = (S + 255) // 256
nblk = nq * H
buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
lib.lmi_debug_set_prof_buffer.argtypes = [C.c_void_p]
assert lib.lmi_debug_set_prof_buffer(buf.data_ptr()) == 0
for _ in range(2):
    ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:], out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
torch.cuda.synchronize()
t = buf.view(nblk, 4, 8).cpu().double()
# wave-tiles in the main loop: wave w of q-block qb computes my_tiles = min(n_tiles, (qb*256 + 64 w + 63) // 64 + 1) tiles, the loop runs my_tiles - 1
tiles = 0
for qb in range(nq):
    n_tiles = (min(S, qb * 256 + 256) + 63) // 64
    for w in range(4):
        if qb * 256 + 64 * w < S:
            tiles += max(0, min(n_tiles, (qb * 256 + 64 * w + 63) // 64 + 1) - 1)
tiles *= H
tot = t[:, :, 1:7].sum((0, 1))
names = ["wait+barrier", "QK half x2", "PV half x2", "region tail x2", "masks", "settle"]
print(f"cycles per wave-tile (sum over waves / {tiles} wave-tiles of the main loop; 64 MFMAs = 2048 matrix-pipe cycles per wave-tile):")
for n, v in zip(names, tot.tolist()):
    print(f"  {n:18s} {v / tiles:8.1f}")
print(f"  {'total':18s} {tot.sum().item() / tiles:8.1f}")
heavy = t[:H].sum(0)                                              # the 32 heaviest blocks (last q-block of every head): bid < H
print("heaviest q-block, per wave (cycles / 1e3):", [[round(x / H / 1e3, 1) for x in row[1:7]] for row in heavy.tolist()])
