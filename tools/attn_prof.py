#!/usr/bin/env python3
"""Per-segment cycle accounting of the LDS-DMA attention kernel (Llama shape, causal, S = 7187).

Needs a library built with -DLMI_ATTN_PROF (s_memtime marks around the segments of the tile loop):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLMI_ATTN_PROF -o tools/_ab/libprof.so leopard_amd/csrc/capi.hip
    python tools/attn_prof.py tools/_ab/libprof.so [lds_pad_bytes]
Segments: 1 wait+barrier, 2 DMA issue, 3 QK^T (to MFMA results), 4 mask+max+rescale, 5 exp/sum/pack, 6 PV (to MFMA results)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402

lib = _lib.bind(sys.argv[1])
ops = Ops(lib)
if len(sys.argv) > 2:
    ops.set_option("attn.lds_pad", int(sys.argv[2]))
S, H, KV, D = 7187, 32, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(S, (H + 2 * KV) * D, generator=g).to(torch.float16).cuda()
out = torch.empty(S, H * D, dtype=torch.float16, device="cuda")
cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
nblk = ((S + 127) // 128) * H
buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
lib.lmi_debug_set_prof_buffer.argtypes = [C.c_void_p]
assert lib.lmi_debug_set_prof_buffer(buf.data_ptr()) == 0
for _ in range(2):
    ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:], out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
torch.cuda.synchronize()
t = buf.view(nblk, 4, 8).cpu().double()
tiles = 105792 * 4          # wave-tiles (upper bound: waves skip their fully masked tile)
tot = t[:, :, 1:7].sum((0, 1))
names = ["wait+barrier", "dma issue", "qk", "mask+max+rescale", "exp", "pv"]
print("cycles per wave-tile (sum over waves / wave-tiles):")
for n, v in zip(names, tot.tolist()):
    print(f"  {n:18s} {v / tiles:8.1f}")
print(f"  {'total':18s} {tot.sum().item() / tiles:8.1f}")
