#!/usr/bin/env python3
"""Reference point, not product code: torch's scaled_dot_product_attention (the ROCm flash-attention back ends) on the two
prefill attention shapes, next to lmi_attn_varlen_fwd on the same operands."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from leopard_amd.ops import Ops  # noqa: E402

dev = "cuda:0"
ops = Ops()


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for dtype in (torch.float16, torch.bfloat16):
    # Llama: S = 7187, 32 query heads / 8 kv heads, d = 128, causal
    S, H, KV, D = 7187, 32, 8, 128
    qkv = torch.randn(S, (H + 2 * KV) * D, device=dev).to(dtype)
    out = torch.empty(S, H * D, dtype=dtype, device=dev)
    cu = torch.tensor([0, S], dtype=torch.int32, device=dev)
    mine = lambda: ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:], out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    q = qkv[:, :H * D].view(S, H, D).transpose(0, 1).unsqueeze(0).contiguous()
    k = qkv[:, H * D:(H + KV) * D].view(S, KV, D).transpose(0, 1).unsqueeze(0).contiguous()
    v = qkv[:, (H + KV) * D:].view(S, KV, D).transpose(0, 1).unsqueeze(0).contiguous()
    kr, vr = k.repeat_interleave(H // KV, dim=1), v.repeat_interleave(H // KV, dim=1)
    lib = lambda: F.scaled_dot_product_attention(q, kr, vr, is_causal=True)
    flop = 2 * H * D * S * (S + 1)
    t_lib, t_mine = timed(lib), timed(mine)
    ref = lib().squeeze(0).transpose(0, 1).reshape(S, H * D)
    mine()
    err = (out.float() - ref.float()).abs().max().item()
    print(f"llama causal S={S} {str(dtype)[6:]}: library {t_lib:.3f} ms ({flop / t_lib / 1e9:.0f} TF/s) | lmi {t_mine:.3f} ms ({flop / t_mine / 1e9:.0f} TF/s) | max diff {err:.1e}")
    # SigLIP: 42 sequences of 676, 16 heads, d = 72, full
    n, T, Hv, Dv = 42, 676, 16, 72
    qkv2 = torch.randn(n * T, 3 * Hv * Dv, device=dev).to(dtype)
    out2 = torch.empty(n * T, Hv * Dv, dtype=dtype, device=dev)
    cu2 = torch.arange(0, (n + 1) * T, T, dtype=torch.int32, device=dev)
    mine2 = lambda: ops.attention(qkv2[:, :1152], qkv2[:, 1152:2304], qkv2[:, 2304:], out2, cu2, cu2, T, Hv, Hv, Dv, Dv ** -0.5, False, True)
    q2, k2, v2 = (qkv2[:, i * 1152:(i + 1) * 1152].view(n, T, Hv, Dv).transpose(1, 2).contiguous() for i in range(3))
    lib2 = lambda: F.scaled_dot_product_attention(q2, k2, v2)
    flop2 = 4 * n * T * T * Hv * Dv
    t_lib, t_mine = timed(lib2), timed(mine2)
    print(f"siglip 42x676 d72 {str(dtype)[6:]}: library {t_lib:.3f} ms ({flop2 / t_lib / 1e9:.0f} TF/s) | lmi {t_mine:.3f} ms ({flop2 / t_mine / 1e9:.0f} TF/s)")
