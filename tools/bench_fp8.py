#!/usr/bin/env python3
"""fp8 (v_mfma_scale_f32_32x32x64_f8f6f4) vs fp16 (v_mfma_f32_32x32x16) GEMMs at the C3 prefill shapes (run on the GPU box).
  python tools/bench_fp8.py [--iters 20] [--rounds 3]
Interleaved rounds in one process, HIP-event timed on the launch stream, random data (never zeros); every fp8 result is checked
against the fp32 product of the dequantised operands."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402
from leopard_amd.weights import interleave_gate_up  # noqa: E402

DEV = "cuda:0"
F8 = torch.float8_e4m3fn
SHAPES = [
    ("llm gate/up swiglu", 7187, 28672, 4096, _lib.EPI_SWIGLU, 0),
    ("llm down  resid", 7187, 4096, 14336, _lib.EPI_RESIDUAL, 0),
    ("llm qkv   store", 7187, 6144, 4096, _lib.EPI_STORE, 0),
    ("llm o     resid", 7187, 4096, 4096, _lib.EPI_RESIDUAL, 0),
    ("vit qkv   store", 28392, 3456, 1152, _lib.EPI_STORE, 0),
    ("vit o     resid", 28392, 1152, 1152, _lib.EPI_RESIDUAL, 0),
    ("vit fc1   gelu", 28392, 4352, 1152, _lib.EPI_STORE, _lib.ACT_GELU_TANH),
    ("vit fc2   resid", 28392, 1152, 4352, _lib.EPI_RESIDUAL, 0),
]


def time_fn(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    ops = Ops()
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"{'shape':22s} {'M':>6s} {'N':>6s} {'K':>6s} |  fp16 ms  TF/s |  fp8 ms   TF/s | speed-up | fp8 max rel err vs fp32(dequantised)")
    tot16 = tot8 = 0.0
    for name, M, N, K, epi, act in SHAPES:
        a = torch.randn(M, K, generator=g, device=DEV)
        w = torch.randn(N, K, generator=g, device=DEV) * 0.25
        a16, w16 = a.half(), w.half()
        a8, w8 = a.clamp(-448, 448).to(F8), w.to(F8)
        if epi == _lib.EPI_SWIGLU:
            w16 = interleave_gate_up(w16[:N // 2], w16[N // 2:])
            w8i = interleave_gate_up(w8.view(torch.uint8)[:N // 2], w8.view(torch.uint8)[N // 2:])
        else:
            w8i = w8.view(torch.uint8)
        a8u = a8.view(torch.uint8)
        if epi == _lib.EPI_RESIDUAL:
            o16 = torch.zeros(M, N, device=DEV)
            o8 = torch.zeros(M, N, device=DEV)
        elif epi == _lib.EPI_SWIGLU:
            o16 = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
            o8 = torch.empty_like(o16)
        else:
            o16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
            o8 = torch.empty_like(o16)
        e = -6                                                     # keeps the outputs in fp16 range whatever K
        f16 = lambda: ops.gemm(a16, w16, o16, epilogue=epi, act=act)
        f8 = lambda: ops.gemm_fp8(a8u, w8i, o8, epilogue=epi, act=act, scale_exp=e)
        # correctness of the fp8 kernel (single launch into a fresh output)
        if epi == _lib.EPI_RESIDUAL:
            o8.zero_()
        f8()
        ref = (a8.float() @ w8.float().T) * 2.0 ** e
        if epi == _lib.EPI_SWIGLU:
            ref = torch.nn.functional.silu(ref[:, :N // 2]) * ref[:, N // 2:]
        elif act == _lib.ACT_GELU_TANH:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        err = ((o8.float() - ref).abs() / (1 + ref.abs())).max().item()
        del ref
        t16 = t8 = 1e9
        for _ in range(args.rounds):
            t16 = min(t16, time_fn(f16, args.iters))
            t8 = min(t8, time_fn(f8, args.iters))
        fl = 2.0 * M * N * K
        tot16 += t16
        tot8 += t8
        print(f"{name:22s} {M:6d} {N:6d} {K:6d} | {t16:7.3f} {fl / t16 / 1e9:6.0f} | {t8:7.3f} {fl / t8 / 1e9:6.0f} | {t16 / t8:7.2f}x | {err:.2e}")
        del a, w, a16, w16, a8, w8, o16, o8
        torch.cuda.empty_cache()
    print(f"sum of the eight shapes: fp16 {tot16:.3f} ms, fp8 {tot8:.3f} ms ({tot16 / tot8:.2f}x)")


if __name__ == "__main__":
    main()
