#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the C3 prefill shapes (run on the GPU box).
  python tools/bench_kernels.py gemm [--cfgs 0,1,2,3,4] [--dtype bf16]
  python tools/bench_kernels.py attn
Every timing is HIP-event based on the launch stream, interleaved rounds (variant x round) in one process;
each GEMM variant is also checked against torch fp32 on random data (races in the LDS ring show up here)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402

DEV = "cuda:0"
# (name, M, N, K, epilogue, act)
GEMM_SHAPES = [
    ("llm gate/up swiglu", 7187, 28672, 4096, _lib.EPI_SWIGLU, 0),
    ("llm down  resid", 7187, 4096, 14336, _lib.EPI_RESIDUAL, 0),
    ("llm qkv   store", 7187, 6144, 4096, _lib.EPI_STORE, 0),
    ("llm o     resid", 7187, 4096, 4096, _lib.EPI_RESIDUAL, 0),
    ("vit qkv   store", 28392, 3456, 1152, _lib.EPI_STORE, 0),
    ("vit o     resid", 28392, 1152, 1152, _lib.EPI_RESIDUAL, 0),
    ("vit fc1   gelu", 28392, 4352, 1152, _lib.EPI_STORE, _lib.ACT_GELU_TANH),
    ("vit fc2   resid", 28392, 1152, 4352, _lib.EPI_RESIDUAL, 0),
]


SMALL_SHAPES = [
    ("m312 gate/up swiglu", 312, 28672, 4096, _lib.EPI_SWIGLU, 0),
    ("m312 down  resid", 312, 4096, 14336, _lib.EPI_RESIDUAL, 0),
    ("m312 qkv   store", 312, 6144, 4096, _lib.EPI_STORE, 0),
    ("m312 o     resid", 312, 4096, 4096, _lib.EPI_RESIDUAL, 0),
]


def time_fn(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_gemm(args):
    ops = Ops(_lib.bind(args.lib)) if args.lib else Ops()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    cfgs = [int(c) for c in args.cfgs.split(",")]
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'shape':22s} {'M':>6s} {'N':>6s} {'K':>6s} | " + " | ".join(f"cfg{c}: ms  TF/s  err" for c in cfgs))
    total = {c: 0.0 for c in cfgs}
    if args.group_m:
        ops.set_option("gemm.group_m", args.group_m)
    ops.set_option("gemm.order", args.order)
    base = SMALL_SHAPES if args.small else GEMM_SHAPES
    if args.extra_shapes:                                       # "name:M,N,K,epilogue;..."  epilogue: store | resid | swiglu
        epi_of = {"store": _lib.EPI_STORE, "resid": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}
        base = []
        for item in args.extra_shapes.split(";"):
            name, spec = item.split(":")
            m, n, k, e = spec.split(",")
            base.append((name, int(m), int(n), int(k), epi_of[e], 0))
    shapes = base if args.only < 0 else [base[args.only]]
    for name, M, N, K, epi, act in shapes:
        a = (torch.randn(M, K, generator=g)).to(dtype).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.02).to(dtype).to(DEV)
        n_out = N // 2 if epi == _lib.EPI_SWIGLU else N
        out_dtype = torch.float32 if epi in (_lib.EPI_RESIDUAL, _lib.EPI_STORE_F32) else dtype
        out = torch.zeros(M, n_out, dtype=out_dtype, device=DEV)
        # reference on a row sample
        rows = torch.randint(0, M, (64,), generator=g).to(DEV)
        lin = a[rows].float() @ w.float().T
        if epi == _lib.EPI_SWIGLU:
            lv = lin.view(64, N // 64, 2, 32)
            ref = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(64, N // 2)
        elif act == _lib.ACT_GELU_TANH:
            ref = torch.nn.functional.gelu(lin, approximate="tanh")
        else:
            ref = lin
        res = {}
        for c in cfgs:
            ops.set_option("gemm.config", c)
            out.zero_()
            ops.gemm(a, w, out, epilogue=epi, act=act)
            err = ((out[rows].float() - ref).abs() / (1 + ref.abs())).max().item()
            fn = lambda: ops.gemm(a, w, out, epilogue=epi, act=act)
            time_fn(fn, 2)
            res[c] = [err, []]
        for rnd in range(args.rounds):
            for c in cfgs:
                ops.set_option("gemm.config", c)
                res[c][1].append(time_fn(lambda: ops.gemm(a, w, out, epilogue=epi, act=act), args.iters))
        cells = []
        for c in cfgs:
            ms = sorted(res[c][1])[len(res[c][1]) // 2]
            total[c] += ms
            cells.append(f"{ms:7.3f} {2.0 * M * N * K / ms / 1e9:6.0f} {res[c][0]:.1e}")
        print(f"{name:22s} {M:6d} {N:6d} {K:6d} | " + " | ".join(cells), flush=True)
    ops.set_option("gemm.config", -1)
    print("sum of medians (one of each shape):", {c: round(v, 3) for c, v in total.items()})


def bench_attn(args):
    ops = Ops(_lib.bind(args.lib)) if args.lib else Ops()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if args.lds_pad:
        ops.set_option("attn.lds_pad", args.lds_pad)
    g = torch.Generator(device="cpu").manual_seed(0)
    S, H, KV, D = 7187, 32, 8, 128
    qkv = torch.randn(S, (H + 2 * KV) * D, generator=g).to(dtype).to(DEV)
    out = torch.empty(S, H * D, dtype=dtype, device=DEV)
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    fn = lambda: ops.attention(qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:], out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    res = {}
    for rows64 in (0, 1, 2, 0, 1, 2):                            # attention.h / attention64.h with 2 blocks per wave / with 1 block per wave
        ops.set_option("attn.rows64", rows64)
        time_fn(fn, 2)
        ms = time_fn(fn, args.iters)
        res[rows64] = out.float().clone()
        print(f"llama causal S={S} rows64={rows64}: {ms:.3f} ms  {2 * H * D * S * (S + 1) / ms / 1e9:.0f} TF/s (causal-counted)", flush=True)
    for v in (1, 2):
        d = (res[0] - res[v]).abs().max().item()
        print(f"  max |attention.h kernel - pipelined variant {v}| = {d:.3e} (outputs of magnitude {res[0].abs().max().item():.2f})")
    ops.set_option("attn.rows64", 0)
    n, T, Hv, Dv = 42, 676, 16, 72
    qkv2 = torch.randn(n * T, 3 * Hv * Dv, generator=g).to(dtype).to(DEV)
    out2 = torch.empty(n * T, Hv * Dv, dtype=dtype, device=DEV)
    cu2 = torch.arange(0, (n + 1) * T, T, dtype=torch.int32, device=DEV)
    fn2 = lambda: ops.attention(qkv2[:, :1152], qkv2[:, 1152:2304], qkv2[:, 2304:], out2, cu2, cu2, T, Hv, Hv, Dv, Dv ** -0.5, False, True)
    time_fn(fn2, 2)
    ms = time_fn(fn2, args.iters)
    print(f"siglip 42x676 d72: {ms:.3f} ms  {4 * n * T * T * Hv * Dv / ms / 1e9:.0f} TF/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "attn"])
    ap.add_argument("--cfgs", default="0,1,2,3,4")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--group-m", type=int, default=0)
    ap.add_argument("--order", type=int, default=0)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--extra-shapes", default="", help='custom shapes instead of the C3 set: "name:M,N,K,store|resid|swiglu;..."')
    ap.add_argument("--lds-pad", type=int, default=0)
    ap.add_argument("--lib", default="", help="A/B: path of another build of libleopard_amd.so")
    ap.add_argument("--only", type=int, default=-1, help="index of a single GEMM shape")
    a = ap.parse_args()
    bench_gemm(a) if a.what == "gemm" else bench_attn(a)
