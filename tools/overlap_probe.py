#!/usr/bin/env python3
"""Does a collective-sized transport kernel get CU time beside the production GEMMs?  (1 GPU; run on the GPU box.)

The tensor-parallel prefill (LeopardEngine._llm_layers_tp) puts chunk c's all-gather / reduce-scatter on a side stream under chunk
c + 1's GEMMs.  Those GEMMs hold ONE 512-thread workgroup with 128 KiB of LDS on every CU (216 VGPRs per lane: 80 registers per SIMD
lane and 32 KiB of LDS are left), so whether a transport kernel is scheduled at all is a property of the hardware dispatcher that no
multi-GPU node was available to show.  This probe runs the C3 gate/up GEMM (7187 x 28672 x 4096, staggered 256x256) in a loop on
the main stream and, on a side stream, a stand-in for the transport kernel: lmi_debug_copy of one exchange buffer (26 MB = one row
chunk of the 16-bit all-gather at 8 ranks... any --mb) on W workgroups of 256 threads, no LDS.  It reports
    the copy alone, the GEMM alone, both together (copy time under the GEMM, GEMM time beside the copy)
for several W.  Reading: copy-under-GEMM ~ copy-alone  -> the transport overlaps freely; copy-under-GEMM ~ GEMM duration -> it only
runs in the gaps between GEMM workgroups (the exchange then serialises behind a GEMM launch: more, smaller row chunks help).

    python tools/overlap_probe.py [--mb 26] [--wgs 8 32 64 256] [--gemms 6]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd import _lib  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=26.0)
    ap.add_argument("--wgs", type=int, nargs="*", default=[8, 32, 64, 256])
    ap.add_argument("--gemms", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = Ops()
    lib = ops.lib
    g = torch.Generator().manual_seed(0)
    M, N, K = 7187, 28672, 4096
    a = torch.randn(M, K, generator=g).to(torch.float16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.float16).to(dev)
    out = torch.empty(M, N // 2, dtype=torch.float16, device=dev)
    nbytes = int(args.mb * 1e6) // 16 * 16
    src = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    main_s, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)

    def gemm_loop():
        for _ in range(args.gemms):
            ops.gemm(a, w, out, epilogue=_lib.EPI_SWIGLU)

    def copy(wgs, stream):
        rc = lib.lmi_debug_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), nbytes, wgs, C.c_void_p(stream.cuda_stream))
        assert rc == 0, lib.lmi_last_error()

    def timed(fn_main=None, fn_side=None):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        side.wait_stream(main_s)
        if fn_main:
            e[0].record(main_s)
            fn_main()
            e[1].record(main_s)
        if fn_side:
            with torch.cuda.stream(side):
                e[2].record(side)
                fn_side()
                e[3].record(side)
        torch.cuda.synchronize()
        return (e[0].elapsed_time(e[1]) if fn_main else None, e[2].elapsed_time(e[3]) if fn_side else None)

    gemm_loop()
    torch.cuda.synchronize()
    t_gemm = min(timed(fn_main=gemm_loop)[0] for _ in range(args.reps))
    print(f"GEMM alone: {args.gemms} x gate/up = {t_gemm:.3f} ms ({t_gemm / args.gemms * 1e3:.0f} us each)")
    print(f"{'workgroups':>10} {'copy alone ms':>14} {'GB/s':>8} {'copy under GEMM ms':>19} {'GB/s':>8} {'GEMMs beside copy ms':>21} {'GEMM slowdown':>14}")
    for wgs in args.wgs:
        copy(wgs, side)
        alone = min(timed(fn_side=lambda: copy(wgs, side))[1] for _ in range(args.reps))
        both = [timed(fn_main=gemm_loop, fn_side=lambda: copy(wgs, side)) for _ in range(args.reps)]
        t_g = min(b[0] for b in both)
        t_c = sorted(b[1] for b in both)[len(both) // 2]
        print(f"{wgs:>10} {alone:14.3f} {2 * nbytes / alone / 1e6:8.0f} {t_c:19.3f} {2 * nbytes / t_c / 1e6:8.0f} {t_g:21.3f} {t_g / t_gemm - 1:13.1%}")


if __name__ == "__main__":
    main()
