#!/usr/bin/env python3
"""Would hi + lo split MFMA operands close the gap to north_star's 1e-3?  (CPU only; uses the oracle's rounding emulation.)

The HIP path's distance from the fp32 reference is the sum of ONE rounding to the 16-bit compute type per operand hand-over
(measured == predicted by oracle.emulate_rounding to < 1 % at every layer).  Handing an operand over as a hi + lo
pair of 16-bit values (acc = hi.W + lo.W: a second MFMA pass over that operand, i.e. the GEMM runs at K' = 2 K) makes that hand-over
exact to ~2^-22.  This tool measures, at FULL depth on configurations C1 / C2, what the logits error would become if

    norm          only the operands that carry the normalised stream (q|k|v, fc1, gate/up A operands) were split,
    norm+attn_out +mlp_act   every A operand of every layer linear were split (all four GEMMs of a layer at 2 K),

by treating exactly those hand-over sites as exact in the emulating oracle and leaving every other rounding (q, k, v, P, the ViT
input pixels, the projector) in place.  The cost side is arithmetic: a split GEMM is the same kernel at twice the K
(tools/bench_kernels.py --extra-shapes measures it), so `norm` doubles q|k|v + gate/up (+ SigLIP qkv, fc1) and the full variant
doubles every layer linear.

    python tools/split_operand_study.py [--configs c1,c2] [--dtype f16] [--out profiles/r03_split_operand_study.txt]
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402
from tools.parity_report import logit_stats, sample_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    t0 = time.perf_counter()
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    out = open(args.out, "w") if args.out else sys.stdout
    print(f"# tools/split_operand_study.py — predicted logits error with hi + lo split operands ({args.dtype} compute type), full depth, "
          f"{torch.get_num_threads()} host threads; weights built in {time.perf_counter() - t0:.0f} s", file=out)
    arms = [("all hand-overs rounded (the HIP path)", ()), ("norm operands split", ("norm",)),
            ("norm + attn_out + mlp_act split (every layer linear at 2K)", ("norm", "attn_out", "mlp_act"))]
    for c in args.configs.split(","):
        n, w, h = {"c1": (1, 336, 336), "c2": (1, 1344, 896)}[c]
        u8, ids, _ = sample_inputs(cfg, n, w, h)
        pix = torch.from_numpy(siglip_normalize(u8))
        ref = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
        print(f"\n=== {c.upper()}: {n} x ({w}x{h}) -> {u8.shape[0]} ViT inputs, S = {ids.shape[1] + u8.shape[0] * (cfg.tokens_per_tile - 1)}; "
              f"max|logit| = {ref.abs().max().item():.3f} ===", file=out)
        print(f"{'arm':<62} {'max-abs':>10} {'/ max|logit|':>13} {'rel RMS':>10} {'argmax':>7}", file=out)
        for name, sites in arms:
            with O.emulate_rounding(dt, exact_sites=sites):
                lg = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
            a, nrm, r, eq = logit_stats(lg, ref)
            print(f"{name:<62} {a:10.3e} {nrm:13.3e} {r:10.3e} {str(eq):>7}", file=out)
            out.flush()
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
