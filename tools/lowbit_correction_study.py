#!/usr/bin/env python3
"""What does a LOW-BIT second k-loop phase buy?  (CPU only; the oracle's rounding emulation — round 5)

The HIP path's distance from the fp32 reference is one rounding of the A operand to the 16-bit compute type per hand-over
(DESIGN.md 2.1).  The split-operand precision mode removes it with a second 16-bit pass (2 K: 1.92 x the step).  The residual
x - T(x) only needs a few bits: this tool predicts, at FULL depth, the logits error when the residual is handed over as an MX
block-scaled low-bit image (per-32 E8M0 scale along K) and multiplied with a low-bit image of the weight (one E8M0 scale per
weight row) into the same accumulators — v_mfma_scale_f32_32x32x64_f8f6f4 at 4 x (fp4 / fp6) or 2 x (fp8) the 16-bit rate.

    python tools/lowbit_correction_study.py [--configs c1,c2] [--dtype f16] [--arms ...] [--out profiles/r05_lowbit_correction_study_c1.txt]
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

ALL = ("norm", "attn_out", "mlp_act")
ARMS = {
    "none": ("all hand-overs rounded once (the fast schedule)", dict()),
    "e2m1": ("fp4 e2m1 residual x fp4 weight image, every layer linear", dict(lo_sites=ALL, lo_fmt="e2m1")),
    "e2m1-noattn": ("fp4 e2m1, norm + mlp_act operands only (attention output rounded once)", dict(lo_sites=("norm", "mlp_act"), lo_fmt="e2m1")),
    "e2m1-no-llm-mlp": ("fp4 e2m1 everywhere except the Llama down_proj operand (SwiGLU output rounded once)",
                        dict(lo_sites=("norm", "attn_out", "vit.mlp_act"), lo_fmt="e2m1")),
    "e2m1-no-vit": ("fp4 e2m1 on the Llama layers only (SigLIP on the fast schedule)", dict(lo_sites=("llm.norm", "llm.attn_out", "llm.mlp_act"), lo_fmt="e2m1")),
    "e2m1-no-llm-attn": ("fp4 e2m1 everywhere except the Llama o_proj operand", dict(lo_sites=("norm", "vit.attn_out", "mlp_act"), lo_fmt="e2m1")),
    "e2m1-wblock": ("fp4 e2m1, every layer linear, per-32 block scales on the weight image too", dict(lo_sites=ALL, lo_fmt="e2m1", lo_wblock=32)),
    "e2m3": ("fp6 e2m3 residual x fp6 weight image, every layer linear", dict(lo_sites=ALL, lo_fmt="e2m3")),
    "e4m3": ("fp8 e4m3 residual x fp8 weight image, every layer linear", dict(lo_sites=ALL, lo_fmt="e4m3")),
    "exact": ("every layer-linear operand exact (hi + lo 16-bit pair: the 2 K precision mode)", dict(exact_sites=ALL)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--arms", default="none,e2m1,e2m1-noattn,e2m3,exact")
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights-cache", default=None, help="torch.save file of the fp32 oracle weights (built on first use, memory-mapped afterwards)")
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    t0 = time.perf_counter()
    if args.weights_cache and os.path.exists(args.weights_cache):
        Wt = torch.load(args.weights_cache, mmap=True)
    else:
        Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
        if args.weights_cache:
            torch.save(Wt, args.weights_cache)
    out = open(args.out, "w") if args.out else sys.stdout
    print(f"# tools/lowbit_correction_study.py — predicted logits error with a low-bit correction pass ({args.dtype} compute type), full depth, "
          f"{torch.get_num_threads()} host threads; weights built in {time.perf_counter() - t0:.0f} s", file=out)
    for c in args.configs.split(","):
        n, w, h = {"c1": (1, 336, 336), "c2": (1, 1344, 896)}[c]
        u8, ids, _ = sample_inputs(cfg, n, w, h)
        pix = torch.from_numpy(siglip_normalize(u8))
        ref = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
        print(f"\n=== {c.upper()}: {n} x ({w}x{h}) -> {u8.shape[0]} ViT inputs, S = {ids.shape[1] + u8.shape[0] * (cfg.tokens_per_tile - 1)}; "
              f"max|logit| = {ref.abs().max().item():.3f} ===", file=out)
        print(f"{'arm':<92} {'max-abs':>10} {'/ max|logit|':>13} {'rel RMS':>10} {'argmax':>7} {'s':>6}", file=out)
        for key in args.arms.split(","):
            name, kw = ARMS[key]
            t1 = time.perf_counter()
            with O.emulate_rounding(dt, **kw):
                lg = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
            a, nrm, r, eq = logit_stats(lg, ref)
            print(f"{name:<92} {a:10.3e} {nrm:13.3e} {r:10.3e} {str(eq):>7} {time.perf_counter() - t1:6.0f}", file=out)
            out.flush()
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
