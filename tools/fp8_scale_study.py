#!/usr/bin/env python3
"""Would per-32-element E8M0 block scales make the fp8 line usably accurate?  (CPU only; oracle rounding emulation.)

The fp8 schedule (BASELINE configs[4]; leopard_amd/fp8.py) quantises the A operands and the weights of the ViT / LLM layer linears to
e4m3 with ONE static power-of-two scale per tensor / activation site.  v_mfma_scale_f32_32x32x64_f8f6f4 can apply one E8M0 scale per
lane, i.e. per (row, 32-element k-block), for free.  This tool predicts — with the emulating oracle, which matches the HIP fp8 path to
< 1 % (tests/test_gpu_parity.py::test_*fp8*) — the logits error of both scale granularities at FULL depth:

    python tools/fp8_scale_study.py [--configs c1] [--out profiles/r03_fp8_scale_study.txt]

A block scale only helps when a tensor's dynamic range exceeds what one scale can place inside e4m3's ~18 binades; the rounding
step of a floating-point format is relative (2^-4 for e4m3) whatever the scale.
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
    ap.add_argument("--configs", default="c1")
    ap.add_argument("--out", default=None)
    ap.add_argument("--outliers", default="1", help="comma list of outlier magnitudes: for each value X != 1, 0.1 %% of the hidden channels of every "
                    "normalised stream (4 of 4096 in the LLM, 2 of 1152 in the ViT) carry activations X times larger — the norm gains of those "
                    "channels are multiplied by X and the matching weight columns of the consuming linears divided by X, so the fp32 function "
                    "is unchanged and only the operands' dynamic range grows (the structure real LLM residual streams have)")
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    out = open(args.out, "w") if args.out else sys.stdout

    def inject(X):
        """Multiply the gains of the outlier channels by X and divide the consuming weight columns by X (in place; X = 1 / previous undoes)."""
        g = torch.Generator().manual_seed(77)
        tc, vc = cfg.text_config, cfg.vision_config
        ch_l = torch.randperm(tc.hidden_size, generator=g)[:max(1, tc.hidden_size // 1000)]
        ch_v = torch.randperm(vc.hidden_size, generator=g)[:max(1, vc.hidden_size // 1000) + 1]
        for i in range(tc.num_hidden_layers):
            p = f"language_model.model.layers.{i}."
            for norm, lins in (("input_layernorm", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")),
                               ("post_attention_layernorm", ("mlp.gate_proj", "mlp.up_proj"))):
                Wt[p + norm + ".weight"][ch_l] *= X
                for l in lins:
                    Wt[p + l + ".weight"][:, ch_l] /= X
        for i in range(vc.num_hidden_layers):
            p = f"vision_tower.vision_model.encoder.layers.{i}."
            for norm, lins in (("layer_norm1", ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj")), ("layer_norm2", ("mlp.fc1",))):
                Wt[p + norm + ".weight"][ch_v] *= X
                Wt[p + norm + ".bias"][ch_v] *= X
                for l in lins:
                    Wt[p + l + ".weight"][:, ch_v] /= X
        return len(ch_l), len(ch_v)
    print(f"# tools/fp8_scale_study.py — predicted logits error of the fp8 schedule (e4m3 operands + weights of the layer linears, f16 elsewhere), "
          f"full depth, {torch.get_num_threads()} host threads", file=out)
    arms = [("f16 path (no fp8)", None, 0), ("e4m3, one scale per tensor (the engine)", torch.float8_e4m3fn, 0),
            ("e4m3, E8M0 scale per 32 elements", torch.float8_e4m3fn, 32)]
    for c, X in [(c, float(x)) for c in args.configs.split(",") for x in args.outliers.split(",")]:
        n, w, h = {"c1": (1, 336, 336), "c2": (1, 1344, 896)}[c]
        u8, ids, _ = sample_inputs(cfg, n, w, h)
        pix = torch.from_numpy(siglip_normalize(u8))
        nl, nv = inject(X) if X != 1 else (0, 0)
        ref = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
        print(f"\n=== {c.upper()}: max|logit| = {ref.abs().max().item():.3f}" + (f"; OUTLIER CHANNELS x{X:g}: {nl} of {cfg.text_config.hidden_size} "
              f"(LLM norms -> q|k|v, gate/up) and {nv} of {cfg.vision_config.hidden_size} (ViT norms -> q|k|v, fc1), function unchanged" if X != 1 else
              "; no outlier channels (the synthetic weights as they are)") + " ===", file=out)
        print(f"{'arm':<44} {'max-abs':>10} {'/ max|logit|':>13} {'rel RMS':>10} {'argmax':>7} {'s':>6}", file=out)
        for name, od, blk in arms:
            t0 = time.perf_counter()
            with O.emulate_rounding(torch.float16, operand_dtype=od, fp8_block=blk):
                lg = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
            a, nrm, r, eq = logit_stats(lg, ref)
            print(f"{name:<44} {a:10.3e} {nrm:13.3e} {r:10.3e} {str(eq):>7} {time.perf_counter() - t0:6.0f}", file=out)
            out.flush()
        if X != 1:
            inject(1.0 / X)
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
