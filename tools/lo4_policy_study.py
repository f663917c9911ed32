#!/usr/bin/env python3
"""Which hand-over sites does the lo4 correction have to cover on a given sample?  (CPU only; round 6)

The lo4 schedule corrects every Llama layer-linear operand; each corrected GEMM pays + 25 % matrix time.  This tool prices the
sites one by one with the rounding-emulating oracle at FULL depth: the SigLIP tower, projector and merge run ONCE (fast-schedule roundings), then
the 32 Llama layers are re-run per arm with the correction emulated on a subset of the sites

    norm1 (q|k|v operand)   attn_out (o_proj operand)   norm2 (gate/up operand)   mlp_act (down_proj operand)      [optionally @layer ranges]

and the last-position logits are compared with the fp32 oracle's (the committed full-depth fixture).  The engine's `lo4_policy` is chosen from
these tables (leopard_amd/engine.py); the GPU tests assert the measured figure.

    python tools/lo4_policy_study.py --config c3 --arms all,none,attn_out+mlp_act,... [--out profiles/r06_lo4_policy_study_c3.txt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402
from tools.gen_fulldepth_fixtures import CASES, sample_inputs  # noqa: E402
from tools.parity_report import logit_stats  # noqa: E402

SITES = ("norm1", "attn_out", "norm2", "mlp_act")


def parse_arm(arm: str):
    """'all' | 'none' | 'attn_out+mlp_act@0-15+norm1' -> tuple of oracle lo_sites entries (tower-qualified).  A trailing '/rows=R' restricts the
    correction to the last R rows of the sequence (handled by the caller)."""
    arm = arm.split("/")[0]
    if arm == "none":
        return ()
    if arm == "all":
        return tuple("llm." + s for s in SITES)
    out = []
    for tok in arm.split("+"):
        name = tok.split("@")[0]
        assert name in SITES, tok
        out.append("llm." + tok)
    return tuple(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--arms", default="all,none")
    ap.add_argument("--vit-lo4", type=int, default=0, help="1 = the SigLIP layer linears corrected too (engine.lo4_vit)")
    ap.add_argument("--vit-fp8", type=int, default=0, help="1 = the SigLIP layer linears on e4m3 operands (per-tensor scales): what the tower's roundings cost "
                                                           "the LAST row's logits when they are 128 x larger (they reach it only through the softmax average)")
    ap.add_argument("--weights-cache", default="/tmp/leopard_oracle_weights.pt")
    ap.add_argument("--embeds-cache", default=None, help="torch.save file of the merged embeddings (emulated tower) + fp32 reference logits")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    t0 = time.perf_counter()
    if os.path.exists(args.weights_cache):
        Wt = torch.load(args.weights_cache, mmap=True)
    else:
        Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
        torch.save(Wt, args.weights_cache)
    n, w, h, n_vit, S = CASES[args.config]
    u8, ids, _ = sample_inputs(cfg, n, w, h, seed=args.seed)
    pix = torch.from_numpy(siglip_normalize(u8))
    ec = args.embeds_cache or f"/tmp/lo4_policy_{args.config}_s{args.seed}_{args.dtype}_v{args.vit_lo4}_f{args.vit_fp8}.pt"
    if os.path.exists(ec):
        st = torch.load(ec)
        emb, pos, ref = st["emb"], st["pos"], st["ref"]
    else:
        fx = os.path.join(REPO, "tests", "golden", f"{args.config}_full_depth.npz")
        ref = None
        if args.seed == 0 and os.path.exists(fx):
            z = np.load(fx)
            key = [k for k in z.files if k.startswith("logits") and "fp32" in k]
            if key:
                ref = torch.from_numpy(z[key[0]]).float().reshape(-1)
        if ref is None:
            ref = O.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
        vit_sites = ("vit.norm", "vit.attn_out", "vit.mlp_act") if args.vit_lo4 else ()
        with O.emulate_rounding(dt, lo_sites=vit_sites, operand_dtype=torch.float8_e4m3fn if args.vit_fp8 else None):
            feats = O.siglip_vision_tower(pix, Wt, cfg)
        with O.emulate_rounding(dt):
            vis = O.projector(feats, Wt)
            emb, _, pos = O.embed_and_merge(ids, vis, Wt, cfg)
        torch.save({"emb": emb, "pos": pos, "ref": ref}, ec)
    out = open(args.out, "a") if args.out else sys.stdout
    print(f"# tools/lo4_policy_study.py --config {args.config} --seed {args.seed} --dtype {args.dtype} --vit-lo4 {args.vit_lo4} --vit-fp8 {args.vit_fp8}: S = {emb.shape[1]}, "
          f"max|logit| = {ref.abs().max().item():.3f}; {torch.get_num_threads()} host threads; tower + merge in {time.perf_counter() - t0:.0f} s", file=out)
    print(f"{'corrected Llama sites (lo4 = fp4 e2m1 residual x fp4 weight image)':<72} {'max-abs':>10} {'/ max|logit|':>13} {'rel RMS':>10} {'argmax':>7} {'s':>6}", file=out)
    out.flush()
    for arm in args.arms.split(","):
        t1 = time.perf_counter()
        rows = int(arm.split("/rows=")[1]) if "/rows=" in arm else 0
        with O.emulate_rounding(dt, lo_sites=parse_arm(arm), lo_row_start=max(emb.shape[1] - rows, 0) if rows else 0):
            lg = O.llama_forward(emb, pos, Wt, cfg, last_only=True)[0, -1]
        a, nrm, r, eq = logit_stats(lg, ref)
        print(f"{arm:<72} {a:10.3e} {nrm:13.3e} {r:10.3e} {str(eq):>7} {time.perf_counter() - t1:6.0f}", file=out)
        out.flush()
    if args.out:
        out.close()


if __name__ == "__main__":
    main()
