#!/usr/bin/env python3
"""Error budget of the HIP prefill path against the fp32 CPU oracle, layer by layer (runs on the GPU box).

    python tools/parity_report.py [--configs c1,c2,c4] [--out profiles/r03_error_growth.txt]

For each configuration (C1 = 1 x 336x336, S = 228; C2 = 1 x 1344x896, N = 7 ViT inputs, S = 1242), at FULL depth and width
(27 SigLIP + 32 Llama-3.1-8B layers, synthetic seeded weights), it runs
  * the fp32 oracle                                   (the reference's CPU arithmetic),
  * the oracle with 16-bit rounding emulated at the kernel hand-over points (oracle.emulate_rounding) = PREDICTED budget,
  * the HIP path in fp16 and bf16                                                                       = MEASURED,
(c4 = Leopard-Idefics2 on BASELINE configs[3]: 4 x 1344x896 -> 980x653, S = 312, 27 NaViT + 3 perceiver + 32 Mistral layers)
and prints, after the embeddings and after every layer, the relative RMS error of the fp32 residual stream
rms(x - x_ref) / rms(x_ref), then the logit errors (max-abs, max-abs / max|logit|, relative RMS, argmax agreement).
"HIP vs emulated oracle" is what remains once the 16-bit operand roundings are accounted for (accumulation order,
v_exp/v_rcp approximations): it must be several times smaller than either of them against fp32.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402


def rel_rms(a, b):
    a, b = a.float().reshape(-1), b.float().reshape(-1)
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def logit_stats(got, ref):
    d = (got.float() - ref.float())
    return (d.abs().max().item(), d.abs().max().item() / ref.abs().max().item(), rel_rms(got, ref),
            int(got.argmax()) == int(ref.argmax()))


def sample_inputs(cfg, n_images, w, h, seed=0):
    from PIL import Image
    from leopard_amd.synth import synth_image_u8, synth_prompt_ids
    from leopard_amd.tiler import tile_sample, to_u8_tiles
    imgs = [Image.fromarray(synth_image_u8(seed + i, w, h)) for i in range(n_images)]
    vit_inputs, plan = tile_sample(imgs)
    u8 = to_u8_tiles(vit_inputs)
    ids = synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=seed)
    return u8, torch.from_numpy(ids).reshape(1, -1), plan


def hip_run(cfg, ops, dtype, ids, u8, dev, split=False):
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device=dev)
    eng.split_operands = split
    trace = []
    eng.trace = lambda name, x: trace.append((name, x.detach().float().cpu().clone()))
    res = eng.prefill(ids.to(dev), torch.from_numpy(u8).to(dev))
    torch.cuda.synchronize()
    del eng, W
    torch.cuda.empty_cache()
    return res.logits_last.float().cpu(), dict(trace)


def oracle_run(cfg, Wt, ids, u8, emulate=None):
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    trace = []
    t0 = time.perf_counter()
    with O.emulate_rounding(emulate, trace=trace):
        logits = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(u8)), Wt, cfg, last_only=True)[0, 0]
    return logits, {k: v for k, v in trace}, time.perf_counter() - t0


def report(name, n_images, w, h, ops, Wt, dev, out, emu_bf16):
    cfg = full_config()
    u8, ids, plan = sample_inputs(cfg, n_images, w, h)
    ref, tr_ref, t_ref = oracle_run(cfg, Wt, ids, u8)
    emu16, tr_e16, _ = oracle_run(cfg, Wt, ids, u8, torch.float16)
    runs = {"oracle(fp16 roundings)": (emu16, tr_e16)}
    if emu_bf16:
        emub, tr_eb, _ = oracle_run(cfg, Wt, ids, u8, torch.bfloat16)
        runs["oracle(bf16 roundings)"] = (emub, tr_eb)
    h16, tr_h16 = hip_run(cfg, ops, torch.float16, ids, u8, dev)
    hb, tr_hb = hip_run(cfg, ops, torch.bfloat16, ids, u8, dev)
    runs["HIP fp16"] = (h16, tr_h16)
    runs["HIP bf16"] = (hb, tr_hb)
    runs["HIP fp16 split operands"] = hip_run(cfg, ops, torch.float16, ids, u8, dev, split=True)
    S = ids.shape[1] + u8.shape[0] * (cfg.tokens_per_tile - 1)
    print(f"\n=== {name}: {n_images} x ({w}x{h}) -> {u8.shape[0]} ViT inputs, S = {S}; 27 + 32 layers, full width; "
          f"fp32 oracle {t_ref:.1f} s on {torch.get_num_threads()} host threads ===", file=out)
    cols = list(runs)
    print("relative RMS error of the fp32 residual stream vs the fp32 oracle, rms(x - x_ref) / rms(x_ref):", file=out)
    print(f"{'after':>10} " + " ".join(f"{c:>24}" for c in cols) + f" {'HIP fp16 vs emulated':>22}", file=out)
    for key in tr_ref:
        if any(key not in runs[c][1] for c in cols):
            continue
        row = []
        for c in cols:
            x = runs[c][1][key]
            row.append(rel_rms(x.reshape(tr_ref[key].shape), tr_ref[key]))
        extra = rel_rms(tr_h16[key].reshape(tr_ref[key].shape), tr_e16[key])
        print(f"{key:>10} " + " ".join(f"{v:24.3e}" for v in row) + f" {extra:22.3e}", file=out)
    print("last-token logits (max-abs, max-abs / max|logit|, relative RMS, argmax equal); max|logit| = "
          f"{ref.abs().max().item():.3f}:", file=out)
    for c in cols:
        a, n, r, eq = logit_stats(runs[c][0], ref)
        print(f"  {c:>24} vs fp32 oracle : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    a, n, r, eq = logit_stats(h16, emu16)
    print(f"  {'HIP fp16':>24} vs oracle(fp16 roundings) : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    if emu_bf16:
        a, n, r, eq = logit_stats(hb, runs["oracle(bf16 roundings)"][0])
        print(f"  {'HIP bf16':>24} vs oracle(bf16 roundings) : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    out.flush()


def idefics2_c4_sample(cfg, n_img=4, seed=0):
    """BASELINE configs[3] sample as bench.py builds it: n_img x (1344x896) -> 980x653 (3220 patches, 64 latents each), S = 312."""
    import numpy as np
    from PIL import Image
    from leopard_amd.synth import synth_image_u8
    ims = [Image.fromarray(synth_image_u8(seed * 16 + i, 1344, 896)) for i in range(n_img)]
    L = cfg.perceiver_config.n_latents
    rng = np.random.default_rng(seed)
    ids = []
    for _ in range(n_img):
        ids += rng.integers(3, 32000, 6).tolist() + [cfg.image_token_id] * L
    ids += rng.integers(3, 32000, 32).tolist()
    return ims, torch.tensor([ids])


def report_idefics2(ops, dev, out):
    """Leopard-Idefics2 at FULL depth (27 NaViT SigLIP + 3 perceiver + 32 Mistral layers) on the C4 sample."""
    from leopard_amd.config import idefics2_full_config
    from leopard_amd.idefics2 import Idefics2Engine, Idefics2SynthSource, Idefics2Weights, preprocess_image_u8
    from oracle import idefics2_oracle as IO
    from oracle import leopard_oracle as O
    cfg = idefics2_full_config()
    ims, ids = idefics2_c4_sample(cfg)
    src = Idefics2SynthSource(cfg, ops, dev, torch.float16)
    Wt = {name: src.get(name).float().cpu() for name in src.specs}
    pix = [IO.image_processor(im, cfg.longest_edge) for im in ims]

    def oracle(emulate):
        trace = []
        t0 = time.perf_counter()
        with O.emulate_rounding(emulate, trace=trace):
            lg = IO.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
        return lg, dict(trace), time.perf_counter() - t0

    def hip(dtype):
        W = Idefics2Weights.build(cfg, Idefics2SynthSource(cfg, ops, dev, dtype), dtype)
        eng = Idefics2Engine(cfg, W, ops=ops, device=dev)
        trace = []
        eng.trace = lambda name, x: trace.append((name, x.detach().float().cpu().clone()))
        u8 = [torch.from_numpy(preprocess_image_u8(im, cfg.longest_edge).copy()) for im in ims]
        res = eng.prefill(ids, u8, keep_parts=True)
        torch.cuda.synchronize()
        tr = dict(trace)
        tr["image_features"] = res.parts["image_features"].float().cpu()
        lg = res.logits_last.float().cpu()
        del eng, W
        torch.cuda.empty_cache()
        return lg, tr
    ref, tr_ref, t_ref = oracle(None)
    e16, tr_e16, _ = oracle(torch.float16)
    eb, tr_eb, _ = oracle(torch.bfloat16)
    h16, tr_h16 = hip(torch.float16)
    hb, tr_hb = hip(torch.bfloat16)
    runs = {"oracle(fp16 roundings)": (e16, tr_e16), "oracle(bf16 roundings)": (eb, tr_eb), "HIP fp16": (h16, tr_h16), "HIP bf16": (hb, tr_hb)}
    print(f"\n=== Idefics2 C4: 4 x (1344x896) -> 980x653, 3220 patches and 64 latents each, S = {ids.shape[1]}; 27 + 3 + 32 layers, full width; "
          f"fp32 oracle {t_ref:.1f} s on {torch.get_num_threads()} host threads ===", file=out)
    cols = list(runs)
    print("relative RMS error vs the fp32 oracle, rms(x - x_ref) / rms(x_ref) (image_features = the 4 x 64 perceiver outputs; llm.* = the fp32 residual stream):", file=out)
    print(f"{'after':>15} " + " ".join(f"{c:>24}" for c in cols) + f" {'HIP fp16 vs emulated':>22}", file=out)
    for key in tr_ref:
        if key not in tr_h16:
            continue
        row = [rel_rms(runs[c][1][key].reshape(tr_ref[key].shape), tr_ref[key]) for c in cols]
        extra = rel_rms(tr_h16[key].reshape(tr_ref[key].shape), tr_e16[key])
        print(f"{key:>15} " + " ".join(f"{v:24.3e}" for v in row) + f" {extra:22.3e}", file=out)
    fmax = tr_ref["image_features"].abs().max().item()
    for c in cols:
        d = (runs[c][1]["image_features"].reshape(tr_ref["image_features"].shape) - tr_ref["image_features"]).abs().max().item()
        print(f"  image features {c:>24}: max-abs {d:.3e}, / max|feature| {d / fmax:.3e}", file=out)
    print(f"last-token logits (max-abs, max-abs / max|logit|, relative RMS, argmax equal); max|logit| = {ref.abs().max().item():.3f}:", file=out)
    for c in cols:
        a, n, r, eq = logit_stats(runs[c][0], ref)
        print(f"  {c:>24} vs fp32 oracle : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    a, n, r, eq = logit_stats(h16, e16)
    print(f"  {'HIP fp16':>24} vs oracle(fp16 roundings) : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    a, n, r, eq = logit_stats(hb, eb)
    print(f"  {'HIP bf16':>24} vs oracle(bf16 roundings) : {a:.3e}  {n:.3e}  {r:.3e}  {eq}", file=out)
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from leopard_amd.ops import Ops
    from leopard_amd.weights import SynthSource
    dev = torch.device("cuda:0")
    ops = Ops()
    cfg = full_config()
    Wt = None
    if any(c in ("c1", "c2") for c in args.configs.split(",")):
        src = SynthSource(cfg, ops, dev, torch.float16)       # synthetic values are exact in fp16 AND bf16: one host copy serves both
        Wt = {name: src.get(name).float().cpu() for name in src.specs}
    out = open(args.out, "w") if args.out else sys.stdout
    print("# tools/parity_report.py — HIP prefill path vs the fp32 CPU oracle, full depth (see the tool's docstring)", file=out)
    for c in args.configs.split(","):
        if c == "c1":
            report("C1", 1, 336, 336, ops, Wt, dev, out, emu_bf16=True)
        elif c == "c2":
            report("C2", 1, 1344, 896, ops, Wt, dev, out, emu_bf16=False)
        elif c == "c4":
            report_idefics2(ops, dev, out)
    if args.out:
        out.close()
        print(open(args.out).read())


if __name__ == "__main__":
    main()
