#!/usr/bin/env python3
"""Full-depth oracle fixtures for the BASELINE configurations (test infrastructure; runs on HOST cores, no GPU).

    python tools/gen_fulldepth_fixtures.py [--cases c1,c2,c3] [--out tests/golden]

For C1 (1 x 336x336, S = 228), C2 (1 x 1344x896 -> 7 ViT inputs, S = 1242) and C3 (the benchmarked configuration:
6 x 1344x896 -> 42 ViT inputs, S = 7187) at FULL depth and width (27 SigLIP + 32 Llama-3.1-8B layers, the seeded synthetic
parameters of leopard_amd.synth, which the GPU generates bit-identically) this runs ``oracle.prefill_logits(..., last_only=True)``

  * in fp32                                           = the reference's CPU arithmetic (EVAL:248-333 semantics),
  * with the kernels' 16-bit hand-over roundings emulated (``oracle.emulate_rounding``; fp16 for every case, bf16 and the
    e4m3-operand schedule for C1)                      = the PREDICTED error budget,

and writes ``tests/golden/<case>_full_depth.npz``: the last-position logits of every run (fp32), the per-layer relative RMS
distance emulated-vs-fp32 of the fp32 residual stream (on the probe rows below), and the fp32 oracle's residual stream on a
few PROBE ROWS after every layer, so that the -m gpu tests can state the HIP path's error layer by layer and on the logits
without recomputing 140 TFLOP on the host at every run (tests/test_gpu_parity.py).  Inputs are regenerated from their seeds by
the tests; the fixture carries their SHA-256 so a drift of the tiler / prompt synthesiser cannot go unnoticed.

The oracle is this repo's own restatement (pinned to the reference by oracle/gen_golden.py -> tests/golden/*), so unlike
gen_golden.py this script does not need /root/reference and may run on any box with ~48 GB of RAM (C3: ~140 TFLOP per run,
~5 min on 8 cores at 0.5 TFLOP/s)."""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402
from leopard_amd.synth import param_specs, synth_array  # noqa: E402

CASES = {"c1": (1, 336, 336, 1, 228), "c2": (1, 1344, 896, 7, 1242), "c3": (6, 1344, 896, 42, 7187)}   # images, W, H -> ViT inputs, S


def probe_rows(kind: str, shape) -> tuple:
    """Rows of the residual stream kept per layer.  ViT stream [N, 676, D]: first / last tile x rows {0, 337, 675};
    LLM stream [1, S, D]: rows {S//3, S//2, S-2, S-1} (the last row is the one the logits come from)."""
    if kind == "vit":
        n = shape[0]
        tiles = sorted({0, n - 1})
        return [(t, r) for t in tiles for r in (0, 337, 675)]
    S = shape[1]
    return [(0, r) for r in sorted({S // 3, S // 2, max(S - 2, 0), S - 1})]


class ProbeTrace(list):
    """Stand-in for the oracle's trace list: keeps only the probe rows of each traced tensor (the full C3 trace is 7 GB)."""

    def append(self, item):
        name, x = item
        idx = probe_rows("vit" if name.startswith("vit") else "llm", x.shape)
        rows = torch.stack([x[t, r] for t, r in idx]).clone()
        super().append((name, rows))


def sample_inputs(cfg, n_images, w, h, seed=0):
    from PIL import Image
    from leopard_amd.synth import synth_image_u8, synth_prompt_ids
    from leopard_amd.tiler import tile_sample, to_u8_tiles
    imgs = [Image.fromarray(synth_image_u8(seed + i, w, h)) for i in range(n_images)]
    vit_inputs, plan = tile_sample(imgs)
    u8 = to_u8_tiles(vit_inputs)
    ids = synth_prompt_ids(plan.vit_inputs_per_image, cfg, seed=seed)
    return u8, torch.from_numpy(ids).reshape(1, -1), plan


def host_weights(cfg, threads):
    specs = list(param_specs(cfg))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        arrs = list(pool.map(lambda s: synth_array(*s), specs))
    W = {s[0]: torch.from_numpy(a) for s, a in zip(specs, arrs)}
    print(f"synthetic parameters: {sum(a.size for a in arrs) / 1e9:.2f} G values in {time.perf_counter() - t0:.0f} s", flush=True)
    return W


def rel_rms(a, b):
    a, b = a.float().reshape(-1), b.float().reshape(-1)
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def run(cfg, W, ids, pix, emulate=None, operand_dtype=None):
    from oracle import leopard_oracle as O
    tr = ProbeTrace()
    t0 = time.perf_counter()
    with O.emulate_rounding(emulate, trace=tr, operand_dtype=operand_dtype):
        logits = O.prefill_logits(ids, pix, W, cfg, last_only=True)[0, 0].clone()
    return logits, list(tr), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="c1,c2,c3")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from leopard_amd.tiler import siglip_normalize
    cfg = full_config()
    W = host_weights(cfg, args.threads)
    for case in args.cases.split(","):
        n, w, h, n_vit, S = CASES[case]
        u8, ids, plan = sample_inputs(cfg, n, w, h)
        assert u8.shape[0] == n_vit and ids.shape[1] + n_vit * (cfg.tokens_per_tile - 1) == S
        pix = torch.from_numpy(siglip_normalize(u8))
        ref, tr_ref, t_ref = run(cfg, W, ids, pix)
        print(f"[{case}] fp32 oracle: {t_ref:.0f} s on {args.threads} threads; max|logit| {ref.abs().max():.4f} argmax {int(ref.argmax())}", flush=True)
        out = {
            "meta": np.asarray([n, w, h, n_vit, S], dtype=np.int64),
            "ids": ids.numpy(),
            "tiles_sha256": np.frombuffer(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest(), dtype=np.uint8),
            "oracle_seconds": np.asarray([t_ref]), "oracle_threads": np.asarray([args.threads]),
            "logits_fp32": ref.numpy(),
            "trace_names": np.asarray([k for k, _ in tr_ref]),
            "probe_fp32": np.concatenate([v.numpy().reshape(-1) for _, v in tr_ref]),
            "probe_offsets": np.cumsum([0] + [v.numel() for _, v in tr_ref]).astype(np.int64),
            "probe_width": np.asarray([v.shape[-1] for _, v in tr_ref], dtype=np.int64),
        }
        emus = [("fp16", torch.float16, None)]
        if case == "c1":
            emus += [("bf16", torch.bfloat16, None), ("fp8", torch.float16, torch.float8_e4m3fn)]
        for tag, dt, op in emus:
            lg, tr, t = run(cfg, W, ids, pix, dt, op)
            d = (lg - ref)
            print(f"[{case}] oracle with {tag} hand-over roundings: {t:.0f} s; logits vs fp32: max-abs {d.abs().max():.3e} "
                  f"normalised-max {d.abs().max() / ref.abs().max():.3e} rel-rms {rel_rms(lg, ref):.3e} "
                  f"argmax equal {int(lg.argmax()) == int(ref.argmax())}", flush=True)
            out[f"logits_emu_{tag}"] = lg.numpy()
            out[f"trace_relrms_emu_{tag}"] = np.asarray([rel_rms(a[1], b[1]) for a, b in zip(tr, tr_ref)])
        path = os.path.join(args.out, f"{case}_full_depth.npz")
        np.savez_compressed(path, **out)
        print(f"[{case}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    main()
