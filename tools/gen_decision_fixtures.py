#!/usr/bin/env python3
"""Decision fixtures: what the fp32 reference arithmetic DECIDES on many samples (test infrastructure; runs on HOST cores, no GPU).

    python tools/gen_decision_fixtures.py [--c1 32] [--c2 8] [--weights-cache /tmp/oracle_w.pt]

VERDICT r04 item 8: a logit error figure says little about usefulness; the first greedy token and the top-5 set do.  For N samples of
BASELINE config C1 (1 x 336x336 image, S = 228; seeds 1000 ..) and a few of C2 (1 x 1344x896, S = 1242) at FULL depth (27 SigLIP +
32 Llama-3.1-8B layers, the seeded synthetic parameters) this runs the fp32 oracle (the reference's CPU arithmetic, EVAL:248-333) and
stores, per sample, the top-8 token ids + logits of the last position and the margin top1 - top2, plus — for the record — what the
ORACLE's own rounding emulations decide (fp16 hand-overs, fp16 + the fp4 correction phase, e4m3 operands).  tests/test_gpu_decisions.py
runs the HIP schedules (fast / lo4 / fp8) on the same seeds and counts argmax agreement and top-5 overlap against this file.
Inputs are regenerated from their seeds; the fixture carries the prompt ids and the tile hashes."""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from leopard_amd.config import full_config  # noqa: E402
from tools.gen_fulldepth_fixtures import sample_inputs  # noqa: E402

SEED0 = 1000
ALL = ("norm", "attn_out", "mlp_act")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c1", type=int, default=32)
    ap.add_argument("--c2", type=int, default=8)
    ap.add_argument("--weights-cache", default=None)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "decisions_full_depth.npz"))
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    if args.weights_cache and os.path.exists(args.weights_cache):
        W = torch.load(args.weights_cache, mmap=True)
    else:
        W = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    out = {}
    arms = [("fp16", dict()), ("fp16_lo4", dict(lo_sites=ALL)), ("fp8", dict(operand_dtype=torch.float8_e4m3fn))]
    part_path = args.out + ".partial.npz"            # resumable: every finished sample is saved (a full run is ~1.5 h of host time)
    done = dict(np.load(part_path)) if os.path.exists(part_path) else {}
    for case, n, (ni, w, h) in (("c1", args.c1, (1, 336, 336)), ("c2", args.c2, (1, 1344, 896))):
        t0 = time.perf_counter()
        for j in range(n):
            key = f"{case}_{j}"
            if f"{key}_top_ids" in done:
                continue
            u8, ids, _ = sample_inputs(cfg, ni, w, h, seed=SEED0 + 16 * j)
            pix = torch.from_numpy(siglip_normalize(u8))
            ref = O.prefill_logits(ids, pix, W, cfg, last_only=True)[0, 0]
            v, i = ref.topk(8)
            done[f"{key}_top_ids"], done[f"{key}_top_logits"], done[f"{key}_ids"] = i.numpy(), v.numpy(), ids.numpy().reshape(-1)
            done[f"{key}_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest(), dtype=np.uint8)
            for a, kw in arms:
                if case == "c2" and a != "fp16_lo4":
                    continue                                   # C2 arms other than lo4: predicted by the C1 set; keep the host time bounded
                with O.emulate_rounding(torch.float16, **kw):
                    lg = O.prefill_logits(ids, pix, W, cfg, last_only=True)[0, 0]
                done[f"{key}_emu_{a}"] = lg.topk(8)[1].numpy()
            np.savez(part_path, **done)
            print(f"[{case} {j + 1}/{n}] {time.perf_counter() - t0:.0f} s  top1 {int(i[0])} margin {float(v[0] - v[1]):.4f}", flush=True)
        out[f"{case}_top_ids"] = np.stack([done[f"{case}_{j}_top_ids"] for j in range(n)])
        out[f"{case}_top_logits"] = np.stack([done[f"{case}_{j}_top_logits"] for j in range(n)])
        out[f"{case}_tiles_sha256"] = np.stack([done[f"{case}_{j}_sha"] for j in range(n)])
        out[f"{case}_ids"] = np.stack([done[f"{case}_{j}_ids"] for j in range(n)])
        for a, _ in arms:
            if f"{case}_0_emu_{a}" in done:
                out[f"{case}_emu_{a}_top_ids"] = np.stack([done[f"{case}_{j}_emu_{a}"] for j in range(n)])
    out["seed0"] = np.asarray([SEED0])
    np.savez_compressed(args.out, **out)
    if os.path.exists(part_path):
        os.remove(part_path)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
