#!/usr/bin/env python3
"""Full-depth oracle fixture of Leopard-Idefics2 on BASELINE configs[3]'s sample (test infrastructure; HOST cores, no GPU).

    python tools/gen_idefics2_fixture.py [--out tests/golden/c4_idefics2_full_depth.npz]

C4 = 4 x (1344x896) -> 980x653 NaViT images (3220 patches, 64 perceiver latents each), S = 312; 27 NaViT SigLIP + 3 perceiver + 32 Mistral-7B
layers at full width, the seeded synthetic parameters of leopard_amd.synth (bit-identical to what the GPU generates).  Writes the fp32 oracle's
last-position logits (= the reference's arithmetic, evaluations/models/idefics2_multiimg.py:22-30,88-97 over third-party Idefics2), the logits of
the oracle that emulates the kernels' 16-bit hand-over roundings (the PREDICTED budget), probe rows of the perceiver's image features, the token ids
and the SHA-256 of the preprocessed u8 images — so that tests/test_gpu_idefics2.py and bench.py's c4 entries can state parity from committed data
instead of recomputing 2 x 23 TFLOP on the GPU box's host at every run (round 6)."""
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

from leopard_amd.config import idefics2_full_config  # noqa: E402
from tools.parity_report import idefics2_c4_sample  # noqa: E402

PROBE_ROWS = (0, 63, 64, 127, 191, 255)              # rows of the [4 * 64, 4096] image features kept


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "c4_idefics2_full_depth.npz"))
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    from leopard_amd.idefics2 import preprocess_image_u8
    from leopard_amd.synth import idefics2_param_specs, synth_array
    from oracle import idefics2_oracle as IO
    from oracle import leopard_oracle as O
    cfg = idefics2_full_config()
    ims, ids = idefics2_c4_sample(cfg)
    u8 = [preprocess_image_u8(im, cfg.longest_edge) for im in ims]
    sha = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in u8)).digest()
    t0 = time.perf_counter()
    specs = list(idefics2_param_specs(cfg))
    with ThreadPoolExecutor(args.threads) as pool:
        arrs = list(pool.map(lambda s: synth_array(*s), specs))
    Wt = {s[0]: torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32) for s, a in zip(specs, arrs)}
    del arrs
    print(f"weights in {time.perf_counter() - t0:.0f} s", flush=True)
    pix = [IO.image_processor(im, cfg.longest_edge) for im in ims]
    t1 = time.perf_counter()
    ref, parts = IO.prefill_logits(ids, pix, Wt, cfg, last_only=True, return_parts=True)
    sec = time.perf_counter() - t1
    with O.emulate_rounding(torch.float16):
        emu = IO.prefill_logits(ids, pix, Wt, cfg, last_only=True)[0, 0]
    feats = parts["image_features"].reshape(-1, parts["image_features"].shape[-1])
    np.savez_compressed(args.out, meta=np.array([len(ims), 1344, 896, ids.shape[1]]), ids=ids.numpy(), images_sha256=np.frombuffer(sha, dtype=np.uint8),
                        logits_fp32=ref[0, 0].numpy().astype(np.float32), logits_emu_fp16=emu.numpy().astype(np.float32),
                        feature_probe_rows=np.array(PROBE_ROWS), feature_probe=feats[list(PROBE_ROWS)].numpy().astype(np.float32),
                        feature_max_abs=np.array([feats.abs().max().item()], dtype=np.float32), oracle_seconds=np.array([sec]),
                        oracle_threads=np.array([torch.get_num_threads()]))
    print(f"wrote {args.out}: S = {ids.shape[1]}, max|logit| = {ref.abs().max().item():.3f}, predicted fp16 error "
          f"{(emu - ref[0, 0]).abs().max().item() / ref.abs().max().item():.3e}, oracle {sec:.0f} s")


if __name__ == "__main__":
    main()
