#!/usr/bin/env python3
"""Full-depth oracle fixture of the DECODE branch (test infrastructure; HOST cores, no GPU; round 6).

    python tools/gen_decode_fixtures.py [--samples 4] [--tokens 16] [--out tests/golden/decode_full_depth.npz]

For ``samples`` C1 inputs (1 x 336x336 + 32-token prompt, S = 228; seeds 0, 16, 32, ...) the fp32 CPU oracle (= the reference's arithmetic:
evaluations/models/llava_multiimg_siglip_anyres.py:448-454 greedy generate over the forward of :201-361, decode branch :291-320) produces
``tokens`` greedy tokens at FULL depth (27 + 32 layers).  The oracle keeps no KV cache: every step is a full fp32 forward of the grown sequence
(3.2 TFLOP, ~6 s on 8 cores), the SigLIP tower and projector run once per sample.  Stored per sample and step t = 0 .. tokens - 1 (t = 0 is the
prefill's last position): the greedy token, the top-8 ids and logits and max|logit| — so that tests/test_gpu_decode_fixture.py can feed the
HIP engine the ORACLE's tokens (teacher forcing: no divergence after a near tie) and state the error of every decode step's logits."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=16)
    ap.add_argument("--weights-cache", default="/tmp/leopard_oracle_weights.pt")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "decode_full_depth.npz"))
    args = ap.parse_args()
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    cfg = full_config()
    if os.path.exists(args.weights_cache):
        Wt = torch.load(args.weights_cache, mmap=True)
    else:
        Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    T = args.tokens
    out = {"seeds": [], "prompt_ids": [], "tiles_sha256": [], "tokens": [], "top_ids": [], "top_logits": [], "max_abs_logit": []}
    t0 = time.perf_counter()
    for j in range(args.samples):
        seed = 16 * j
        u8, ids, _ = sample_inputs(cfg, 1, 336, 336, seed=seed)
        pix = torch.from_numpy(siglip_normalize(u8))
        with torch.no_grad():
            feats = O.projector(O.siglip_vision_tower(pix, Wt, cfg), Wt)
            cur = ids.reshape(1, -1).clone()
            toks, tids, tlog, mx = [], [], [], []
            for t in range(T):
                emb, _, pos = O.embed_and_merge(cur, feats, Wt, cfg)
                lg = O.llama_forward(emb, pos, Wt, cfg, last_only=True)[0, -1]
                top = lg.topk(8)
                toks.append(int(top.indices[0])); tids.append(top.indices.numpy()); tlog.append(top.values.numpy()); mx.append(float(lg.abs().max()))
                cur = torch.cat([cur, top.indices[:1].reshape(1, 1)], dim=1)
                print(f"sample {j} step {t}: token {toks[-1]}  margin {float(top.values[0] - top.values[1]):.4f}  ({time.perf_counter() - t0:.0f} s)", flush=True)
        out["seeds"].append(seed); out["prompt_ids"].append(ids.numpy().reshape(-1)); out["tiles_sha256"].append(np.frombuffer(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest(), dtype=np.uint8))
        out["tokens"].append(toks); out["top_ids"].append(np.stack(tids)); out["top_logits"].append(np.stack(tlog)); out["max_abs_logit"].append(mx)
    np.savez_compressed(args.out, seeds=np.array(out["seeds"]), prompt_ids=np.stack(out["prompt_ids"]), tiles_sha256=np.stack(out["tiles_sha256"]),
                        tokens=np.array(out["tokens"], dtype=np.int64), top_ids=np.stack(out["top_ids"]).astype(np.int64),
                        top_logits=np.stack(out["top_logits"]).astype(np.float32), max_abs_logit=np.array(out["max_abs_logit"], dtype=np.float32),
                        oracle_seconds=np.array([time.perf_counter() - t0]))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
