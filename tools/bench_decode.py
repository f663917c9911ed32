#!/usr/bin/env python3
"""Decode-step latency after a prefill (greedy generate, EVAL:448-454): ms/token at a short and at the C3 context."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leopard_amd.config import full_config  # noqa: E402
from leopard_amd.engine import KVCache, LeopardEngine  # noqa: E402
from leopard_amd.ops import Ops  # noqa: E402
from leopard_amd.synth import synth_prompt_ids  # noqa: E402
from leopard_amd.weights import EngineWeights, SynthSource  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tokens", type=int, default=32)
ap.add_argument("--tiles", type=int, nargs="*", default=[1, 42])
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = full_config()
ops = Ops()
W = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, torch.float16), torch.float16)
eng = LeopardEngine(cfg, W, ops=ops, device=dev)
for n_tiles in args.tiles:
    per_image = [n_tiles] if n_tiles <= 8 else [7] * (n_tiles // 7)
    ids = torch.from_numpy(synth_prompt_ids(per_image, cfg, seed=1)).reshape(1, -1)
    tiles = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (sum(per_image), 364, 364, 3), dtype=np.uint8)).to(dev)
    S = ids.shape[1] + sum(per_image) * (cfg.tokens_per_tile - 1)
    cache = KVCache(cfg, S + 2 * args.tokens + 8, torch.float16, dev)
    res = eng.prefill(ids, tiles, cache=cache)
    nxt = int(res.logits_last.argmax())
    for _ in range(3):
        nxt = int(eng.decode_step(nxt, cache).argmax())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.tokens):
        nxt = int(eng.decode_step(nxt, cache).argmax())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.tokens
    # the generate() loop: token and position stay on the device, the host only reads the new id
    st = eng._decode_state(cache)
    eng._decode_seed(st, cache, nxt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.tokens):
        eng._decode_run(st, cache)
        cache.length += 1
        nxt = int(st.tok.item())
    dt2 = (time.perf_counter() - t0) / args.tokens
    print(f"context S={S}: decode_step {dt * 1e3:.2f} ms/token, generate loop {dt2 * 1e3:.2f} ms/token ({1 / dt2:.1f} tok/s); "
          f"weight stream floor 16.06 GB / 6.3 TB/s = 2.55 ms")
