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
ap.add_argument("--batch", type=int, nargs="*", default=[], help="also time the batched decode step for these batch sizes (1 = the batch-1 loop as the base)")
ap.add_argument("--skip-single", action="store_true")
ap.add_argument("--eng", action="append", default=[], help="engine attribute name=int, e.g. decode_prefetch_bytes=33554432")
ap.add_argument("--opt", action="append", default=[], help="library option key=value (lmi_set_option), e.g. attn.decode_split_tiles=8")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = full_config()
ops = Ops()
for kv in args.opt:
    ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
W = EngineWeights.build(cfg, SynthSource(cfg, ops, dev, torch.float16), torch.float16)
eng = LeopardEngine(cfg, W, ops=ops, device=dev)
for kv in args.eng:
    assert hasattr(eng, kv.split("=")[0]), kv
    setattr(eng, kv.split("=")[0], int(kv.split("=")[1]))
for n_tiles in ([] if args.skip_single else args.tiles):
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

# ---- batched decode: B sequences per step through the pooled cache and the skinny-M projections ------------------
if args.batch:
    base_tok_s = None
    for n_tiles in args.tiles:
        per_image = [n_tiles] if n_tiles <= 8 else [7] * (n_tiles // 7)
        S = None
        for B in args.batch:
            samples = []
            for j in range(B):
                ids = torch.from_numpy(synth_prompt_ids(per_image, cfg, seed=10 + j)).reshape(1, -1)
                tiles = torch.from_numpy(np.random.default_rng(j).integers(0, 256, (sum(per_image), 364, 364, 3), dtype=np.uint8)).to(dev)
                samples.append((ids, tiles))
            S = samples[0][0].shape[1] + sum(per_image) * (cfg.tokens_per_tile - 1)
            if B == 1:
                cache = eng._generation_cache(S + args.tokens + 8)
                res = eng.prefill(samples[0][0], samples[0][1], cache=cache)
                st = eng._decode_state(cache)
                eng._decode_seed(st, cache, int(res.logits_last.argmax()))
                eng._decode_run(st, cache); cache.length += 1
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.tokens):
                    eng._decode_run(st, cache)
                    cache.length += 1
                    _ = int(st.tok.item())
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / args.tokens * 1e3
                base_tok_s = 1e3 / ms
                print(f"batch 1 (captured batch-1 step), S={S}: {ms:.2f} ms/step = {base_tok_s:.1f} tok/s", flush=True)
                continue
            # prefill the batch once through generate_batch's own path (1 new token), then time the captured batch step
            eng.generate_batch(samples, max_new_tokens=2, eos_token_id=())
            st = eng._batch_states[B]
            st.pos.fill_(S); st.k_len.fill_(S + 1)
            for _ in range(3):
                eng._batch_decode_run(st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.tokens):
                eng._batch_decode_run(st)
                _ = st.tok.tolist()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.tokens * 1e3
            tok_s = B * 1e3 / ms
            rel = f", x{tok_s / base_tok_s:.2f} the batch-1 rate" if base_tok_s else ""
            print(f"batch {B}, S={S}: {ms:.2f} ms/step = {tok_s:.1f} tok/s{rel} (weight stream floor 2.55 ms/step)", flush=True)
            del eng._batch_states[B]
            torch.cuda.empty_cache()
