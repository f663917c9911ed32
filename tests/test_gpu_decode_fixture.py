"""-m gpu: the DECODE branch at full depth against a committed oracle fixture (round 6).

tests/golden/decode_full_depth.npz (tools/gen_decode_fixtures.py): for 4 C1 samples the fp32 CPU oracle's greedy tokens and, per step, its top-8 ids /
logits and max|logit| over 16 steps (step 0 = the prefill's last position; the reference's decode branch: llava_multiimg_siglip_anyres.py:291-320,
448-454, fp32).  The HIP engine is fed the ORACLE's tokens (teacher forcing), so every step's logits can be compared whatever a near tie does to the
greedy choice.  Two schedules: fast (one rounding per operand hand-over, prefill and decode) and lo4 — whose decode steps run on operand PAIRS
(csrc/skinny.h "hl": T(x) and T(x - T(x)) rows into the same sums; LeopardEngine.decode_hl) — asserted at north_star's 1e-3 on every step."""
import hashlib
import os

import numpy as np
import pytest
import torch

from leopard_amd.config import full_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_full_depth.npz")


@pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/decode_full_depth.npz not generated")
def test_decode_steps_at_full_depth_against_the_fp32_oracle():
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.ops import Ops
    from leopard_amd.weights import EngineWeights, SynthSource
    from tests.test_gpu_parity import sample_inputs
    cfg, ops, dtype = full_config(), Ops(), torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, torch.device(DEV), dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device=torch.device(DEV))
    z = np.load(FIX)
    n, T = z["tokens"].shape
    errs = {}
    for mode in ("fast", "lo4"):
        eng.precision = mode
        e = np.zeros((n, T))
        agree = np.zeros((n, T), dtype=bool)
        for j in range(n):
            u8, ids, _ = sample_inputs(cfg, 1, 336, 336, seed=int(z["seeds"][j]))
            assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest() == z["tiles_sha256"][j].tobytes()
            assert (ids.numpy().reshape(-1) == z["prompt_ids"][j]).all()
            cache = KVCache(cfg, 512, dtype, DEV)
            res = eng.prefill(ids, torch.from_numpy(u8).to(DEV), cache=cache)
            logits = res.logits_last.float().reshape(-1)
            for t in range(T):
                if t > 0:
                    logits = eng.decode_step(int(z["tokens"][j, t - 1]), cache).float()
                got = logits.cpu()[torch.from_numpy(z["top_ids"][j, t])]
                e[j, t] = (got - torch.from_numpy(z["top_logits"][j, t])).abs().max().item() / float(z["max_abs_logit"][j, t])
                agree[j, t] = int(logits.argmax()) == int(z["tokens"][j, t])
            assert cache._decode_state.hl == (mode == "lo4")
        errs[mode] = (e, agree)
        margins = (z["top_logits"][:, :, 0] - z["top_logits"][:, :, 1]) / z["max_abs_logit"]
        print(f"[decode full depth {mode}] {n} samples x {T} steps (step 0 = prefill): normalised max error of the oracle's top-8 logits: prefill "
              f"{e[:, 0].max():.3e}; decode steps worst {e[:, 1:].max():.3e}, median {np.median(e[:, 1:]):.3e}; greedy token == oracle's on "
              f"{int(agree.sum())}/{agree.size} steps (smallest oracle margin {margins.min():.2e} of the logit scale)")
        # a flipped greedy choice may only be a near tie: the oracle's top-2 margin inside twice the measured error
        flipped = ~agree
        assert (margins[flipped] <= 2.0 * e[flipped] + 1e-6).all()
    e_fast, e_lo4 = errs["fast"][0], errs["lo4"][0]
    assert e_lo4.max() <= 1.0e-3, e_lo4.max()                           # north_star's figure on EVERY step of the precision mode, prefill and decode
    assert np.median(e_lo4[:, 1:]) < 0.7 * np.median(e_fast[:, 1:])     # and it is the decode steps' own hand-over roundings that went
    assert e_fast.max() <= 3.0e-3
