import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from leopard_amd.config import full_config
from leopard_amd.engine import LeopardEngine
from leopard_amd.ops import Ops
from leopard_amd.weights import EngineWeights, SynthSource
from tests.test_gpu_parity import sample_inputs
DEV = torch.device("cuda:0")
cfg, ops, dtype = full_config(), Ops(), torch.float16
W = EngineWeights.build(cfg, SynthSource(cfg, ops, DEV, dtype), dtype)
eng = LeopardEngine(cfg, W, ops=ops, device=DEV)
u8, ids, _ = sample_inputs(cfg, 1, 336, 336, seed=0)
tiles = torch.from_numpy(u8).to(DEV)
for mode, dp in (("fast", True), ("lo4", True), ("lo4", False)):
    eng.precision, eng.decode_precision = mode, dp
    out = eng.generate(ids, tiles, max_new_tokens=8, eos_token_id=())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.generate(ids, tiles, max_new_tokens=64, eos_token_id=())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    eng.prefill(ids, tiles); torch.cuda.synchronize(); t1 = time.perf_counter(); eng.prefill(ids, tiles); torch.cuda.synchronize(); pf = time.perf_counter() - t1
    print(mode, "decode_precision", dp, "hl", eng._gen_cache._decode_state.hl, f"{(dt - pf) / 63 * 1e3:.3f} ms/token", out[0, -8:].tolist())
samples = [(ids, tiles)] * 8
for mode in ("fast", "lo4"):
    eng.precision, eng.decode_precision = mode, True
    eng.generate_batch(samples, max_new_tokens=4, eos_token_id=())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = eng.generate_batch(samples, max_new_tokens=33, eos_token_id=())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("batch 8", mode, "hl", eng._batch_states[8].hl, f"{dt * 1e3:.1f} ms for prefill + 32 steps", outs[0][0, -4:].tolist(), outs[7][0, -4:].tolist())
