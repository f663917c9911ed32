import os, sys, torch, numpy as np
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
z = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/c1_full_depth.npz"))
ref = torch.from_numpy(z["logits_fp32"])
for mode in ("fast", "lo4"):
    eng.precision = mode
    outs = {}
    for mid in (1, 0):
        ops.set_option("gemm.mid_m", mid)
        o = eng.prefill(ids, tiles).logits_last.float().cpu().reshape(-1)
        outs[mid] = o
        print(mode, "mid_m", mid, "err", f"{(o - ref).abs().max().item() / ref.abs().max().item():.3e}")
    print(mode, "bit-identical across geometries:", torch.equal(outs[0], outs[1]), "max diff", (outs[0] - outs[1]).abs().max().item())
ops.set_option("gemm.mid_m", 1)
