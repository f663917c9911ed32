"""Multi-rank paths on ONE MI355X: two processes share cuda:0 and talk over gloo (RCCL needs one device per rank), which is
enough to check through the real kernels that the tile-sharded vision encode + all-gather and the tensor-parallel LLM
(leopard_amd.dist) reproduce the single-rank results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from leopard_amd import dist as D
    from leopard_amd.config import mid_config
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.ops import Ops
    from leopard_amd.synth import synth_prompt_ids
    from leopard_amd.weights import EngineWeights, SynthSource
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D.init(backend="gloo")
    ops, cfg, dtype = Ops(), mid_config(), torch.float16
    src = SynthSource(cfg, ops, dev, dtype)
    eng = LeopardEngine(cfg, EngineWeights.build(cfg, src, dtype, tp_rank=rank, tp_size=world), ops=ops, device=dev)
    tiles = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (5, 364, 364, 3), dtype=np.uint8)).to(dev)
    ids = torch.from_numpy(synth_prompt_ids([2, 3], cfg, seed=4)).reshape(1, -1)
    S = ids.shape[1] + 5 * (cfg.tokens_per_tile - 1)
    cache = KVCache(cfg, S + 8, dtype, dev, tp_size=world)
    vis = D.encode_images_sharded(eng, tiles)
    res = eng.prefill(ids, None, cache=cache, visual_tokens=vis)
    step = eng.decode_step(int(res.logits_last.argmax()), cache).clone()
    ref = None
    if rank == 0:
        one = LeopardEngine(cfg, EngineWeights.build(cfg, src, dtype), ops=ops, device=dev)
        c1 = KVCache(cfg, S + 8, dtype, dev)
        r1 = one.prefill(ids, tiles, cache=c1)
        s1 = one.decode_step(int(r1.logits_last.argmax()), c1)
        scale = float(r1.logits_last.abs().max())
        ref = (bool(torch.equal(vis, one.encode_images(tiles))), float((res.logits_last - r1.logits_last).abs().max()) / scale,
               float((step - s1).abs().max()) / scale, int(res.logits_last.argmax()) == int(r1.logits_last.argmax()))
    torch.cuda.synchronize()
    out.put((rank, res.logits_last.cpu().tolist(), ref))
    D.barrier()


def test_tensor_parallel_and_tile_sharding_two_ranks_one_gpu():
    mp.set_start_method("spawn", force=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, ref), (_, l1, _) = res
    assert l0 == l1
    vis_equal, d_prefill, d_decode, same_argmax = ref
    assert vis_equal                                   # tile-sharded encode + all-gather: bit-identical
    assert d_prefill <= 2.5e-3 and d_decode <= 2.5e-3 and same_argmax
