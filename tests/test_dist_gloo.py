"""world_size-2 tests of the N>1 path on CPU (gloo): the sample split, the barrier / max-over-ranks timing reduce used
by bench.py, and the tile-sharded vision encode + all-gather (kernels run on the CPU logic emulator)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from leopard_amd import dist as D
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    from tests.emu_util import emu_ops
    from tests.test_emu_engine import micro_config
    r, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    # sample split == reference split_shard
    rows = list(range(17))
    mine = D.shard_records(rows, rank, world)
    # timing reduce
    tmax = D.max_over_ranks(1.0 + rank, "cpu")
    D.barrier()
    # tile-sharded encode
    ops = emu_ops()
    cfg = micro_config()
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    tiles = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (3, 28, 28, 3), dtype=np.uint8))
    vis = D.encode_images_sharded(eng, tiles)
    full = eng.encode_images(tiles)
    out.put((rank, mine, tmax, bool(torch.equal(vis, full)), tuple(vis.shape)))
    D.barrier()


def test_two_rank_gloo_path():
    mp.set_start_method("spawn", force=True)
    from tests.emu_util import emu_ops
    emu_ops()                                            # build the emulator library once, before forking
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, t0, ok0, sh0), (r1, m1, t1, ok1, sh1) = res
    assert m0 + m1 == list(range(17)) and len(m0) == 9           # len//2 + 1
    assert t0 == t1 == 2.0                                       # max over ranks
    assert ok0 and ok1 and sh0 == sh1 == (3, 128)                # all-gathered == unsharded, bit-exact


def test_tile_slices_balanced():
    from leopard_amd.dist import tile_slices
    assert tile_slices(42, 8) == [(0, 6), (6, 12), (12, 17), (17, 22), (22, 27), (27, 32), (32, 37), (37, 42)]
    assert tile_slices(1, 2) == [(0, 1), (1, 1)]
    for n in range(0, 60):
        for w in (1, 2, 4, 8):
            s = tile_slices(n, w)
            assert s[0][0] == 0 and s[-1][1] == n and all(a[1] == b[0] for a, b in zip(s, s[1:]))
            assert max(b - a for a, b in s) - min(b - a for a, b in s) <= 1
