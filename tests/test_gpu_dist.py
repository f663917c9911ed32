"""Multi-rank paths on ONE MI355X: two processes share cuda:0 and talk over gloo (RCCL needs one device per rank), which is
enough to check through the real kernels that the tile-sharded vision encode + all-gather and the sequence-parallel
tensor-parallel LLM (leopard_amd.dist, LeopardEngine._llm_layers_tp) reproduce the single-rank results; and the
RCCL entry points of the C ABI (lmi_comm_init / lmi_allgather / lmi_reduce_scatter / lmi_allreduce / lmi_broadcast) on a
one-rank communicator — all a 1-GPU box can run of RCCL."""
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
    cache = KVCache(cfg, eng.tp_padded_len(S) + 8, dtype, dev, tp_size=world)
    vis = D.encode_images_sharded(eng, tiles)               # default exchange dtype (fp32): bit-identical visual tokens, asserted below
    res = eng.prefill(ids, None, cache=cache, visual_tokens=vis)
    assert res.seq_len == S and type(eng.comm).__name__ == "TorchComm" and eng.comm.sent_bytes > 0
    step = eng.decode_step(int(res.logits_last.argmax()), cache).clone()
    # the chunk pipeline and both exchange dtypes: row chunks per layer 1 / 2 / 4 (chunk c's collectives under chunk
    # c + 1's GEMMs), partial products reduce-scattered in the compute type or as exact fp32 sums, the visual tokens gathered in 16 bits
    variants = {}
    for chunks, cdt, vdt in [(1, None, None), (4, None, None), (2, torch.float32, None), (4, torch.float32, dtype), (1, torch.float32, dtype)]:
        eng.tp_chunks, eng.tp_comm_dtype, eng.tp_vision_gather_dtype = chunks, cdt, vdt
        v = D.encode_images_sharded(eng, tiles)
        variants[(chunks, str(cdt), str(vdt))] = eng.prefill(ids, None, visual_tokens=v).logits_last.float().cpu()
    eng.tp_chunks, eng.tp_comm_dtype, eng.tp_vision_gather_dtype = 2, None, None
    ref = None
    if rank == 0:
        one = LeopardEngine(cfg, EngineWeights.build(cfg, src, dtype), ops=ops, device=dev)
        c1 = KVCache(cfg, S + 8, dtype, dev)
        r1 = one.prefill(ids, tiles, cache=c1)
        s1 = one.decode_step(int(r1.logits_last.argmax()), c1)
        scale = float(r1.logits_last.abs().max())
        ref = (bool(torch.equal(vis, one.encode_images(tiles))), float((res.logits_last - r1.logits_last).abs().max()) / scale,
               float((step - s1).abs().max()) / scale, int(res.logits_last.argmax()) == int(r1.logits_last.argmax()),
               {k: float((v - r1.logits_last.float().cpu()).abs().max()) / scale for k, v in variants.items()})
    torch.cuda.synchronize()
    out.put((rank, res.logits_last.cpu().tolist(), ref, {k: v.tolist() for k, v in variants.items()}))
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
    (_, l0, ref, v0), (_, l1, _, v1) = res
    assert l0 == l1 and v0 == v1                       # every rank holds the same logits, under every variant
    vis_equal, d_prefill, d_decode, same_argmax, d_variants = ref
    assert vis_equal                                   # tile-sharded encode + all-gather (default fp32 exchange): bit-identical
    assert d_prefill <= 2.5e-3 and d_decode <= 2.5e-3 and same_argmax
    print("[tp variants] (chunks, exchange dtype, visual gather dtype) -> max|logit diff| / max|logit| vs one rank:", d_variants)
    assert len(d_variants) == 5 and all(d <= 2.5e-3 for d in d_variants.values())


def _rccl_worker(out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    from leopard_amd import dist as D
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    comm = D.RcclComm(device=dev)
    side = torch.cuda.Stream(device=dev)
    res = {"ranks": comm.ranks_seen(), "backend": comm.backend}
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        x = torch.randn(4096, 64, device=dev).to(dt)
        ag, rs, ar = torch.empty_like(x), torch.empty_like(x), x.clone()
        side.wait_stream(torch.cuda.current_stream(dev))
        comm.all_gather(ag, x, side)                      # one rank: every collective is the identity
        comm.reduce_scatter(rs, x, side)
        comm.all_reduce(ar, side)
        comm.broadcast(ar, 0, side)
        side.synchronize()
        res[str(dt)] = bool(torch.equal(ag, x) and torch.equal(rs, x) and torch.equal(ar, x))
    # graph capture of the C-ABI collectives (what the tensor-parallel decode step relies on): capture all-reduce + all-gather between two
    # elementwise kernels, replay twice, compare with the eager sequence
    x = torch.randn(1, 4096, device=dev)
    y, g_out = x.clone(), torch.empty(1, 4096, device=dev)
    cap = torch.cuda.Stream(device=dev)
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cap):
        comm.all_reduce(y, cap)                           # warm-up outside capture (RCCL channel setup)
    torch.cuda.current_stream(dev).wait_stream(cap)
    y.copy_(x)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s_ = torch.cuda.current_stream(dev)
        y.mul_(2.0)
        comm.all_reduce(y, s_)
        comm.all_gather(g_out, y, s_)
        g_out.add_(1.0)
    y.copy_(x)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    res["graph"] = bool(torch.equal(y, x * 4.0) and torch.equal(g_out, x * 4.0 + 1.0))
    res["sent"] = comm.sent_bytes                     # one rank: nothing goes on a link
    comm.destroy()
    dist.destroy_process_group()
    out.put(res)


def test_rccl_c_abi_single_rank_communicator():
    """lmi_comm_unique_id / lmi_comm_init / collectives / lmi_comm_destroy through dlopen'ed librccl on the device, on a side
    stream, in all three exchange dtypes (a child process: the communicator binds the device)."""
    mp.set_start_method("spawn", force=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["ranks"] == 1 and res["backend"].startswith("rccl")
    assert res["graph"], "lmi_allreduce / lmi_allgather inside a captured HIP graph did not replay correctly"
    assert res["torch.float16"] and res["torch.bfloat16"] and res["torch.float32"] and res["sent"] == 0
