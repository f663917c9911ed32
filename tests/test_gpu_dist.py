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


def _worker_c2_full_depth(rank, world, port, out, ckpt_dir):
    """BASELINE config C2 (1 x 1344x896 -> 7 ViT inputs, S = 1242) at FULL depth through the two-rank path, both exchange dtypes; and a
    checkpoint -> tensor-parallel shard -> device load through compat.from_pretrained(tp_rank=, tp_size=)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from leopard_amd import compat
    from leopard_amd import dist as D
    from leopard_amd.config import full_config
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.ops import Ops
    from leopard_amd.weights import EngineWeights, SynthSource
    from tests.test_gpu_parity import sample_inputs
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D.init(backend="gloo")
    ops, cfg, dtype = Ops(), full_config(), torch.float16
    src = SynthSource(cfg, ops, dev, dtype)
    eng = LeopardEngine(cfg, EngineWeights.build(cfg, src, dtype, tp_rank=rank, tp_size=world), ops=ops, device=dev)
    u8, ids, _ = sample_inputs(cfg, 1, 1344, 896)
    tiles = torch.from_numpy(u8).to(dev)
    res = {}
    for name, cdt in (("16-bit reduce-scatter", None), ("fp32 reduce-scatter", torch.float32)):
        eng.tp_chunks, eng.tp_comm_dtype = 2, cdt
        vis = D.encode_images_sharded(eng, tiles)
        res[name] = eng.prefill(ids, None, visual_tokens=vis).logits_last.float().cpu()
    eng.precision = "lo4"                                   # round 5: the lo4 hand-overs through the sequence-parallel norms and all-gathers
    eng.tp_comm_dtype = torch.float32
    res["lo4, fp32 reduce-scatter"] = eng.prefill(ids, tiles).logits_last.float().cpu()
    del eng
    torch.cuda.empty_cache()
    # checkpoint -> TP shard -> device (SURVEY.md 8 f1 "optional TP pre-sharding on load")
    m = compat.from_pretrained(ckpt_dir, torch_dtype=torch.float16, tp_rank=rank, tp_size=world).eval().to(dev)
    S = m.config.vision_config.image_size
    px = torch.from_numpy(np.random.default_rng(4).integers(0, 256, (3, S, S, 3), dtype=np.uint8)).to(dev)
    tok = torch.tensor([[7, 500, 11, 500, 500, 12, 13]]).to(dev)
    lg_tp = m.engine.prefill(tok, px).logits_last.float().cpu()
    shard_rows = m.engine.W.llm_layers[0].o_w.shape[1]
    lg_one = None
    if rank == 0:
        one = compat.from_pretrained(ckpt_dir, torch_dtype=torch.float16).eval().to(dev)
        lg_one = one.engine.prefill(tok, px).logits_last.float().cpu()
        assert one.engine.W.llm_layers[0].o_w.shape[1] == 2 * shard_rows
    torch.cuda.synchronize()
    out.put((rank, {k: v.tolist() for k, v in res.items()}, lg_tp.tolist(), None if lg_one is None else lg_one.tolist(), m.precision))
    D.barrier()


def test_tensor_parallel_c2_full_depth_vs_oracle_fixture_and_tp_checkpoint_load(tmp_path):
    """what tensor parallelism does to PARITY, measured against the committed fp32 oracle fixture of C2 at full depth
    (tests/golden/c2_full_depth.npz), for both dtypes of the reduce-scattered partial products; and f1's TP pre-sharding on load."""
    from leopard_amd.checkpoint import save_synthetic_checkpoint
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    ck = str(tmp_path / "ckpt")
    small = LeopardConfig(          # the kernel shape rules at tiny depth, with 4 q / 2 kv heads so that two ranks get whole heads
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=256, num_hidden_layers=2, num_attention_heads=16, image_size=56, patch_size=14),
        text_config=TextConfig(hidden_size=512, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                               vocab_size=512, rope_scaling=RopeScaling()),
        image_token_index=500)
    save_synthetic_checkpoint(ck, small, shard_bytes=8 << 20)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_full_depth.npz"))
    ref = torch.from_numpy(z["logits_fp32"])
    mp.set_start_method("spawn", force=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c2_full_depth, args=(r, 2, port, q, ck)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1500) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, v0, tp0, one, prec), (_, v1, tp1, _, _) = res
    assert v0 == v1 and tp0 == tp1 and prec == "fast"          # every rank holds the same logits; a 16-bit torch_dtype request = the fast schedule
    scale = ref.abs().max().item()
    errs = {k: (torch.tensor(v) - ref).abs().max().item() / scale for k, v in v0.items()}
    print("[tp2 C2 full depth fp16] max|logit diff| / max|logit| vs the fp32 oracle fixture:", {k: f"{e:.3e}" for k, e in errs.items()},
          "(one rank, fast schedule: 1.14e-3; one rank, lo4: 3.9e-4)")
    for k, v in v0.items():
        assert int(torch.tensor(v).argmax()) == int(ref.argmax()), k
    # the 16-bit exchange adds one rounding per partial product and half layer; the fp32 exchange must stay on the one-rank fast budget
    assert errs["fp32 reduce-scatter"] <= 1.6e-3 and errs["16-bit reduce-scatter"] <= 2.5e-3
    assert errs["lo4, fp32 reduce-scatter"] <= 1.0e-3          # north_star's figure on two ranks
    d = (torch.tensor(tp0) - torch.tensor(one)).abs().max().item() / torch.tensor(one).abs().max().item()
    print(f"[tp2 checkpoint load] TP-sharded from_pretrained vs one rank: {d:.3e} of the logit scale")
    assert d <= 2.5e-3


def _worker_c2_tp8(rank, world, port, out):
    """C2 at FULL depth on EIGHT ranks sharing one device (1/8 of the LLM per rank): the 8-way reduce-scatter order, the 8-way column split of the
    head, and the tile-sharded vision encode with 7 ViT inputs on 8 ranks — one rank has nothing to encode."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from leopard_amd import dist as D
    from leopard_amd.config import full_config
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.ops import Ops
    from leopard_amd.weights import EngineWeights, SynthSource
    from tests.test_gpu_parity import sample_inputs
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D.init(backend="gloo")
    ops, cfg, dtype = Ops(), full_config(), torch.float16
    eng = LeopardEngine(cfg, EngineWeights.build(cfg, SynthSource(cfg, ops, dev, dtype), dtype, tp_rank=rank, tp_size=world), ops=ops, device=dev)
    u8, ids, _ = sample_inputs(cfg, 1, 1344, 896)
    tiles = torch.from_numpy(u8).to(dev)
    assert tiles.shape[0] == 7 and world == 8
    res = {}
    eng.tp_chunks, eng.tp_comm_dtype = 2, torch.float32
    res["fast, fp32 reduce-scatter"] = eng.prefill(ids, tiles).logits_last.float().cpu()
    eng.precision = "lo4"
    res["lo4, fp32 reduce-scatter"] = eng.prefill(ids, tiles).logits_last.float().cpu()
    shard = (eng.W.llm_layers[0].o_w.shape, eng.W.lm_head.shape[0])
    torch.cuda.synchronize()
    out.put((rank, {k: v.tolist() for k, v in res.items()}, shard))
    D.barrier()


def test_tensor_parallel_c2_full_depth_eight_ranks_one_gpu():
    """8-rank readiness without an 8-GPU node — the reduce order of 8 partial products, the 8-way head and the idle vision rank
    meet the fp32 oracle fixture (tests/golden/c2_full_depth.npz) at full depth, fast and lo4 (host-staged gloo collectives: times mean nothing)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c2_full_depth.npz"))
    ref = torch.from_numpy(z["logits_fp32"])
    mp.set_start_method("spawn", force=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c2_tp8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=2400) for _ in procs)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    v0 = res[0][1]
    for r in res[1:]:
        assert r[1] == v0 and r[2] == res[0][2]                  # every rank holds the same logits and an equal shard
    assert res[0][2][0][1] * 8 == 4096                           # o_proj: 1/8 of the contraction per rank
    scale = ref.abs().max().item()
    errs = {k: (torch.tensor(v) - ref).abs().max().item() / scale for k, v in v0.items()}
    print("[tp8 on one device, C2 full depth fp16] max|logit diff| / max|logit| vs the fp32 oracle fixture:", {k: f"{e:.3e}" for k, e in errs.items()},
          "(two ranks: 1.20e-3 / 3.9e-4; one rank: 1.38e-3 / 3.8e-4)")
    for k, v in v0.items():
        assert int(torch.tensor(v).argmax()) == int(ref.argmax()), k
    assert errs["fast, fp32 reduce-scatter"] <= 1.7e-3
    assert errs["lo4, fp32 reduce-scatter"] <= 1.0e-3            # north_star's figure on eight ranks
