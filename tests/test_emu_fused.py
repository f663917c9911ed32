"""Kernel-logic tests (CPU emulator) of the RMSNorm / RoPE fusion into the GEMM epilogues: lmi_gemm_ex (producer: second
normalised output + row partial sums of squares; consumer: row scale) and lmi_rmsnorm_rope (q|k|v projection + rotate-half
RoPE on permuted weight rows + KV-cache append), every tile geometry, ragged M, against plain fp32 definitions; and the fused
LLM layer schedule of LeopardEngine against the unfused one."""
import numpy as np
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.weights import interleave_gate_up, rope_permute_rows
from tests.emu_util import emu_ops

DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    o = emu_ops()
    yield o
    o.set_option("gemm.config", -1)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def eps(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7


def test_rope_permute_rows_is_the_documented_order():
    w = torch.arange(2 * 128).float().view(256, 1)
    p = rope_permute_rows(w)
    want = list(range(0, 32)) + list(range(64, 96)) + list(range(32, 64)) + list(range(96, 128))
    assert p[:128, 0].tolist() == [float(v) for v in want]
    assert p[128:, 0].tolist() == [float(128 + v) for v in want]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 5, 8, 10])
def test_gemm_ex_producer_and_consumer(ops, dtype, cfg):
    """x += a.w^T; h = T(x * gamma), rowsq partials; then a consumer GEMM on h with the row scale == the GEMM on rmsnorm(x)."""
    M, N, K = 300, 256, 128
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, 0.1)
    x0 = rnd((M, N), torch.float32, 3)
    gamma = torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5
    ops.set_option("gemm.config", cfg)
    try:
        x = x0.clone()
        h = torch.full((M, N), float("nan"), dtype=dtype)
        sq = torch.full((M, N // 64), float("nan"))
        ops.gemm_ex(a, w, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=gamma, rowsq_out=sq)
        x_ref = x0 + a.float() @ w.float().T
        assert (x - x_ref).abs().max() <= 1e-4
        assert torch.equal(h, (x * gamma).to(dtype))                               # same fp32 values, one rounding
        sq_ref = x.pow(2).view(M, N // 64, 64).sum(-1)
        assert (sq - sq_ref).abs().max() <= 1e-4 * sq_ref.abs().max()
        # the same producer without the extra outputs gives the same x, bit for bit
        x2 = x0.clone()
        ops.gemm(a, w, x2, epilogue=_lib.EPI_RESIDUAL)
        assert torch.equal(x, x2)
        # consumer: plain store with row scale
        N2 = 384
        w2 = rnd((N2, N), dtype, 5, 0.1)
        out = torch.empty(M, N2, dtype=dtype)
        ops.gemm_ex(h, w2, out, rowsq_in=sq, norm_dim=N, norm_eps=1e-5)
        rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5)
        ref = (h.float() @ w2.float().T) * rstd
        err = ((out.float() - ref).abs() / (1 + ref.abs())).max().item()
        assert err <= 2 * eps(dtype), err
        # ... and it IS the RMSNorm: compare with the GEMM on the separately normalised operand (two roundings apart at most)
        hn = torch.empty(M, N, dtype=dtype)
        ops.rmsnorm(x, gamma, hn, 1e-5)
        out2 = torch.empty(M, N2, dtype=dtype)
        ops.gemm(hn, w2, out2)
        assert ((out.float() - out2.float()).abs() / (1 + out2.float().abs())).max().item() <= 6 * eps(dtype)
        # consumer: SwiGLU with row scale
        F = 128
        gate, up = rnd((F, N), dtype, 6, 0.1), rnd((F, N), dtype, 7, 0.1)
        sw = torch.empty(M, F, dtype=dtype)
        ops.gemm_ex(h, interleave_gate_up(gate, up), sw, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=N, norm_eps=1e-5)
        g_ref, u_ref = (h.float() @ gate.float().T) * rstd, (h.float() @ up.float().T) * rstd
        ref = torch.nn.functional.silu(g_ref) * u_ref
        assert ((sw.float() - ref).abs() / (1 + ref.abs())).max().item() <= 3 * eps(dtype)
    finally:
        ops.set_option("gemm.config", -1)


def rope_ref(x, cos, sin):
    """x [S, heads, 128] fp32; rotate-half (rotary_pos_embedding.py:197-239)."""
    h = x.shape[-1] // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    c = torch.cat([cos, cos], -1)[:, None, :]
    s = torch.cat([sin, sin], -1)[:, None, :]
    return x * c + rot * s


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 5, 8, 10])
@pytest.mark.parametrize("with_norm", [False, True])
def test_rmsnorm_rope_qkv_projection(ops, dtype, cfg, with_norm):
    S, nq, nkv, D, K = 150, 2, 1, 128, 256
    a = rnd((S, K), dtype, 10)
    wq, wk, wv = rnd((nq * D, K), dtype, 11, 0.1), rnd((nkv * D, K), dtype, 12, 0.1), rnd((nkv * D, K), dtype, 13, 0.1)
    w_nat = torch.cat([wq, wk, wv], 0)
    w_rope = torch.cat([rope_permute_rows(torch.cat([wq, wk], 0)), wv], 0).contiguous()
    pos = torch.arange(3000, 3000 + S).float()
    inv = 1.0 / (5e5 ** (torch.arange(0, D, 2).float() / D))
    cos, sin = (pos[:, None] * inv[None]).cos().contiguous(), (pos[:, None] * inv[None]).sin().contiguous()
    sq = None
    rstd = torch.ones(S, 1)
    if with_norm:
        xr = rnd((S, K), torch.float32, 14, 2.0)
        sq = xr.pow(2).view(S, K // 64, 64).sum(-1).contiguous()
        rstd = torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)
    ops.set_option("gemm.config", cfg)
    try:
        qkv = torch.full((S, (nq + 2 * nkv) * D), float("nan"), dtype=dtype)
        kc = torch.zeros(S + 7, nkv * D, dtype=dtype)
        vc = torch.zeros_like(kc)
        ops.rmsnorm_rope(a, w_rope, qkv, sq, 1e-5, cos, sin, kc, vc, 3, nq, nkv, D)
        acc = (a.float() @ w_nat.float().T) * rstd
        ref = acc.clone().view(S, nq + 2 * nkv, D)
        ref[:, :nq + nkv] = rope_ref(ref[:, :nq + nkv], cos, sin)
        ref = ref.view(S, -1)
        err = ((qkv.float() - ref).abs() / (1 + ref.abs())).max().item()
        assert err <= 2 * eps(dtype), err
        assert torch.equal(kc[3:3 + S], qkv[:, nq * D:(nq + nkv) * D]) and torch.equal(vc[3:3 + S], qkv[:, (nq + nkv) * D:])
        assert kc[:3].abs().max() == 0 and kc[3 + S:].abs().max() == 0 and vc[:3].abs().max() == 0
        # no cache
        qkv2 = torch.empty_like(qkv)
        ops.rmsnorm_rope(a, w_rope, qkv2, sq, 1e-5, cos, sin, None, None, 0, nq, nkv, D)
        assert torch.equal(qkv, qkv2)
    finally:
        ops.set_option("gemm.config", -1)


def test_fused_llm_schedule_matches_unfused():
    """LeopardEngine.llm_prefill with fuse_norm_rope on / off on a micro model (head_dim 128, GQA 2:1): same logits up to the
    moved rounding points, same KV cache up to one rounding, and both within tolerance of the fp32 oracle."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.weights import EngineWeights, SynthSource
    from oracle import leopard_oracle as O
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    assert W.llm_layers[0].qkv_w_rope is not None
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    tiles = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17, 33, 101, 7]])
    res = {}
    for fused in (True, False):
        eng.fuse_norm_rope = fused
        cache = KVCache(cfg, 24, dtype, "cpu")
        r = eng.prefill(ids, tiles, cache=cache, all_logits=True)
        nxt = eng.decode_step(int(r.logits_last.argmax()), cache).clone()
        res[fused] = (r, cache, nxt)
    (rf, cf, nf), (ru, cu, nu) = res[True], res[False]
    scale = ru.logits_all.abs().max().item()
    assert (rf.logits_all - ru.logits_all).abs().max().item() <= 4e-3 * scale
    for i in range(3):
        assert (cf.k[i].float() - cu.k[i].float()).abs().max().item() <= 4e-3 * cu.k[i].float().abs().max().item()
        assert (cf.v[i].float() - cu.v[i].float()).abs().max().item() <= 4e-3 * cu.v[i].float().abs().max().item()
    assert (nf - nu).abs().max().item() <= 4e-3 * scale
    from leopard_amd.tiler import siglip_normalize
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    ref = O.prefill_logits(ids, torch.from_numpy(siglip_normalize(tiles.numpy())), Wt, cfg)[0]
    assert (rf.logits_all - ref).abs().max().item() <= 4e-3 * scale


def test_fused_schedule_is_position_independent():
    """Packing two sequences into one varlen launch must give each of them exactly the bits of its own launch (the GPU tests
    assert this at the C3 / C5 sizes): the folded norm's partial sums and row scales may not depend on where a row sits in
    the packed batch or on the tile geometry chosen for the batch size."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    g = torch.Generator().manual_seed(3)
    xa, xb = torch.randn(37, 256, generator=g), torch.randn(11, 256, generator=g)
    for cfg_id in (-1, 0, 5):
        ops.set_option("gemm.config", cfg_id)
        la, _ = eng.llm_prefill(xa.clone(), [37])
        lb, _ = eng.llm_prefill(xb.clone(), [11])
        lab, _ = eng.llm_prefill(torch.cat([xa, xb]), [37, 11])
        lba, _ = eng.llm_prefill(torch.cat([xb, xa]), [11, 37])
        assert torch.equal(lab[0], la[0]) and torch.equal(lab[1], lb[0])
        assert torch.equal(lba[0], lb[0]) and torch.equal(lba[1], la[0])
        if cfg_id == -1:
            base = la
        else:
            assert torch.equal(la, base)                 # ... nor on the tile geometry
    ops.set_option("gemm.config", -1)


def test_generate_batch_equals_per_sample_generate():
    """several samples through ONE packed prefill (pooled KV cache, split per sample afterwards) and the captured
    decode step give exactly the ids of per-sample generate()."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", torch.float16), torch.float16)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    rng = np.random.default_rng(9)
    samples = [(torch.tensor([[5, 250, 9, 250, 17]]), torch.from_numpy(rng.integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))),
               (torch.tensor([[8, 3, 250, 44, 45, 46, 47]]), torch.from_numpy(rng.integers(0, 256, (1, 28, 28, 3), dtype=np.uint8))),
               (torch.tensor([[11, 12, 13]]), None)]
    batch = eng.generate_batch(samples, max_new_tokens=4, eos_token_id=())
    for (ids, tiles), got in zip(samples, batch):
        one = eng.generate(ids, tiles, max_new_tokens=4, eos_token_id=())
        assert torch.equal(one, got)


@pytest.mark.parametrize("cfg", [-1, 0, 5, 10])
def test_gemm_ex_consumer_at_hidden_4096(ops, cfg):
    """The row-scale prologue's batched-load path (64 partials per row = hidden size 4096), both threads-per-row geometries."""
    dtype = torch.float16
    M, N, K = 70, 128, 4096
    a, w = rnd((M, K), dtype, 21, 0.5), rnd((N, K), dtype, 22, 0.05)
    sq = (torch.rand(M, K // 64, generator=torch.Generator().manual_seed(23)) + 0.5) * 64
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.empty(M, N, dtype=dtype)
        ops.gemm_ex(a, w, out, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
        rstd = torch.rsqrt(sq.sum(-1, keepdim=True) / K + 1e-5)
        ref = (a.float() @ w.float().T) * rstd
        assert ((out.float() - ref).abs() / (1 + ref.abs())).max().item() <= 2 * eps(dtype)
    finally:
        ops.set_option("gemm.config", -1)
