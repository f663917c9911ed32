"""Kernel-logic tests (CPU emulator) of the fp8 path (BASELINE config 5): lmi_quantize_fp8 against PyTorch's float8_e4m3fn
conversion bit for bit, and lmi_gemm_fp8 (v_mfma_scale_f32_32x32x64_f8f6f4 operand layout, 2 k-steps per k-tile, E8M0 output
scale) on every production geometry and epilogue against fp32 matmuls of the dequantised operands (products of fp8 values are
exact in fp32, so only the summation order differs)."""
import numpy as np
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.weights import interleave_gate_up
from tests.emu_util import emu_ops

F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def ops():
    o = emu_ops()
    yield o
    o.set_option("gemm.config", -1)


def q8(x, scale=1.0):
    """PyTorch's own fp8 conversion of x * scale (saturating), as uint8 bytes."""
    return (x.float() * scale).clamp(-448, 448).to(F8).view(torch.uint8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_quantize_fp8_matches_torch_bit_for_bit(ops, dtype):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(37, 64, generator=g) * torch.logspace(-4, 3, 64)[None, :]              # subnormals ... saturation
    x[0, :8] = torch.tensor([0.0, -0.0, 448.0, -448.0, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10])
    x[1, :6] = torch.tensor([0.0546875, 0.05078125, 0.017578125, 464.0, 1.0625, 1.1875])    # rounding ties
    x = x.to(dtype)
    for scale in (1.0, 16.0, 0.25):
        out = torch.zeros(37, 64, dtype=torch.uint8)
        ops.quantize_fp8(x, out, scale)
        want = q8(x, scale)
        same = (out == want) | ((out & 0x7F) == 0) & ((want & 0x7F) == 0)                   # +0 / -0 after underflow
        assert bool(same.all()), (out[~same][:8], want[~same][:8])


@pytest.mark.parametrize("cfg", [-1, 0, 2, 5, 8])
def test_gemm_fp8_every_geometry_and_epilogue(ops, cfg):
    M, N, K = 300, 256, 256
    g = torch.Generator().manual_seed(2)
    a8, w8 = q8(torch.randn(M, K, generator=g)), q8(torch.randn(N, K, generator=g) * 0.5)
    ref = a8.view(F8).float() @ w8.view(F8).float().T
    bias = torch.randn(N, generator=g)
    ops.set_option("gemm.config", cfg)
    try:
        for dtype in (torch.float16, torch.bfloat16):
            eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            for e in (0, -3):
                out = torch.full((M, N), float("nan"), dtype=dtype)
                ops.gemm_fp8(a8, w8, out, bias=bias, scale_exp=e)
                want = ref * 2.0 ** e + bias
                assert ((out.float() - want).abs() / (1 + want.abs())).max().item() <= 2 * eps
            out = torch.empty(M, N, dtype=dtype)
            ops.gemm_fp8(a8, w8, out, bias=bias, act=_lib.ACT_GELU_TANH, scale_exp=-4)
            want = torch.nn.functional.gelu(ref * 2.0 ** -4 + bias, approximate="tanh")
            assert ((out.float() - want).abs() / (1 + want.abs())).max().item() <= 3 * eps
            x0 = torch.randn(M, N, generator=g)
            x = x0.clone()
            ops.gemm_fp8(a8, w8, x, epilogue=_lib.EPI_RESIDUAL, scale_exp=-2)
            assert (x - (x0 + ref * 0.25)).abs().max().item() <= 1e-4 * ref.abs().max().item()
            o32 = torch.empty(M, N)
            ops.gemm_fp8(a8, w8, o32, bias=bias, epilogue=_lib.EPI_STORE_F32, scale_exp=-2)
            assert (o32 - (ref * 0.25 + bias)).abs().max().item() <= 1e-4 * ref.abs().max().item()
            F = N // 2
            sw = torch.empty(M, F, dtype=dtype)
            ops.gemm_fp8(a8, interleave_gate_up(w8[:F], w8[F:]), sw, epilogue=_lib.EPI_SWIGLU, scale_exp=-3)
            want = torch.nn.functional.silu(ref[:, :F] / 8) * (ref[:, F:] / 8)
            assert ((sw.float() - want).abs() / (1 + want.abs())).max().item() <= 3 * eps
    finally:
        ops.set_option("gemm.config", -1)


def test_gemm_fp8_transpose_detecting_and_k_order(ops):
    """One-hot A rows pick single W columns: catches an operand-layout or k-step ordering error exactly."""
    M, N, K = 64, 128, 384
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 7 + 3) % K] = 1.0
    w = ((torch.arange(N * K).reshape(N, K) % 13).float() - 6) / 4
    out = torch.empty(M, N, dtype=torch.float16)
    ops.gemm_fp8(q8(a), q8(w), out)
    assert torch.equal(out.float(), a @ q8(w).view(F8).float().T)


def test_gemm_fp8_rejects_bad_shapes(ops):
    a8, w8 = torch.zeros(8, 192, dtype=torch.uint8), torch.zeros(128, 192, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="K % 128"):
        ops.gemm_fp8(a8, w8, torch.empty(8, 128, dtype=torch.float16))


def _fp8_close(out_u8, want_scaled):
    """fp8 outputs against fp8(want): identical bytes except where the fp32 value sits within summation noise of a rounding tie
    (then one e4m3 ulp apart)."""
    want8 = q8(want_scaled)
    got, ref = out_u8.view(F8).float(), want8.view(F8).float()
    assert float((out_u8 == want8).float().mean()) >= 0.99
    assert bool(((got - ref).abs() <= 0.126 * ref.abs() + 2.0 ** -9).all())


@pytest.mark.parametrize("cfg", [-1, 0, 5])
def test_gemm_fp8_fp8_outputs_gelu_and_swiglu(ops, cfg):
    """SigLIP fc1 -> fc2 and Llama gate/up -> down hand-overs: the epilogue writes fp8(result * out_scale) directly."""
    M, N, K = 200, 256, 256
    g = torch.Generator().manual_seed(5)
    a8, w8 = q8(torch.randn(M, K, generator=g)), q8(torch.randn(N, K, generator=g) * 0.5)
    ref = a8.view(F8).float() @ w8.view(F8).float().T
    bias = torch.randn(N, generator=g)
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.zeros(M, N, dtype=torch.uint8)
        ops.gemm_fp8(a8, w8, out, bias=bias, act=_lib.ACT_GELU_TANH, scale_exp=-4, out_scale=8.0)
        _fp8_close(out, 8.0 * torch.nn.functional.gelu(ref * 2.0 ** -4 + bias, approximate="tanh"))
        ops.gemm_fp8(a8, w8, out, bias=bias, scale_exp=-4, out_scale=4.0)
        _fp8_close(out, 4.0 * (ref * 2.0 ** -4 + bias))
        F = N // 2
        sw = torch.zeros(M, F, dtype=torch.uint8)
        ops.gemm_fp8(a8, interleave_gate_up(w8[:F], w8[F:]), sw, epilogue=_lib.EPI_SWIGLU, scale_exp=-3, out_scale=16.0)
        _fp8_close(sw, 16.0 * torch.nn.functional.silu(ref[:, :F] / 8) * (ref[:, F:] / 8))
        with pytest.raises(RuntimeError, match="fp8 output"):
            ops.gemm_fp8(a8, w8, out, epilogue=_lib.EPI_RESIDUAL)
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("D", [256, 1152, 4096])
def test_norm_fp8_layernorm_and_rmsnorm(ops, D):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(21, D, generator=g) * 3 + 0.5
    w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    out = torch.zeros(21, D, dtype=torch.uint8)
    ops.norm_fp8(x, w, b, out, 1e-6, 16.0)
    _fp8_close(out, 16.0 * torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6))
    ops.norm_fp8(x, w, None, out, 1e-5, 8.0)
    _fp8_close(out, 8.0 * (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * w))


# ---- the engine's fp8 schedule (leopard_amd.fp8) end to end on the emulator --------------------------------------------------------
def test_engine_fp8_schedule_matches_fp8_emulating_oracle(ops):
    """enable_fp8(): calibration, weight quantisation, the launch sequence (lmi_norm_fp8 -> lmi_gemm_fp8 -> ... with fp8-output
    GELU / SwiGLU epilogues).  The fp8 path must sit where the oracle that rounds the same operands to e4m3 predicts: its error
    against the fp32 oracle equals the predicted budget, and it is much closer to the emulating oracle than to the fp32 one is
    not required (the emulation is a statistical twin) — but both errors must have the same size."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from leopard_amd.weights import EngineWeights, SynthSource
    from oracle import leopard_oracle as O
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=2, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    u8 = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (3, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 250, 17, 33]])
    calib = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    base = eng.prefill(ids, u8, all_logits=True).logits_all.clone()
    plan = eng.enable_fp8([(torch.tensor([[3, 250, 250, 8]]), calib)])
    assert len(plan.vit) == 2 and len(plan.llm) == 2 and set(plan.llm[0].lin) == {"qkv", "qkv_rope", "o", "gu", "down"}
    assert all(l.w8.dtype == torch.uint8 for lay in plan.vit + plan.llm for l in lay.lin.values())
    got = eng.prefill(ids, u8, all_logits=True).logits_all
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    exact = O.prefill_logits(ids, pix, Wt, cfg)[0]
    with O.emulate_rounding(dtype, operand_dtype=torch.float8_e4m3fn):
        emu = O.prefill_logits(ids, pix, Wt, cfg)[0]
    scale = exact.abs().max().item()
    predicted = (emu - exact).abs().max().item() / scale
    measured = (got - exact).abs().max().item() / scale
    e16 = (base - exact).abs().max().item() / scale
    assert measured > 3 * e16                              # the fp8 schedule really ran (an fp16 path would be far more accurate)
    assert 0.5 * predicted <= measured <= 2.0 * predicted, (predicted, measured)
    assert (got - emu).abs().max().item() / scale <= 2.0 * predicted
    # fused hand-overs (default: the attention kernel writes the fp8 o_proj operand, the q|k|v GEMM rotates + appends K / V) vs the
    # separate conversion / RoPE launches: the same values up to ONE rounding order (fp32 -> e4m3 directly instead of via the 16-bit type)
    calls = []
    for name in ("quantize_fp8", "rope_qk", "attention_fp8out", "rope_qkv_fp8"):
        fn = getattr(ops, name)
        setattr(ops, name, (lambda f, n: (lambda *a, **k: (calls.append(n), f(*a, **k))[1]))(fn, name))
    try:
        cache = __import__("leopard_amd.engine", fromlist=["KVCache"]).KVCache(cfg, 64, dtype, "cpu")
        eng.prefill(ids, u8, cache=cache)
        assert "attention_fp8out" in calls and "rope_qkv_fp8" in calls and "quantize_fp8" not in calls and "rope_qk" not in calls
        k_fused = cache.k[1][:cache.length].clone()
        eng.fp8_fused = False
        calls.clear()
        cache.length = 0
        unf = eng.prefill(ids, u8, cache=cache, all_logits=True).logits_all
        assert "quantize_fp8" in calls and "rope_qk" in calls and "attention_fp8out" not in calls
        assert (unf - got).abs().max().item() / scale <= predicted           # two fp8 roundings apart at most
        assert (cache.k[1][:cache.length].float() - k_fused.float()).abs().max().item() <= 0.25 * k_fused.float().abs().max().item()
    finally:
        eng.fp8_fused = True
        for name in ("quantize_fp8", "rope_qk", "attention_fp8out", "rope_qkv_fp8"):
            delattr(ops, name)
    eng.fp8 = None                                         # reverts to the 16-bit schedule bit for bit
    assert torch.equal(eng.prefill(ids, u8, all_logits=True).logits_all, base)


def _e4m3(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("causal,lens,dtype", [(True, [150, 64, 1], torch.float16), (False, [70], torch.bfloat16), (True, [257], torch.float16)])
def test_attention_fp8_operands_and_products(ops, causal, lens, dtype):
    """lmi_attn_prep_fp8 + lmi_attn_fp8_fwd (attention_fp8.h): QK^T and PV on the fp8 matrix pipe.  (a) the prepared operands are exactly
    e4m3(x * scale) in the documented places — q8 rows, the swizzled K image, the transposed V image in key-slot order, zeros past a
    sequence's end; (b) the attention equals, to the rounding of P alone, an fp32 softmax attention over those e4m3 operands — packed
    sequences of ragged lengths, GQA, causal and full; T and fp8 outputs agree."""
    H, KV, D = 4, 2, 128
    S = sum(lens)
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(S, (H + 2 * KV) * D, generator=g) * 1.5).to(dtype)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    tiles = [(L + 63) // 64 for L in lens]
    tb = torch.tensor([0] + list(np.cumsum(tiles)), dtype=torch.int32)
    NT = int(tb[-1])
    sq, sk, sv = 32.0, 16.0, 64.0
    q8 = torch.zeros(S, H * D, dtype=torch.uint8)
    k_img = torch.full((KV * NT * 8192,), 0xAA, dtype=torch.uint8)
    v_img = torch.full((KV * NT * 8192,), 0xAA, dtype=torch.uint8)
    ops.attn_prep_fp8(qkv, cu, tb, NT, H, KV, D, sq, sk, sv, q8, k_img, v_img)
    qf, kf, vf = qkv[:, :H * D].float(), qkv[:, H * D:(H + KV) * D].float(), qkv[:, (H + KV) * D:].float()
    q_ref, k_ref, v_ref = _e4m3(qf * sq), _e4m3(kf * sk), _e4m3(vf * sv)
    assert torch.equal(q8.view(torch.float8_e4m3fn).float(), q_ref.float())
    # decode the images back to [kv head][key row][d]
    slot_key = [(b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) for hi in range(2) for b in range(2) for r in range(16)]
    kd = torch.zeros(KV, NT * 64, D)
    vd = torch.zeros(KV, NT * 64, D)
    ki, vi = k_img.view(KV, NT, 8192), v_img.view(KV, NT, 8192)
    for h in range(KV):
        for t in range(NT):
            kt, vt = ki[h, t].view(torch.float8_e4m3fn).float(), vi[h, t].view(torch.float8_e4m3fn).float()
            for r in range(64):
                for c in range(8):
                    o = r * 128 + ((c ^ ((r >> 1) & 7)) << 4)
                    kd[h, t * 64 + r, c * 16:(c + 1) * 16] = kt[o:o + 16]
            for d in range(D):
                for c in range(4):
                    o = d * 64 + ((c ^ ((d >> 2) & 3)) << 4)
                    for e in range(16):
                        vd[h, t * 64 + slot_key[c * 16 + e], d] = vt[o + e]
    outs = []
    for s_i, L in enumerate(lens):
        r0, t0 = int(cu[s_i]), int(tb[s_i]) * 64
        for h in range(KV):
            assert torch.equal(kd[h, t0:t0 + L], k_ref[r0:r0 + L, h * D:(h + 1) * D].float())
            assert torch.equal(vd[h, t0:t0 + L], v_ref[r0:r0 + L, h * D:(h + 1) * D].float())
            pad = tiles[s_i] * 64 - L
            assert pad == 0 or (kd[h, t0 + L:t0 + L + pad].abs().max() == 0 and vd[h, t0 + L:t0 + L + pad].abs().max() == 0)
        # fp32 attention over the e4m3 operands
        qs = q_ref[r0:r0 + L].float().view(L, H, D).transpose(0, 1) / sq
        ks = k_ref[r0:r0 + L].float().view(L, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0) / sk
        vs = v_ref[r0:r0 + L].float().view(L, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0) / sv
        sc = (qs @ ks.transpose(-1, -2)) * D ** -0.5
        if causal:
            sc = sc.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
        outs.append((torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(L, H * D))
    ref = torch.cat(outs, 0)
    out = torch.full((S, H * D), float("nan"), dtype=dtype)
    ops.attention_fp8(q8, k_img, v_img, out, cu, tb, NT, max(lens), H, KV, D, D ** -0.5, sq, sk, sv, causal=causal)
    err = (out.float() - ref).abs().max().item()
    assert not torch.isnan(out.float()).any() and err <= 0.04 * ref.abs().max().item(), err      # P carries 3 mantissa bits; everything else is exact or fp32
    o8 = torch.zeros(S, H * D, dtype=torch.uint8)
    ops.attention_fp8(q8, k_img, v_img, o8, cu, tb, NT, max(lens), H, KV, D, D ** -0.5, sq, sk, sv, causal=causal, out_fp8_scale=64.0, dtype=dtype)
    assert (o8.view(torch.float8_e4m3fn).float() / 64.0 - out.float()).abs().max().item() <= 0.07 * out.float().abs().max().item()


def test_engine_fp8_attention_arithmetic_is_an_option_of_the_fp8_schedule(ops):
    """engine.fp8_attention: calibration records the rotated q / k / v of every Llama layer, the schedule then runs lmi_attn_prep_fp8 +
    lmi_attn_fp8_fwd instead of the 16-bit attention (packed samples of different lengths, the KV cache still receives the 16-bit K / V), and
    the logits stay within the fp8 schedule's own error of the fp8-linears-only result."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.weights import EngineWeights, SynthSource
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16, image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                               vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    eng = LeopardEngine(cfg, EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype), ops=ops, device="cpu")
    u8 = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17, 33] + list(range(40, 110))])             # 84 rows: two key tiles, the second ragged
    base = eng.prefill(ids, u8).logits_last.clone()
    plan = eng.enable_fp8([(ids, u8)])
    assert all(k in lay.act for lay in plan.llm for k in ("q", "k", "v"))
    lin_only = eng.prefill(ids, u8).logits_last.clone()
    calls, orig = [], {}
    for name in ("attn_prep_fp8", "attention_fp8", "attention_fp8out"):
        orig[name] = getattr(ops, name)
        setattr(ops, name, (lambda f, n: (lambda *a, **k: (calls.append(n), f(*a, **k))[1]))(orig[name], name))
    try:
        eng.fp8_attention = True
        cache = KVCache(cfg, 128, dtype, "cpu")
        got = eng.prefill(ids, u8, cache=cache).logits_last
        # two Llama layers on the fp8 attention; the one SigLIP layer keeps the 16-bit arithmetic with the fp8 output
        assert calls.count("attn_prep_fp8") == 2 and calls.count("attention_fp8") == 2 and calls.count("attention_fp8out") == 1
        eng.fp8_attention = False
        cache2 = KVCache(cfg, 128, dtype, "cpu")
        eng.prefill(ids, u8, cache=cache2)
        assert torch.equal(cache.k[0][:cache.length], cache2.k[0][:cache2.length])       # layer 0's K rows do not depend on the attention arithmetic
        eng.fp8_attention = True
        two, _ = eng.prefill_batch([(ids, u8), (torch.tensor([[7, 8, 9]]), None)])        # packed: per-sequence tile bases
        assert (two[0] - got).abs().max() == 0
    finally:
        eng.fp8_attention = False
        for name, fn in orig.items():
            setattr(ops, name, fn)
    scale = base.abs().max().item()
    e_lin = (lin_only - base).abs().max().item() / scale
    e_a8 = (got - base).abs().max().item() / scale
    assert e_a8 <= 3.0 * e_lin + 1e-2, (e_lin, e_a8)
    # and it sits where the oracle that also rounds q / k / v / P to e4m3 predicts (emulate_rounding(..., fp8_attention=True))
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from oracle import leopard_oracle as O
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))
    exact = O.prefill_logits(ids, pix, Wt, cfg)[0, -1]
    with O.emulate_rounding(dtype, operand_dtype=torch.float8_e4m3fn, fp8_attention=True):
        emu = O.prefill_logits(ids, pix, Wt, cfg)[0, -1]
    with O.emulate_rounding(dtype, operand_dtype=torch.float8_e4m3fn):
        emu_lin = O.prefill_logits(ids, pix, Wt, cfg)[0, -1]
    rr = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    predicted, measured = rr(emu, exact), rr(got, exact)
    assert not torch.equal(emu, emu_lin)                        # the option changes the emulation
    assert 0.4 * predicted <= measured <= 2.5 * predicted, (predicted, measured)


def test_attention_fp8_entries_reject_bad_arguments(ops):
    """lmi_attn_prep_fp8 / lmi_attn_fp8_fwd fail loudly (LMI_EINVAL + message) instead of launching on a head size, stride or scale they cannot serve;
    empty work is a no-op."""
    H, KV, D, S = 2, 1, 128, 70
    qkv = torch.zeros(S, (H + 2 * KV) * D, dtype=torch.float16)
    cu = torch.tensor([0, S], dtype=torch.int32)
    tb = torch.tensor([0, 2], dtype=torch.int32)
    q8 = torch.zeros(S, H * D, dtype=torch.uint8)
    k_img, v_img = torch.zeros(KV * 2 * 8192, dtype=torch.uint8), torch.zeros(KV * 2 * 8192, dtype=torch.uint8)
    out = torch.zeros(S, H * D, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="head_dim"):
        ops.attn_prep_fp8(qkv, cu, tb, 2, H, KV, 64, 1.0, 1.0, 1.0, q8, k_img, v_img)
    with pytest.raises(RuntimeError, match="scales"):
        ops.attn_prep_fp8(qkv, cu, tb, 2, H, KV, D, 0.0, 1.0, 1.0, q8, k_img, v_img)
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.attn_prep_fp8(qkv, cu, tb, 2, H, KV, D, 1.0, 1.0, 1.0, torch.zeros(S, (H - 1) * D, dtype=torch.uint8), k_img, v_img)   # q8 rows narrower than the heads
    with pytest.raises(RuntimeError, match="head_dim"):
        ops.attention_fp8(q8, k_img, v_img, out, cu, tb, 2, S, H, KV, 96, 0.1, 1.0, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="bad argument"):
        ops.attention_fp8(q8, k_img, v_img, out, cu, tb, 2, S, 3, 2, D, 0.1, 1.0, 1.0, 1.0)               # 3 query heads over 2 kv heads
    ops.attn_prep_fp8(qkv, cu, tb, 0, H, KV, D, 1.0, 1.0, 1.0, q8, k_img, v_img)                            # no tiles: nothing to do
    ops.attention_fp8(q8, k_img, v_img, out, cu, tb, 0, S, H, KV, D, 0.1, 1.0, 1.0, 1.0)
