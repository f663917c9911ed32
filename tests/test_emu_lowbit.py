"""Kernel-logic tests (CPU emulator) of the LOW-BIT CORRECTION PHASE (csrc/lowbit.h, the LO4 instantiations of csrc/gemm.h):
the MX fp4 images of the rounding residuals — encoding rule == the oracle's emulation (oracle.leopard_oracle._lo_round), bit for
bit —, the weight image, the norm / split producers, and the GEMM that multiplies the images into the accumulators of its 16-bit
pass, on every tile geometry and epilogue the engine uses, ragged M, K not a multiple of 256 (SigLIP's 1152), against plain fp32
definitions over the DEQUANTISED images.  What the fp4 MFMA does with those bytes on the hardware is pinned on the device
(tools/ubench/mfma_fp4_layout.hip, tests/test_gpu_lowbit.py)."""
import numpy as np
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.ops import Lo4Act, Lo4Weight, lo4_k4
from leopard_amd.weights import interleave_gate_up, rope_permute_rows
from oracle import leopard_oracle as O
from tests.emu_util import emu_ops

DTYPES = [torch.float16, torch.bfloat16]
E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


@pytest.fixture(scope="module")
def ops():
    o = emu_ops()
    yield o
    o.set_option("gemm.config", -1)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def decode_img(img: torch.Tensor, sc: torch.Tensor, K: int, per_row: bool = False) -> torch.Tensor:
    """fp4 image [M, K4 / 2] bytes + E8M0 scales ([M, K4 / 32] or [M] with per_row) -> fp32 [M, K] (independent of the kernels)."""
    M, half = img.shape
    b = img.to(torch.int64)
    codes = torch.stack([b & 15, b >> 4], dim=-1).reshape(M, half * 2)           # element k in nibble k & 1 of byte k >> 1
    val = E2M1[codes & 7] * torch.where((codes & 8) != 0, -1.0, 1.0)
    s = torch.exp2(sc.to(torch.float32) - 127.0)
    s = s[:, None].expand(M, half * 2) if per_row else s[:, :half * 2 // 32].repeat_interleave(32, dim=1)
    return (val * s)[:, :K]


def eps(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7


@pytest.mark.parametrize("dtype", DTYPES)
def test_split_lo4_is_the_oracle_rule_bit_for_bit(ops, dtype):
    M, K = 7, 1152                                                       # K4 = 1280: 128 padding elements
    x = rnd((M, K), torch.float32, 1, 3.0)
    x[2, 64:96] = 0.0                                                    # an all-zero block
    x[3, 100] = 1e4                                                      # an outlier: its block's other residuals flush to zero
    act = Lo4Act.empty(M, K, dtype, "cpu")
    act.img.fill_(0xAB); act.sc.fill_(0xCD)
    ops.split_lo4(x, act)
    hi = x.to(dtype)
    assert torch.equal(act.hi, hi)
    lo = x - hi.float()
    want = O._lo_round(lo, "e2m1", 32)
    got = decode_img(act.img, act.sc, K)
    assert torch.equal(got, want)
    assert act.K4 == 1280 and act.img[:, K // 2:].abs().max() == 0 and act.sc[:, K // 32:].abs().max() == 0
    # what the image buys: the residual of the residual is ~0.15 of it (fp4: one mantissa bit)
    assert ((lo - got).pow(2).mean().sqrt() / lo.pow(2).mean().sqrt()).item() < 0.2


@pytest.mark.parametrize("dtype", DTYPES)
def test_quantize_w4_is_the_oracle_rule_bit_for_bit(ops, dtype):
    N, K = 9, 320                                                        # K4 = 512
    w = rnd((N, K), dtype, 2, 0.05)
    w[4] = 0
    w4 = ops.quantize_w4(w)
    assert w4.img.shape == (N, 256) and w4.sc.shape == (N,)
    got = decode_img(w4.img, w4.sc, K, per_row=True)
    assert torch.equal(got, O._lo_round(w.float(), "e2m1", 0))
    assert w4.img[:, K // 2:].abs().max() == 0
    assert ((w.float() - got).pow(2).mean().sqrt() / w.float().pow(2).mean().sqrt()).item() < 0.2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rms,D", [(False, 1152), (True, 4096), (True, 256)])
def test_norm_lo4_hands_over_the_same_16_bit_rows_plus_their_residual(ops, dtype, rms, D):
    M = 37 if D < 4096 else 35
    x = rnd((M, D), torch.float32, 3, 2.0) + 0.3
    w = torch.rand(D, generator=torch.Generator().manual_seed(4)) + 0.5
    b = None if rms else rnd((D,), torch.float32, 5, 0.2)
    act = Lo4Act.empty(M, D, dtype, "cpu")
    act.img.fill_(0xAB); act.sc.fill_(0xCD)
    ops.norm_lo4(x, w, b, act, 1e-5)
    plain = torch.empty(M, D, dtype=dtype)
    (ops.rmsnorm(x, w, plain, 1e-5) if rms else ops.layernorm(x, w, b, plain, 1e-5))
    assert torch.equal(act.hi, plain)                                     # the 16-bit operand is the one the fast schedule hands over
    y = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w) if rms else torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    lo = decode_img(act.img, act.sc, D)
    before = (y - act.hi.float()).pow(2).mean().sqrt().item()
    after = (y - act.hi.float() - lo).pow(2).mean().sqrt().item()
    assert after < 0.25 * before, (before, after)
    k4 = lo4_k4(D)
    if k4 > D:
        assert act.img[:, D // 2:].abs().max() == 0 and act.sc[:, D // 32:k4 // 32].abs().max() == 0


def _operands(M, N, K, dtype, seed):
    """A fp32 'true' activation, its Lo4Act through the split kernel's rule (host side, oracle quantiser), a weight and its image."""
    x = rnd((M, K), torch.float32, seed, 2.0)
    w = rnd((N, K), dtype, seed + 1, 0.1)
    return x, w


def _act_from(ops, x, dtype):
    act = Lo4Act.empty(x.shape[0], x.shape[1], dtype, "cpu")
    ops.split_lo4(x.contiguous(), act)
    return act


def _ref_acc(act: Lo4Act, w, w4: Lo4Weight):
    K = act.K
    return act.hi.float() @ w.float().T + decode_img(act.img, act.sc, K) @ decode_img(w4.img, w4.sc, K, per_row=True).T


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 2, 5, 8, 10])
def test_gemm_lo4_store_gelu_and_the_residual_image_of_its_output(ops, dtype, cfg):
    """STORE + bias (+ GELU): hi.W^T + img(A).img(W)^T == the product over the dequantised images; the product is closer to the fp32
    x.W^T than the 16-bit pass alone; and the epilogue's image of ITS OWN output's residual decodes to the oracle rule's neighbourhood."""
    M, N, K = 300, 256, 320                                               # K4 = 512: 5 16-bit k-tiles + 2 fp4 k-tiles; ragged second row tile
    x, w = _operands(M, N, K, dtype, 10)
    bias = rnd((N,), torch.float32, 12)
    act, w4 = _act_from(ops, x, dtype), ops.quantize_w4(w)
    ops.set_option("gemm.config", cfg)
    try:
        ref = _ref_acc(act, w, w4) + bias
        for a_id, f in ((_lib.ACT_NONE, lambda t: t), (_lib.ACT_GELU_TANH, lambda t: torch.nn.functional.gelu(t, approximate="tanh"))):
            out4 = Lo4Act.empty(M, N, dtype, "cpu")
            out4.img.fill_(0xAB); out4.sc.fill_(0xCD)
            ops.gemm_lo4(act, w, w4, out4.hi, bias=bias, act=a_id, out4=out4)
            y = f(ref)
            assert ((out4.hi.float() - y).abs() / (1 + y.abs())).max().item() <= 2 * eps(dtype)
            lo = decode_img(out4.img, out4.sc, N)
            before = (y - out4.hi.float()).pow(2).mean().sqrt().item()
            after = (y - out4.hi.float() - lo).pow(2).mean().sqrt().item()
            assert after < 0.3 * before, (before, after)
        # the correction does what it is for: against the UNROUNDED activation the corrected product is several times closer
        exact = x @ w.float().T + bias
        plain = torch.empty(M, N, dtype=torch.float32)
        ops.gemm(act.hi, w, plain, bias=bias, epilogue=_lib.EPI_STORE_F32)
        corr = torch.empty(M, N, dtype=torch.float32)
        ops.gemm_lo4(act, w, w4, corr, bias=bias, epilogue=_lib.EPI_STORE_F32)
        e_plain, e_corr = (plain - exact).pow(2).mean().sqrt().item(), (corr - exact).pow(2).mean().sqrt().item()
        assert (corr - ref).abs().max() <= 1e-4 * ref.abs().max()
        assert e_corr < 0.3 * e_plain, (e_plain, e_corr)
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 2, 5, 8])
def test_gemm_lo4_residual_producer_then_swiglu_consumer(ops, dtype, cfg):
    """The Llama half layer of the lo4 schedule: o_proj (RESIDUAL producer: x += ..., h = T(x gamma) + its residual image + row partial
    sums) feeding gate/up (SwiGLU consumer with the folded row scale, writing the image of its own products for down_proj)."""
    M, N, K = 200, 256, 256
    x_att, w = _operands(M, N, K, dtype, 20)
    x0 = rnd((M, N), torch.float32, 22)
    gamma = torch.rand(N, generator=torch.Generator().manual_seed(23)) + 0.5
    act, w4 = _act_from(ops, x_att, dtype), ops.quantize_w4(w)
    ops.set_option("gemm.config", cfg)
    try:
        xs = x0.clone()
        h = Lo4Act.empty(M, N, dtype, "cpu")
        h.img.fill_(0xAB); h.sc.fill_(0xCD)
        sq = torch.full((M, N // 64), float("nan"))
        ops.gemm_lo4(act, w, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq, out4=h)
        x_ref = x0 + _ref_acc(act, w, w4)
        assert (xs - x_ref).abs().max() <= 1e-4 * x_ref.abs().max()
        assert torch.equal(h.hi, (xs * gamma).to(dtype))
        want = O._lo_round(xs * gamma - h.hi.float(), "e2m1", 32)           # same fp32 values in, the oracle rule out: bit for bit
        assert torch.equal(decode_img(h.img, h.sc, N), want)
        assert (sq - xs.pow(2).view(M, N // 64, 64).sum(-1)).abs().max() <= 1e-4 * sq.abs().max()
        # consumer
        F = 128
        gate, up = rnd((F, N), dtype, 24, 0.1), rnd((F, N), dtype, 25, 0.1)
        gu_w = interleave_gate_up(gate, up)
        gu4 = ops.quantize_w4(gu_w)
        prod = Lo4Act.empty(M, F, dtype, "cpu")
        ops.gemm_lo4(h, gu_w, gu4, prod.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=N, norm_eps=1e-5, out4=prod)
        rstd = torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-5)
        acc = _ref_acc(h, gu_w, gu4) * rstd                                 # interleaved [32 gate | 32 up] blocks
        acc = acc.view(M, F // 32, 2, 32)
        y = (torch.nn.functional.silu(acc[:, :, 0]) * acc[:, :, 1]).reshape(M, F)
        assert ((prod.hi.float() - y).abs() / (1 + y.abs())).max().item() <= 3 * eps(dtype)
        lo = decode_img(prod.img, prod.sc, F)
        before = (y - prod.hi.float()).pow(2).mean().sqrt().item()
        after = (y - prod.hi.float() - lo).pow(2).mean().sqrt().item()
        assert after < 0.3 * before, (before, after)
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 5, 8])
def test_rmsnorm_rope_lo4(ops, dtype, cfg):
    from tests.test_emu_fused import rope_ref
    S, nq, nkv, D, K = 150, 2, 1, 128, 256
    xa = rnd((S, K), torch.float32, 30, 2.0)
    wq, wk, wv = rnd((nq * D, K), dtype, 31, 0.1), rnd((nkv * D, K), dtype, 32, 0.1), rnd((nkv * D, K), dtype, 33, 0.1)
    w_nat = torch.cat([wq, wk, wv], 0)
    w_rope = torch.cat([rope_permute_rows(torch.cat([wq, wk], 0)), wv], 0).contiguous()
    act = _act_from(ops, xa, dtype)
    w4_rope, w4_nat = ops.quantize_w4(w_rope), ops.quantize_w4(w_nat)
    pos = torch.arange(3000, 3000 + S).float()
    inv = 1.0 / (5e5 ** (torch.arange(0, D, 2).float() / D))
    cos, sin = (pos[:, None] * inv[None]).cos().contiguous(), (pos[:, None] * inv[None]).sin().contiguous()
    xr = rnd((S, K), torch.float32, 34, 2.0)
    sq = xr.pow(2).view(S, K // 64, 64).sum(-1).contiguous()
    rstd = torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)
    ops.set_option("gemm.config", cfg)
    try:
        qkv = torch.full((S, (nq + 2 * nkv) * D), float("nan"), dtype=dtype)
        kc = torch.zeros(S + 7, nkv * D, dtype=dtype)
        vc = torch.zeros_like(kc)
        ops.rmsnorm_rope_lo4(act, w_rope, w4_rope, qkv, sq, 1e-5, cos, sin, kc, vc, 3, nq, nkv, D)
        acc = _ref_acc(act, w_nat, w4_nat) * rstd
        ref = acc.clone().view(S, nq + 2 * nkv, D)
        ref[:, :nq + nkv] = rope_ref(ref[:, :nq + nkv], cos, sin)
        ref = ref.view(S, -1)
        assert ((qkv.float() - ref).abs() / (1 + ref.abs())).max().item() <= 2 * eps(dtype)
        assert torch.equal(kc[3:3 + S], qkv[:, nq * D:(nq + nkv) * D]) and torch.equal(vc[3:3 + S], qkv[:, (nq + nkv) * D:])
    finally:
        ops.set_option("gemm.config", -1)


def test_gemm_lo4_k_1152_and_packed_weights(ops):
    """SigLIP's K = 1152 (K4 = 1280: the last fp4 k-tile is half padding) and the packed order of the 16-bit weight copy."""
    from leopard_amd.weights import as_packed
    dtype = torch.float16
    M, N, K = 130, 128, 1152
    x, w = _operands(M, N, K, dtype, 40)
    act, w4 = _act_from(ops, x, dtype), ops.quantize_w4(w)
    ref = _ref_acc(act, w, w4)
    out = torch.empty(M, N, dtype=torch.float32)
    ops.gemm_lo4(act, w, w4, out, epilogue=_lib.EPI_STORE_F32)
    assert (out - ref).abs().max() <= 1e-4 * ref.abs().max()
    M, N, K = 70, 256, 256
    x, w = _operands(M, N, K, dtype, 42)
    act, w4 = _act_from(ops, x, dtype), ops.quantize_w4(w)
    ref = _ref_acc(act, w, w4)
    for cfg in (5, 8):
        ops.set_option("gemm.config", cfg)
        try:
            a = torch.empty(M, N, dtype=torch.float32)
            b = torch.empty(M, N, dtype=torch.float32)
            ops.gemm_lo4(act, w, w4, a, epilogue=_lib.EPI_STORE_F32)
            ops.gemm_lo4(act, as_packed(w), w4, b, epilogue=_lib.EPI_STORE_F32)
            assert torch.equal(a, b) and (a - ref).abs().max() <= 1e-4 * ref.abs().max()
        finally:
            ops.set_option("gemm.config", -1)


def test_lo4_entry_points_reject_bad_arguments(ops):
    dtype = torch.float16
    x, w = _operands(8, 128, 256, dtype, 50)
    act, w4 = _act_from(ops, x, dtype), ops.quantize_w4(w)
    out = torch.empty(8, 128, dtype=dtype)
    with pytest.raises(RuntimeError, match="lo4"):
        ops.gemm_lo4(act, w, Lo4Weight(w4.img[:, :64].contiguous(), w4.sc), out)         # weight image narrower than k4 / 2
    x64 = rnd((8, 64), torch.float32, 51)
    with pytest.raises(RuntimeError, match="lo4|K >= 128"):
        a64 = _act_from(ops, x64, dtype)
        ops.gemm_lo4(a64, w[:, :64].contiguous(), ops.quantize_w4(w[:, :64].contiguous()), out)
    with pytest.raises(RuntimeError, match="lmi_split_lo4"):
        ops.split_lo4(rnd((4, 40), torch.float32, 52), Lo4Act(torch.empty(4, 40, dtype=dtype), torch.empty(4, 128, dtype=torch.uint8),
                                                              torch.empty(4, 8, dtype=torch.uint8)))


def test_engine_lo4_mode_lands_on_the_oracle_prediction():
    """LeopardEngine.precision = "lo4" on a micro model (head_dim 128, GQA 2:1, the fused Llama schedule): the logits sit where the oracle
    that emulates exactly this arithmetic (emulate_rounding(lo_sites = every layer-linear operand)) predicts, closer to fp32 than the fast
    schedule; K / V are still appended by the q|k|v epilogue; switching back restores the fast path bit for bit."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from leopard_amd.weights import EngineWeights, SynthSource
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    tiles = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 250, 17, 33, 101, 7]])
    pix = torch.from_numpy(siglip_normalize(tiles.numpy()))
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    ref = O.prefill_logits(ids, pix, Wt, cfg)[0]
    sites = ("norm", "attn_out", "mlp_act")
    with O.emulate_rounding(dtype, lo_sites=sites):
        emu = O.prefill_logits(ids, pix, Wt, cfg)[0]
    with O.emulate_rounding(dtype, exact_sites=sites):
        floor = O.prefill_logits(ids, pix, Wt, cfg)[0]
    base = eng.prefill(ids, tiles, all_logits=True).logits_all.clone()
    assert eng.precision == "fast" and eng.lo4_vit == "auto"         # default: the tower is corrected for short samples only; here: the LLM layers only
    eng.lo4_vit = False
    eng.precision = "lo4"
    llm_only = eng.prefill(ids, tiles, all_logits=True).logits_all.clone()
    with O.emulate_rounding(dtype, lo_sites=("llm.norm", "llm.attn_out", "llm.mlp_act")):
        emu_llm = O.prefill_logits(ids, pix, Wt, cfg)[0]
    e_l, e_lp = (llm_only - ref).abs().max().item() / ref.abs().max().item(), (emu_llm - ref).abs().max().item() / ref.abs().max().item()
    # (1 + 3 layers: the uncorrected SigLIP layer and the attention operands dominate here; full depth: tests/test_gpu_parity.py)
    assert e_l <= 1.05 * (base - ref).abs().max().item() / ref.abs().max().item() and 0.5 * e_lp <= e_l <= 2.0 * e_lp, (e_l, e_lp)
    eng.lo4_vit = True                                                # + the SigLIP layer linears (what the oracle arm `sites` emulates)
    cache = KVCache(cfg, 64, dtype, "cpu")
    res = eng.prefill(ids, tiles, cache=cache, all_logits=True)
    got = res.logits_all
    scale = ref.abs().max().item()
    e_base, e_lo4 = (base - ref).abs().max().item() / scale, (got - ref).abs().max().item() / scale
    e_pred, e_floor = (emu - ref).abs().max().item() / scale, (floor - ref).abs().max().item() / scale
    assert e_lo4 < e_base, (e_lo4, e_base)
    assert 0.6 * e_pred <= e_lo4 <= 1.5 * e_pred, (e_lo4, e_pred, e_floor)
    # (two runs with the same rounding points are as far from each other as each is from fp32 — DESIGN.md 2.1 — so got vs emu is not asserted)
    assert cache.length == res.seq_len and bool(cache.k[2][:cache.length].abs().sum() > 0)
    eng.precision = "fast"
    eng.lo4_vit = False
    assert torch.equal(eng.prefill(ids, tiles, all_logits=True).logits_all, base)
    with pytest.raises(ValueError):
        eng.precision = "fp64"


@pytest.mark.parametrize("hd,H,KV,causal", [(128, 2, 1, True), (72, 8, 8, False)])
def test_attention_lo4_image_is_the_oracle_rule_in_the_padded_head_order(ops, hd, H, KV, causal):
    """lmi_attn_varlen_fwd_lo4: the 16-bit rows are those of lmi_attn_varlen_fwd bit for bit; the image, read back through the per-head padded
    k order (head h -> blocks [h NB, (h + 1) NB)), is the oracle's MX e2m1 rule applied to (fp32 output - its 16-bit rounding) of every head
    zero-padded to NB * 32; a GEMM whose weight image is built with the same head_pad consumes it."""
    from leopard_amd.ops import lo4_head_k4
    dtype = torch.float16
    lens = [70, 45]
    S = sum(lens)
    cu = torch.tensor([0, 70, 115], dtype=torch.int32)
    qkv = rnd((S, (H + 2 * KV) * hd), dtype, 60)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + KV) * hd], qkv[:, (H + KV) * hd:]
    plain = torch.empty(S, H * hd, dtype=dtype)
    ops.attention(q, k, v, plain, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
    o32 = torch.empty(S, H * hd, dtype=torch.float32)
    ops.attention_f32out(q, k, v, o32, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
    act = Lo4Act.empty(S, H * hd, dtype, "cpu", k4=lo4_head_k4(H, hd))
    ops.attention_lo4(q, k, v, act, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
    assert torch.equal(act.hi, plain)
    nb = (hd + 31) // 32 * 32
    lo = torch.zeros(S, H, nb)
    lo[:, :, :hd] = (o32 - plain.float()).view(S, H, hd)
    want = O._lo_round(lo.view(S, H * nb), "e2m1", 32)
    got = decode_img(act.img, act.sc, H * nb)
    assert torch.equal(got, want)
    if H * nb < act.K4:
        assert act.img[:, H * nb // 2:].abs().max() == 0 and act.sc[:, H * nb // 32:].abs().max() == 0
    # the consuming projection: weight image in the same padded order
    N = 128
    w = rnd((N, H * hd), dtype, 61, 0.1)
    w4 = ops.quantize_w4(w, head_pad=(H, hd))
    out = torch.empty(S, N, dtype=torch.float32)
    ops.gemm_lo4(act, w, w4, out, epilogue=_lib.EPI_STORE_F32)
    wq = O._lo_round(w.float(), "e2m1", 0)
    ref = plain.float() @ w.float().T + want.view(S, H, nb)[:, :, :hd].reshape(S, H * hd) @ wq.T
    assert (out - ref).abs().max() <= 1e-4 * ref.abs().max()
    exact = o32 @ w.float().T
    base = plain.float() @ w.float().T
    assert (out - exact).pow(2).mean().sqrt() < 0.35 * (base - exact).pow(2).mean().sqrt()


# ---- row selection of the correction phase (round 6: GemmArgs::row_sel / unit_sel, LeopardEngine.lo4_rows) ----------------------------------
def _ranges(row):
    """[begin, end) runs of the selected rows: the host-side tile-order hint (lmi_lo4.sel_ranges)."""
    import numpy as np
    r = np.flatnonzero(np.diff(np.concatenate([[0], row.numpy().astype(np.int8), [0]])))
    return np.ascontiguousarray(r.reshape(-1, 2).astype(np.int32))


def _selection(M, rows):
    row = torch.zeros(M, dtype=torch.uint8)
    row[list(rows)] = 1
    unit = torch.zeros((M + 63) // 64 * 64, dtype=torch.uint8)
    unit[:M] = row
    return row, unit.view(-1, 64).max(dim=1).values.contiguous(), _ranges(row)


def _select(act: Lo4Act, sel):
    """The contract of a selected pass: unselected rows carry all-zero images."""
    act.row_sel, act.unit_sel, act.sel_ranges = sel
    keep = sel[0].bool()
    act.img[~keep] = 0
    act.sc[~keep] = 0
    return act


@pytest.mark.parametrize("cfg", [-1, 0, 2, 5, 8, 10])
def test_gemm_lo4_row_selection_selected_rows_are_lo4_the_others_fast_bit_for_bit(ops, cfg):
    """Rows with row_sel != 0 get exactly the corrected result, the others exactly the fast schedule's, whatever tile they share; tiles
    without a selected row skip the fp4 k-tiles; the output image is written for selected rows only (the rest of it stays as it was: zero)."""
    dtype = torch.float16
    M, N, K = 700, 256, 320
    x, w = _operands(M, N, K, dtype, 60)
    bias = rnd((N,), torch.float32, 61)
    rows = list(range(100, 131)) + [255, 256] + list(range(690, 700))          # inside a tile, across a tile edge, the ragged tail; tiles 384.. unselected
    sel = _selection(M, rows)
    full, w4 = _act_from(ops, x, dtype), ops.quantize_w4(w)
    part = _select(_act_from(ops, x, dtype), sel)
    keep = sel[0].bool()
    ops.set_option("gemm.config", cfg)
    try:
        want_lo4, want_fast, got = (torch.empty(M, N, dtype=torch.float32) for _ in range(3))
        ops.gemm_lo4(full, w, w4, want_lo4, bias=bias, epilogue=_lib.EPI_STORE_F32)
        ops.gemm(full.hi, w, want_fast, bias=bias, epilogue=_lib.EPI_STORE_F32)
        ops.gemm_lo4(part, w, w4, got, bias=bias, epilogue=_lib.EPI_STORE_F32)
        assert torch.equal(got[keep], want_lo4[keep]) and torch.equal(got[~keep], want_fast[~keep])
        assert not torch.equal(want_lo4[keep], want_fast[keep])
        # producer side: STORE + GELU with the image of its own output
        o_full, o_part = Lo4Act.empty(M, N, dtype, "cpu"), Lo4Act.empty(M, N, dtype, "cpu", sel=sel)
        ops.gemm_lo4(full, w, w4, o_full.hi, bias=bias, act=_lib.ACT_GELU_TANH, out4=o_full)
        ops.gemm_lo4(part, w, w4, o_part.hi, bias=bias, act=_lib.ACT_GELU_TANH, out4=o_part)
        assert torch.equal(o_part.hi[keep], o_full.hi[keep])
        assert torch.equal(o_part.img[keep], o_full.img[keep]) and torch.equal(o_part.sc[keep], o_full.sc[keep])
        assert o_part.img[~keep].abs().max() == 0 and o_part.sc[~keep].abs().max() == 0
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("cfg", [-1, 5, 8])
def test_lo4_row_selection_through_the_llama_half_layer(ops, cfg):
    """RESIDUAL producer -> SwiGLU consumer with a selection: the producer writes T(x gamma) for every row and the image for selected rows; the
    consumer's products equal the full-lo4 products on selected rows and the fast products on the others."""
    dtype = torch.float16
    M, N, K, F = 330, 256, 256, 128
    x_att, w = _operands(M, N, K, dtype, 70)
    x0 = rnd((M, N), torch.float32, 72)
    gamma = torch.rand(N, generator=torch.Generator().manual_seed(73)) + 0.5
    gu_w = interleave_gate_up(rnd((F, N), dtype, 74, 0.1), rnd((F, N), dtype, 75, 0.1))
    w4, gu4 = ops.quantize_w4(w), ops.quantize_w4(gu_w)
    sel = _selection(M, range(300, 330))
    keep = sel[0].bool()

    def run(mode):
        a = _act_from(ops, x_att, dtype)
        s = sel if mode == "sel" else None
        if s is not None:
            _select(a, s)
        xs = x0.clone()
        sq = torch.empty(M, N // 64)
        prod = Lo4Act.empty(M, F, dtype, "cpu", sel=s)
        if mode == "fast":
            h = torch.empty(M, N, dtype=dtype)
            ops.gemm_ex(a.hi, w, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=gamma, rowsq_out=sq)
            ops.gemm_ex(h, gu_w, prod.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=N, norm_eps=1e-5)
            return xs, h, prod
        h = Lo4Act.empty(M, N, dtype, "cpu", sel=s)
        ops.gemm_lo4(a, w, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq, out4=h)
        ops.gemm_lo4(h, gu_w, gu4, prod.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=N, norm_eps=1e-5, out4=prod)
        return xs, h, prod

    ops.set_option("gemm.config", cfg)
    try:
        (x_f, h_f, p_f), (x_l, h_l, p_l), (x_s, h_s, p_s) = run("fast"), run("lo4"), run("sel")
        assert torch.equal(x_s[keep], x_l[keep]) and torch.equal(x_s[~keep], x_f[~keep])
        assert torch.equal(h_s.hi[keep], h_l.hi[keep]) and torch.equal(h_s.hi[~keep], h_f[~keep])
        assert torch.equal(h_s.img[keep], h_l.img[keep]) and h_s.img[~keep].abs().max() == 0 and h_s.sc[~keep].abs().max() == 0
        assert torch.equal(p_s.hi[keep], p_l.hi[keep]) and torch.equal(p_s.hi[~keep], p_f.hi[~keep])
        assert torch.equal(p_s.img[keep], p_l.img[keep]) and p_s.img[~keep].abs().max() == 0
    finally:
        ops.set_option("gemm.config", -1)


def test_norm_and_attention_lo4_row_selection(ops):
    dtype = torch.float16
    M, D = 70, 256
    x = rnd((M, D), torch.float32, 80, 2.0)
    w = torch.rand(D, generator=torch.Generator().manual_seed(81)) + 0.5
    sel = _selection(M, [0, 5, 63, 64, 69])
    keep = sel[0].bool()
    full, part = Lo4Act.empty(M, D, dtype, "cpu"), Lo4Act.empty(M, D, dtype, "cpu", sel=sel)
    ops.norm_lo4(x, w, None, full, 1e-5)
    ops.norm_lo4(x, w, None, part, 1e-5)
    assert torch.equal(part.hi, full.hi)
    assert torch.equal(part.img[keep], full.img[keep]) and torch.equal(part.sc[keep], full.sc[keep])
    assert part.img[~keep].abs().max() == 0 and part.sc[~keep].abs().max() == 0
    # attention: two sequences, head_dim 128, causal
    from leopard_amd.ops import lo4_head_k4
    H, KV, hd = 2, 1, 128
    lens = [150, 37]
    S = sum(lens)
    cu = torch.tensor([0, lens[0], S], dtype=torch.int32)
    q, k, v = rnd((S, H * hd), dtype, 82), rnd((S, KV * hd), dtype, 83), rnd((S, KV * hd), dtype, 84)
    sel = _selection(S, list(range(140, 150)) + [S - 1])
    keep = sel[0].bool()
    k4 = lo4_head_k4(H, hd)
    full, part = Lo4Act.empty(S, H * hd, dtype, "cpu", k4=k4), Lo4Act.empty(S, H * hd, dtype, "cpu", k4=k4, sel=sel)
    ops.attention_lo4(q, k, v, full, cu, cu, max(lens), H, KV, hd, hd ** -0.5, True)
    ops.attention_lo4(q, k, v, part, cu, cu, max(lens), H, KV, hd, hd ** -0.5, True)
    assert torch.equal(part.hi, full.hi)
    assert torch.equal(part.img[keep], full.img[keep]) and torch.equal(part.sc[keep], full.sc[keep])
    assert part.img[~keep].abs().max() == 0 and part.sc[~keep].abs().max() == 0 and full.img[~keep].abs().max() > 0


def test_engine_lo4_rows_policy_and_packed_equals_separate():
    """LeopardEngine.lo4_rows: 'auto' / 'all' / an int — the selection is a property of a row's distance from the end of ITS sequence, so a
    packed batch gives each sample exactly the logits of its own prefill; the selected-rows run lands on the oracle that corrects those rows."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import LeopardEngine
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from leopard_amd.weights import EngineWeights, SynthSource
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=3, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    assert eng.lo4_rows == "auto" and eng.lo4_tail_rows(228) == 228 and eng.lo4_tail_rows(7187) == eng.LO4_TAIL_ROWS
    assert eng._lo4_selection([228, 100]) is None                     # every row of every (short) sequence: no tables
    eng.lo4_rows = 3
    row, unit, ranges = eng._lo4_selection([70, 9])
    assert row.tolist() == [0] * 67 + [1] * 3 + [0] * 6 + [1] * 3 and unit.tolist() == [0, 1] and ranges.tolist() == [[67, 70], [76, 79]]
    tiles = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (2, 28, 28, 3), dtype=np.uint8))
    ids_a = torch.tensor([[5, 250, 9, 250, 17, 33, 101, 7, 3, 11, 200, 90]])
    ids_b = torch.tensor([[5, 250, 9, 17, 33]])
    eng.precision = "lo4"
    one_a, one_b = eng.prefill(ids_a, tiles).logits_last.clone(), eng.prefill(ids_b, tiles[:1]).logits_last.clone()
    both, _ = eng.prefill_batch([(ids_a, tiles), (ids_b, tiles[:1])])
    assert torch.equal(both[0], one_a.reshape(-1)) and torch.equal(both[1], one_b.reshape(-1))
    # where the run sits: the oracle with the correction on the last 3 rows of the LLM stream
    pix = torch.from_numpy(siglip_normalize(tiles.numpy()))
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    ref = O.prefill_logits(ids_a, pix, Wt, cfg, last_only=True)[0, 0]
    S = eng.prefill(ids_a, tiles).seq_len
    with O.emulate_rounding(dtype, lo_sites=("llm.norm", "llm.attn_out", "llm.mlp_act"), lo_row_start=S - 3):
        pred = O.prefill_logits(ids_a, pix, Wt, cfg, last_only=True)[0, 0]
    scale = ref.abs().max().item()
    e_got, e_pred = (one_a.reshape(-1) - ref).abs().max().item() / scale, (pred - ref).abs().max().item() / scale
    assert 0.5 * e_pred <= e_got <= 2.0 * e_pred, (e_got, e_pred)
    eng.lo4_rows = "all"
    all_rows = eng.prefill(ids_a, tiles).logits_last.clone()
    eng.lo4_rows = 10 ** 6                                             # more rows than the sequence has = every row
    assert torch.equal(eng.prefill(ids_a, tiles).logits_last, all_rows)
