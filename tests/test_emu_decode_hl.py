"""Kernel-logic tests (CPU emulator) of the DECODE PRECISION MODE (round 6; csrc/skinny.h "hl", lmi_*_hl): every operand of a decode-step
projection is a PAIR of 16-bit rows — T(x) and T(x - T(x)) — summed into the same fp32 accumulators, so the hand-over roundings of the token's
own path (which its logits' error is made of: tools/lo4_policy_study.py, rows=1) are gone at no extra weight traffic.  Reference served: the
decode branch of the reference forward runs fp32 (evaluations/models/llava_multiimg_siglip_anyres.py:291-320,373)."""
import numpy as np
import pytest
import torch

from leopard_amd.weights import interleave_gate_up, rope_permute_rows, skinny_pack
from tests.emu_util import emu_ops


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def pair(x32, dtype):
    hi = x32.to(dtype)
    lo = (x32 - hi.float()).to(dtype)
    return torch.cat([hi, lo], 0).contiguous()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 3, 8])
def test_split_rows_and_gemm_skinny_hl_every_epilogue(ops, dtype, M):
    N, K = 128, 384
    x32 = rnd((M, K), torch.float32, 1, 2.0)
    w = rnd((N, K), dtype, 2, 0.1)
    X = torch.full((2 * M + 1, K), 7.0, dtype=dtype)
    ops.split_rows_hl(x32, X[:2 * M])
    assert torch.equal(X[:2 * M], pair(x32, dtype)) and bool((X[2 * M] == 7.0).all())
    X = X[:2 * M].contiguous()
    exact = x32 @ w.float().T                                                # what the pair buys: the product of the UNROUNDED operand
    seen = (X[:M].float() + X[M:].float()) @ w.float().T                     # what the kernel is given
    single = X[:M].float() @ w.float().T
    scale = exact.abs().max().item()
    for packed in (False, True):
        wr = skinny_pack(w) if packed else w
        o32 = torch.zeros(M, N)
        ops.gemm_skinny(wr, X, o32, 3, packed=packed, hl=True)
        assert (o32 - seen).abs().max() <= 2e-6 * scale
        assert (o32 - exact).abs().max() < 0.02 * (single - exact).abs().max()      # ~2^-11 of the one-row error (fp16) / 2^-8 (bf16)
        acc = rnd((M, N), torch.float32, 3)
        acc0 = acc.clone()
        ops.gemm_skinny(wr, X, acc, 1, packed=packed, hl=True)
        assert (acc - (acc0 + seen)).abs().max() <= 4e-6 * max(scale, 1.0)
        # STORE: the 16-bit result as a pair again
        o = torch.full((2 * M + 1, N), 7.0, dtype=dtype)
        ops.gemm_skinny(wr, X, o[:2 * M], 0, packed=packed, hl=True)
        assert torch.equal(o[:2 * M], pair(o32, dtype)) and bool((o[2 * M] == 7.0).all())
    # SwiGLU: products as a pair
    F = N // 2
    gu = interleave_gate_up(w[:F].contiguous(), w[F:].contiguous())
    lin = (X[:M].float() + X[M:].float()) @ gu.float().T
    lv = lin.view(M, N // 64, 2, 32)
    want = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, F)
    o = torch.zeros(2 * M, F, dtype=dtype)
    ops.gemm_skinny(gu, X, o, 2, hl=True)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert ((o[:M].float() + o[M:].float()) - want).abs().max() <= 4 * eps * eps * max(1.0, want.abs().max().item()) + 2e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M", [1, 8])
def test_gemm_skinny_hl_folded_norm_producer_and_consumer(ops, M):
    """Producer (residual + T(x gamma) as a pair + row partials) feeding the SwiGLU consumer with the folded row scale, all on pairs."""
    dtype, D, F = torch.float16, 256, 128
    a32 = rnd((M, D), torch.float32, 10, 2.0)
    w_o = rnd((D, D), dtype, 11, 0.1)
    x0 = rnd((M, D), torch.float32, 12)
    gamma = torch.rand(D, generator=torch.Generator().manual_seed(13)) + 0.5
    A = pair(a32, dtype)
    xs = x0.clone()
    h = torch.zeros(2 * M, D, dtype=dtype)
    sq = torch.full((M, D // 16), float("nan"))
    ops.gemm_skinny(w_o, A, xs, 1, norm_out=h, norm_gamma=gamma, rowsq_out=sq, hl=True)
    x_ref = x0 + (A[:M].float() + A[M:].float()) @ w_o.float().T
    assert (xs - x_ref).abs().max() <= 4e-6 * x_ref.abs().max()
    assert torch.equal(h, pair(xs * gamma, dtype))
    assert (sq - xs.pow(2).view(M, D // 16, 16).sum(-1)).abs().max() <= 1e-4 * sq.abs().max()
    gu = interleave_gate_up(rnd((F, D), dtype, 14, 0.1), rnd((F, D), dtype, 15, 0.1))
    prod = torch.zeros(2 * M, F, dtype=dtype)
    ops.gemm_skinny(gu, h, prod, 2, rowsq_in=sq, norm_dim=D, norm_eps=1e-5, hl=True)
    rstd = torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-5)
    lin = ((h[:M].float() + h[M:].float()) @ gu.float().T) * rstd
    lv = lin.view(M, (2 * F) // 64, 2, 32)
    want = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, F)
    got = prod[:M].float() + prod[M:].float()
    assert (got - want).abs().max() <= 3e-6 * max(1.0, want.abs().max().item())
    # against the fp32 definition of the half layer the pair path is far closer than the one-row path
    exact = torch.nn.functional.silu(((xs * gamma * rstd) @ gu.float().T).view(M, -1, 2, 32)[:, :, 0]) * ((xs * gamma * rstd) @ gu.float().T).view(M, -1, 2, 32)[:, :, 1]
    one = torch.zeros(M, F, dtype=dtype)
    h1 = (xs * gamma).to(dtype)
    ops.gemm_skinny(gu, h1, one, 2, rowsq_in=sq, norm_dim=D, norm_eps=1e-5)
    e_pair, e_one = (got - exact.reshape(M, F)).abs().max().item(), (one.float() - exact.reshape(M, F)).abs().max().item()
    assert e_pair < 0.05 * e_one, (e_pair, e_one)


@pytest.mark.parametrize("packed", [False, True])
def test_rope_qkv_skinny_hl(ops, packed):
    dtype, H, KV, hd, cap, B, K = torch.float16, 2, 1, 128, 12, 3, 384
    w = rnd(((H + 2 * KV) * hd, K), dtype, 7, 0.1)
    x32 = rnd((B, K), torch.float32, 8, 2.0)
    X = pair(x32, dtype)
    f = torch.arange(cap).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = torch.tensor([4, 0, 11], dtype=torch.int32)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    kp, vp = torch.zeros(B * cap, KV * hd, dtype=dtype), torch.zeros(B * cap, KV * hd, dtype=dtype)
    got = torch.full((B + 1, (H + 2 * KV) * hd), 7.0, dtype=dtype)
    ops.rope_qkv_skinny(skinny_pack(w_rope) if packed else w_rope, X, got[:B], H, KV, hd, cos, sin, kp, vp, cap, pos, packed=packed, hl=True)
    assert bool((got[B] == 7.0).all())
    lin = x32 @ w.float().T                                                 # the UNROUNDED operand
    ref = lin.clone()
    for s in range(B):
        c, sn = cos[pos[s]], sin[pos[s]]
        for hh in range(H + KV):
            a, b = lin[s, hh * hd:hh * hd + 64], lin[s, hh * hd + 64:(hh + 1) * hd]
            ref[s, hh * hd:hh * hd + 64] = a * c - b * sn
            ref[s, hh * hd + 64:(hh + 1) * hd] = b * c + a * sn
    # one rounding (the output's own) away from the fp32 statement on the unrounded operand
    assert ((got[:B].float() - ref).abs() <= 2.0 ** -11 * ref.abs() + 1e-5 * ref.abs().max()).all()
    for s in range(B):
        r = s * cap + int(pos[s])
        assert torch.equal(kp[r], got[s, H * hd:(H + KV) * hd]) and torch.equal(vp[r], got[s, (H + KV) * hd:])


def test_attention_decode_hl_writes_the_output_rows_as_pairs(ops):
    dtype, H, KV, hd, cap = torch.float16, 4, 1, 128, 256
    B, lens = 2, [200, 77]
    q = rnd((B, H * hd), dtype, 20)
    k, v = rnd((B * cap, KV * hd), dtype, 21), rnd((B * cap, KV * hd), dtype, 22)
    cu_q = torch.arange(B + 1, dtype=torch.int32)
    k_begin = (torch.arange(B, dtype=torch.int32) * cap)
    k_len = torch.tensor(lens, dtype=torch.int32)
    ws = torch.empty(ops.decode_workspace_elems(B, H, hd, cap), dtype=torch.float32)
    one = torch.zeros(B, H * hd, dtype=dtype)
    two = torch.full((2 * B + 1, H * hd), 7.0, dtype=dtype)
    ops.attention_decode_pool(q, k, v, one, cu_q, k_begin, k_len, cap, H, KV, hd, hd ** -0.5, ws)
    ops.attention_decode_pool(q, k, v, two[:2 * B], cu_q, k_begin, k_len, cap, H, KV, hd, hd ** -0.5, ws, hl=True)
    assert torch.equal(two[:B], one) and bool((two[2 * B] == 7.0).all())
    ref = torch.zeros(B, H * hd)
    for s in range(B):
        ks, vs = k[s * cap:s * cap + lens[s]].float(), v[s * cap:s * cap + lens[s]].float()
        for hh in range(H):
            p = torch.softmax((q[s, hh * hd:(hh + 1) * hd].float() @ ks.T) * hd ** -0.5, -1)
            ref[s, hh * hd:(hh + 1) * hd] = p @ vs
    e_one = (one.float() - ref).abs().max().item()
    e_two = (two[:B].float() + two[B:2 * B].float() - ref).abs().max().item()
    assert e_two < 0.8 * e_one, (e_two, e_one)                            # what is left is the kernel's own arithmetic: P handed to the P.V MFMA in 16 bits
    lo = two[B:2 * B].float()
    assert (lo.abs() <= 2.0 ** -11 * two[:B].float().abs() + 1e-7).all() and lo.abs().max() > 0      # a rounding residual: at most half an ulp of its hi row


def test_engine_decode_precision_mode():
    """precision = "lo4": the decode steps run on operand pairs (decode_hl) — closer to the fp32 oracle's next-token logits than the fast step,
    tokens of generate() unchanged, state rebuilt when the mode changes; LMI_DECODE_PRECISION=0 / decode_precision = False keeps the fast step."""
    from leopard_amd.config import LeopardConfig, RopeScaling, TextConfig, VisionConfig
    from leopard_amd.engine import KVCache, LeopardEngine
    from leopard_amd.synth import synth_state_dict_numpy
    from leopard_amd.tiler import siglip_normalize
    from leopard_amd.weights import EngineWeights, SynthSource
    from oracle import leopard_oracle as O
    ops = emu_ops()
    cfg = LeopardConfig(
        vision_config=VisionConfig(hidden_size=1152, intermediate_size=100, num_hidden_layers=1, num_attention_heads=16,
                                   image_size=28, patch_size=14),
        text_config=TextConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2,
                               num_key_value_heads=1, vocab_size=256, rope_scaling=RopeScaling()),
        image_token_index=250)
    dtype = torch.float16
    W = EngineWeights.build(cfg, SynthSource(cfg, ops, "cpu", dtype), dtype)
    eng = LeopardEngine(cfg, W, ops=ops, device="cpu")
    assert eng.llm_packed and not eng.decode_hl(1)
    tiles = torch.from_numpy(np.random.default_rng(7).integers(0, 256, (1, 28, 28, 3), dtype=np.uint8))
    ids = torch.tensor([[5, 250, 9, 17, 33, 101, 7]])
    pix = torch.from_numpy(siglip_normalize(tiles.numpy()))
    Wt = O.weights_from_numpy(synth_state_dict_numpy(cfg))
    nxt = 42
    ref = O.prefill_logits(torch.cat([ids, torch.tensor([[nxt]])], 1), pix, Wt, cfg, last_only=True)[0, 0]

    def step(precision, decode_precision=True):
        eng.precision, eng.decode_precision = precision, decode_precision
        cache = KVCache(cfg, 64, dtype, "cpu")
        eng.prefill(ids, tiles, cache=cache)
        return eng.decode_step(nxt, cache).clone(), cache._decode_state.hl

    fast, hl_f = step("fast")
    lo4, hl_l = step("lo4")
    lo4_fastdec, hl_x = step("lo4", decode_precision=False)
    assert (hl_f, hl_l, hl_x) == (False, True, False)
    scale = ref.abs().max().item()
    e = {k: (v - ref).abs().max().item() / scale for k, v in (("fast", fast), ("lo4", lo4), ("lo4_fastdec", lo4_fastdec))}
    # (1 + 4 layers, S = 8: the uncorrected SigLIP layer and the attention operands dominate here — what the pairs buy at full depth is measured on
    # the device against the committed decode fixture, tests/test_gpu_decode_fixture.py; here: the mode runs, moves the logits, and does not hurt)
    assert e["lo4"] <= e["fast"] and not torch.equal(lo4, lo4_fastdec) and e["lo4"] <= 1.1 * e["lo4_fastdec"], e
    # batched step on pairs == the batch-1 step on pairs (same kernels, one row), and generate() still works
    eng.precision, eng.decode_precision = "lo4", True
    a = eng.generate(ids, tiles, max_new_tokens=5, eos_token_id=())
    b = eng.generate_batch([(ids, tiles), (ids, tiles)], max_new_tokens=5, eos_token_id=())
    assert torch.equal(a[0], b[0][0]) and torch.equal(a[0], b[1][0])
    assert eng._batch_states[2].hl
    eng.precision = "fast"
    c = eng.generate(ids, tiles, max_new_tokens=5, eos_token_id=())
    assert c.shape == a.shape
