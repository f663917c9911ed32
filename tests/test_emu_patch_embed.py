"""Kernel-logic tests (CPU emulator) of the SURVEY.md 8(b) entry points added in round 2:

  * lmi_patch_embed — normalise + im2col + patch convolution + bias + position embedding in one MFMA GEMM whose A tiles are staged
    from the image itself: against torch's conv2d over the processor-normalised pixels (rounded to the operand type, as the
    kernel rounds them), u8 tiles and fp32 pixel_values giving bit-identical results, ragged row tails, several tile sizes;
  * lmi_kv_append, lmi_gemm_bias_act."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from leopard_amd import _lib
from leopard_amd.tiler import siglip_normalize
from leopard_amd.weights import interleave_gate_up, patch_weight_image_order
from tests.emu_util import emu_ops


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


def test_patch_weight_image_order_layout():
    w = torch.arange(2 * 3 * 14 * 14, dtype=torch.float32).reshape(2, 3, 14, 14)
    f = patch_weight_image_order(w, 14)
    assert f.shape == (2, 704)
    for (d, c, ky, kx) in [(0, 0, 0, 0), (1, 2, 13, 13), (0, 1, 5, 7), (1, 0, 9, 2)]:
        assert f[d, ky * 48 + kx * 3 + c] == w[d, c, ky, kx]
    pad = torch.ones(704, dtype=torch.bool)
    for ky in range(14):
        pad[ky * 48:ky * 48 + 42] = False
    assert bool((f[:, pad] == 0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,S,N", [(3, 28, 128), (2, 56, 256), (5, 70, 128)])
def test_patch_embed_matches_conv2d(ops, dtype, n, S, N):
    """M = n * (S/14)^2 = 12 / 32 / 125 rows: every case has a ragged 128-row tile; N = 256 exercises two column tiles."""
    P, G = 14, S // 14
    g = torch.Generator().manual_seed(n * 100 + S)
    u8 = torch.from_numpy(np.random.default_rng(S + n).integers(0, 256, (n, S, S, 3), dtype=np.uint8))
    w = (torch.randn(N, 3, P, P, generator=g) * 0.05).to(dtype)
    bias, pos = torch.randn(N, generator=g), torch.randn(G * G, N, generator=g)
    wf = patch_weight_image_order(w, P)
    out = torch.full((n * G * G, N), float("nan"))
    ops.patch_embed(u8, wf, bias, pos, out, S, P)
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))                       # the processor's fp32 pixel_values [n, 3, S, S]
    ref = F.conv2d(pix.to(dtype).float(), w.float(), bias, stride=P)            # operands rounded as the kernel rounds them
    ref = ref.flatten(2).transpose(1, 2).reshape(n * G * G, N) + pos.repeat(n, 1)
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())      # fp32 accumulation order only
    out32 = torch.full_like(out, float("nan"))
    ops.patch_embed(pix.contiguous(), wf, bias, pos, out32, S, P)               # reference-shaped fp32 input: the same bits
    assert torch.equal(out32, out)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_patch_embed_every_byte_value_matches_the_processor(ops, dtype):
    """All 256 pixel values through the kernel's u8 -> operand conversion (f16: one FMA, proven here to give the processor's
    16-bit value for every input; bf16: the processor's three-step arithmetic): identity weights read the operands back."""
    P, S, N = 14, 28, 640
    vals = torch.arange(4 * 588, dtype=torch.int64) % 256                      # 4 patches x 588 pixel values: every byte 9 times
    u8 = torch.zeros(1, S, S, 3, dtype=torch.uint8)
    for m in range(4):
        py, px = divmod(m, 2)
        u8[0, py * P:(py + 1) * P, px * P:(px + 1) * P, :] = vals[m * 588:(m + 1) * 588].reshape(P, P, 3).to(torch.uint8)
    w = torch.zeros(N, 3, P, P)
    for k in range(588):                                                        # output column k = pixel (ky, kx, c) of the patch
        ky, r = divmod(k, 42)
        kx, c = divmod(r, 3)
        w[k, c, ky, kx] = 1.0
    out = torch.empty(4, N)
    ops.patch_embed(u8, patch_weight_image_order(w.to(dtype), P), torch.zeros(N), torch.zeros(4, N), out, S, P)
    want = torch.from_numpy(siglip_normalize(np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)))[0, 0].reshape(256).to(dtype).float()
    assert torch.equal(out[:, :588].reshape(-1), want[vals])


def test_patch_embed_rejects_bad_arguments(ops):
    u8 = torch.zeros(1, 28, 28, 3, dtype=torch.uint8)
    wf = torch.zeros(128, 704, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="weight rows must hold 704"):
        ops.patch_embed(u8, wf[:, :640].contiguous(), torch.zeros(128), torch.zeros(4, 128), torch.zeros(4, 128), 28, 14)
    with pytest.raises(RuntimeError, match="bad shape"):
        ops.patch_embed(u8, wf, torch.zeros(128), torch.zeros(4, 128), torch.zeros(4, 128), 28, 13)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kv_append(ops, dtype):
    g = torch.Generator().manual_seed(3)
    pool_k, pool_v = torch.randn(40, 256, generator=g).to(dtype), torch.randn(40, 256, generator=g).to(dtype)
    kc, vc = torch.zeros(32, 256, dtype=dtype), torch.zeros(32, 256, dtype=dtype)
    ops.kv_append(pool_k[7:24], pool_v[7:24], kc, vc, 5)
    assert torch.equal(kc[5:22], pool_k[7:24]) and torch.equal(vc[5:22], pool_v[7:24])
    assert bool((kc[:5] == 0).all()) and bool((kc[22:] == 0).all()) and bool((vc[:5] == 0).all()) and bool((vc[22:] == 0).all())
    ops.kv_append(pool_k[:0], pool_v[:0], kc, vc, 0)                              # empty: no launch
    with pytest.raises(AssertionError):
        ops.kv_append(pool_k[:30], pool_v[:30], kc, vc, 5)                        # would run past the cache


def test_gemm_bias_act_equals_lmi_gemm(ops):
    g = torch.Generator().manual_seed(4)
    a = torch.randn(70, 128, generator=g).half()
    w = (torch.randn(256, 128, generator=g) * 0.1).half()
    bias = torch.randn(256, generator=g)
    for act in (_lib.ACT_NONE, _lib.ACT_GELU_TANH, _lib.ACT_GELU_ERF):
        o1, o2 = torch.empty(70, 256, dtype=torch.float16), torch.empty(70, 256, dtype=torch.float16)
        ops.gemm(a, w, o1, bias=bias, act=act)
        ops.gemm_bias_act(a, w, o2, bias=bias, act=act)
        assert torch.equal(o1, o2)
    x1 = torch.randn(70, 256, generator=g)
    x2 = x1.clone()
    ops.gemm(a, w, x1, bias=bias, epilogue=_lib.EPI_RESIDUAL)
    ops.gemm_bias_act(a, w, x2, bias=bias, residual=True)
    assert torch.equal(x1, x2)
    wi = interleave_gate_up(w[:128], w[128:])
    s1, s2 = torch.empty(70, 128, dtype=torch.float16), torch.empty(70, 128, dtype=torch.float16)
    ops.gemm(a, wi, s1, epilogue=_lib.EPI_SWIGLU)
    ops.gemm_bias_act(a, wi, s2, act=_lib.ACT_SWIGLU)
    assert torch.equal(s1, s2)
    with pytest.raises(RuntimeError, match="SwiGLU has no residual"):
        ops.gemm_bias_act(a, wi, x2, act=_lib.ACT_SWIGLU, residual=True)
