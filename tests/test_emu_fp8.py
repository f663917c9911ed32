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
