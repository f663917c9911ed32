"""-m gpu: the PRODUCTION kernels at PRODUCTION shapes against fp32 references computed on the device.

What the step actually runs (profiles/README.md): `gemm_stagger_kernel` 256x256 for every Llama projection and SigLIP qkv / fc1
(M = 7187 = 28 x 256 + 19 rows -> the tail row-tile with its padding-block skip; M = 28392), the 256x128 3-slot ring for SigLIP
fc2, 128x128 for out_proj / patch embedding, and `attn_fwd_dma_kernel` at S = 7187 (d = 128, causal GQA) and 42 x 676 (d = 72).
The small shapes of tests/test_gpu_kernels.py dispatch to the small-M ring and never reach those kernels, so here every geometry /
schedule (`gemm.config` 0..8) is forced on every epilogue at M in {7187, 28392, 566}, repeated to catch a schedule hazard (a race
between the LDS-DMA ring and the fragment reads gives run-to-run differences or wrong tiles), and compared element by element
with `a.float() @ w.float().T` (rocBLAS fp32 on the same 16-bit operand values: the only differences are summation order and
the output rounding)."""
import pytest
import torch

from leopard_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
# 0-4 ring geometries (128x128/2, 256x256/2, 256x128/3, 256x128/2, 128x256/3), 5-7 staggered 256x256 variants, 8 small-M ring
ALL_CFGS = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10]


@pytest.fixture(scope="module")
def ops():
    from leopard_amd.ops import Ops
    o = Ops()
    yield o
    o.set_option("gemm.config", -1)


def eps(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7


_cache = {}


def operands(M, N, K, dtype, scale_w):
    """Device-generated operands (one set per shape and dtype, reused across configs)."""
    key = (M, N, K, dtype)
    if key not in _cache:
        if len(_cache) > 2:
            _cache.clear()
            torch.cuda.empty_cache()
        g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
        a = torch.randn(M, K, generator=g, device=DEV).to(dtype)
        w = (torch.randn(N, K, generator=g, device=DEV) * scale_w).to(dtype)
        bias = torch.randn(N, generator=g, device=DEV)
        ref = a.float() @ w.float().T
        _cache[key] = (a, w, bias, ref)
    return _cache[key]


def rel_err(out, ref):
    return ((out.float() - ref).abs() / (1.0 + ref.abs())).max().item()


def run3(fn):
    """Three launches: results must be bit-identical (a DMA/read race shows up as run-to-run differences)."""
    o0 = fn()
    for _ in range(2):
        assert torch.equal(fn(), o0), "two launches of the same GEMM differ"
    return o0


# (name, M, N, K): the GEMM shapes of the C3 step and of the mid configuration (S = 566: three row tiles)
LLAMA_SHAPES = [("qkv", 7187, 6144, 4096), ("o_proj", 7187, 4096, 4096), ("down", 7187, 4096, 14336), ("qkv_mid", 566, 6144, 4096)]
SIGLIP_SHAPES = [("vit_qkv", 28392, 3456, 1152), ("vit_out", 28392, 1152, 1152), ("vit_fc2", 28392, 1152, 4352),
                 ("patch", 28392, 1152, 640)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", ALL_CFGS)
@pytest.mark.parametrize("shape", LLAMA_SHAPES + SIGLIP_SHAPES, ids=lambda s: s[0])
def test_gemm_store_and_residual_every_config(ops, dtype, cfg, shape):
    """EPI_STORE (+bias) and EPI_RESIDUAL (fp32 +=) on every geometry / schedule."""
    _, M, N, K = shape
    a, w, bias, ref = operands(M, N, K, dtype, 0.02)
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.empty(M, N, dtype=dtype, device=DEV)

        def store():
            out.fill_(float("nan"))
            ops.gemm(a, w, out, bias=bias)
            return out.clone()
        o = run3(store)
        e = rel_err(o, ref + bias)
        assert e <= 3 * eps(dtype), f"STORE cfg {cfg} {shape}: {e:.3e}"
        x0 = torch.randn(M, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))

        def resid():
            x = x0.clone()
            ops.gemm(a, w, x, bias=bias, epilogue=_lib.EPI_RESIDUAL)
            return x
        x = run3(resid)
        # fp32 output: only the summation order differs from the reference
        assert (x - (x0 + ref + bias)).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item()), f"RESID cfg {cfg} {shape}"
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", ALL_CFGS)
@pytest.mark.parametrize("M", [7187, 566])
def test_gemm_swiglu_gate_up_every_config(ops, dtype, cfg, M):
    """The dominant kernel of the step: Llama gate/up + SwiGLU, N = 28672 interleaved rows, K = 4096."""
    from leopard_amd.weights import interleave_gate_up
    F, K = 14336, 4096
    a, w, _, ref = operands(M, 2 * F, K, dtype, 0.02)                  # rows 0..F-1 = gate, F..2F-1 = up
    wi = interleave_gate_up(w[:F], w[F:])
    want = torch.nn.functional.silu(ref[:, :F]) * ref[:, F:]
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.empty(M, F, dtype=dtype, device=DEV)

        def go():
            out.fill_(float("nan"))
            ops.gemm(a, wi, out, epilogue=_lib.EPI_SWIGLU)
            return out.clone()
        o = run3(go)
        e = rel_err(o, want)
        assert e <= 3 * eps(dtype), f"SWIGLU cfg {cfg} M {M}: {e:.3e}"
    finally:
        ops.set_option("gemm.config", -1)
        del wi


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_gemm_activations_store_f32_addmat_every_config(ops, dtype, cfg):
    """SigLIP fc1 (GELU-tanh, N padded 4304 -> 4352), the projector's GELU-erf, and the patch-embedding epilogue
    (fp32 store + bias + position-table row m % 676) at M = 28392."""
    M = 28392
    a, w, bias, ref = operands(M, 4352, 1152, dtype, 0.02)
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.empty(M, 4352, dtype=dtype, device=DEV)
        ops.gemm(a, w, out, bias=bias, act=_lib.ACT_GELU_TANH)
        e = rel_err(out, torch.nn.functional.gelu(ref + bias, approximate="tanh"))
        assert e <= 3 * eps(dtype), f"GELU-tanh cfg {cfg}: {e:.3e}"
        ops.gemm(a, w, out, bias=bias, act=_lib.ACT_GELU_ERF)
        e = rel_err(out, torch.nn.functional.gelu(ref + bias))
        assert e <= 3 * eps(dtype), f"GELU-erf cfg {cfg}: {e:.3e}"
        del out
        a, w, bias, ref = operands(M, 1152, 640, dtype, 0.02)
        pos = torch.randn(676, 1152, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
        o32 = torch.full((M, 1152), float("nan"), device=DEV)
        ops.gemm(a, w, o32, bias=bias, addmat=pos, epilogue=_lib.EPI_STORE_F32)
        want = ref + bias + pos[torch.arange(M, device=DEV) % 676]
        assert (o32 - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), f"STORE_F32+addmat cfg {cfg}"
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_gemm_pixel_shuffle_42_tiles_every_config(ops, dtype, cfg):
    """Projector linear_1 at the C3 size: A gathered through the 2x2 pixel shuffle from the ViT output of 42 tiles
    (M = 42 x 169 = 7098 shuffled rows, K = 4 x 1152), GELU-erf."""
    tiles, G, C, N = 42, 26, 1152, 4096
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.randn(tiles * G * G, C, generator=g, device=DEV).to(dtype)
    w = (torch.randn(N, 4 * C, generator=g, device=DEV) * 0.02).to(dtype)
    bias = torch.randn(N, generator=g, device=DEV)
    # closed form of EVAL:165-176: out[n, ph*13 + pw, (dh*2 + dw)*C + c] = x[n, (2ph + dh)*26 + 2pw + dw, c]
    shuf = x.view(tiles, G // 2, 2, G // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(tiles * 169, 4 * C)
    want = torch.nn.functional.gelu(shuf.float() @ w.float().T + bias)
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.full((tiles * 169, N), float("nan"), dtype=dtype, device=DEV)
        ops.gemm(x, w, out, bias=bias, act=_lib.ACT_GELU_ERF, a_mode=_lib.A_PIXEL_SHUFFLE, ps_grid=G, M=tiles * 169)
        e = rel_err(out, want)
        assert e <= 3 * eps(dtype), f"PIXSHUF cfg {cfg}: {e:.3e}"
    finally:
        ops.set_option("gemm.config", -1)


def test_gemm_row_map_and_add_rows_at_full_m(ops):
    """Row scatter (row_map) and indexed position rows (add_rows) at M = 7187 on the production schedule."""
    dtype = torch.float16
    M, N, K = 7187, 4096, 4096
    a, w, bias, ref = operands(M, N, K, dtype, 0.02)
    perm = torch.randperm(M + 77, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))[:M].to(torch.int32)
    big = torch.zeros(M + 77, N, device=DEV)
    ops.gemm(a, w, big, bias=bias, row_map=perm, epilogue=_lib.EPI_STORE_F32)
    assert (big[perm.long()] - (ref + bias)).abs().max().item() <= 1e-4 * ref.abs().max().item()
    untouched = torch.ones(M + 77, dtype=torch.bool, device=DEV)
    untouched[perm.long()] = False
    assert big[untouched].abs().max().item() == 0
    table = torch.randn(4900, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    idx = torch.randint(0, 4900, (M,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)).to(torch.int32)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a, w, out, bias=bias, addmat=table, add_rows=idx, epilogue=_lib.EPI_STORE_F32)
    assert (out - (ref + bias + table[idx.long()])).abs().max().item() <= 1e-4 * ref.abs().max().item()


# ---- attention at the production sizes ------------------------------------------------------------------------------------
def sampled_rows(S):
    """Query rows that matter: first / last rows, every 128-row workgroup edge and 64-key tile edge region sampled, the tail."""
    rows = set([0, 1, 31, 32, 63, 64, 65, 127, 128, 129, S - 1, S - 2, S - 19, S - 20, S - 33, S - 64, S - 65, S - 128, S - 129])
    g = torch.Generator().manual_seed(S)
    rows |= set(int(r) for r in torch.randint(0, S, (420,), generator=g))
    rows |= set(range(4096 - 3, 4096 + 3)) | set(range(7168 - 2, min(S, 7168 + 2)))
    return torch.tensor(sorted(r for r in rows if 0 <= r < S), device=DEV)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("window", [0, 4096])
def test_attention_llama_s7187_vs_fp32_on_sampled_rows(ops, dtype, window):
    """attn_fwd_dma_kernel<128, causal>, S = 7187, 32 q / 8 kv heads (the C3 Llama attention; window = 4096 is the Mistral
    sliding window of the Idefics2 path): ~500 query rows x all heads against an fp32 softmax(QK^T)V over their full key range."""
    H, KV, D, S = 32, 8, 128, 7187
    g = torch.Generator(device=DEV).manual_seed(70)
    qkv = torch.randn(S, (H + 2 * KV) * D, generator=g, device=DEV).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    out = torch.full((S, H * D), float("nan"), dtype=dtype, device=DEV)
    ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True, window=window)
    assert torch.isfinite(out.float()).all()
    rows = sampled_rows(S)
    qs = q[rows].float().view(-1, H, D).transpose(0, 1)                                   # [H, R, D]
    ks = k.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)           # [H, S, D]
    vs = v.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
    sc = qs @ ks.transpose(-1, -2) * D ** -0.5
    keys = torch.arange(S, device=DEV)[None, :]
    vis = keys <= rows[:, None]
    if window:
        vis &= rows[:, None] - keys < window
    sc = sc.masked_fill(~vis[None], float("-inf"))
    ref = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(len(rows), H * D)
    err = (out[rows].float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), f"S=7187 causal window={window}: {err:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_siglip_42_tiles_vs_fp32(ops, dtype):
    """attn_fwd_dma_kernel<72>, 42 sequences of 676 tokens, 16 heads x 72 (the C3 SigLIP attention), every row."""
    H, D, n, T = 16, 72, 42, 676
    g = torch.Generator(device=DEV).manual_seed(71)
    qkv = torch.randn(n * T, 3 * H * D, generator=g, device=DEV).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    cu = torch.arange(0, (n + 1) * T, T, dtype=torch.int32, device=DEV)
    out = torch.full((n * T, H * D), float("nan"), dtype=dtype, device=DEV)
    ops.attention(q, k, v, out, cu, cu, T, H, H, D, D ** -0.5, False, True)
    qs = q.float().view(n, T, H, D).permute(0, 2, 1, 3)
    ks = k.float().view(n, T, H, D).permute(0, 2, 1, 3)
    vs = v.float().view(n, T, H, D).permute(0, 2, 1, 3)
    ref = (torch.softmax(qs @ ks.transpose(-1, -2) * D ** -0.5, -1) @ vs).permute(0, 2, 1, 3).reshape(n * T, H * D)
    err = (out.float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), f"42 x 676 d=72: {err:.3e}"


def test_attention_deferred_rescale_branch_is_forced(ops):
    """cdna guide rule 26: the deferred-rescale branch (reference moves only when a row outgrows it by 2^8) is rare on random
    data, so force it: one key late in the sequence scores far above everything before it for a few query rows."""
    H, KV, D, S = 32, 8, 128, 1500
    dtype = torch.float16
    g = torch.Generator(device=DEV).manual_seed(72)
    qkv = (torch.randn(S, (H + 2 * KV) * D, generator=g, device=DEV) * 0.5).to(dtype)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    # key 900 of kv head 0 is aligned with query rows 1000..1003 of heads 0..3 (score ~ +28 after scaling, the others |s| < 10)
    for r in range(1000, 1004):
        q[r, :4 * D] = k[900, :D].repeat(4) * 10
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    out = torch.empty(S, H * D, dtype=dtype, device=DEV)
    ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    qs = q.float().view(S, H, D).transpose(0, 1)
    ks = k.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
    vs = v.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
    sc = qs @ ks.transpose(-1, -2) * D ** -0.5
    mask = torch.arange(S, device=DEV)[None, :] <= torch.arange(S, device=DEV)[:, None]
    ref = (torch.softmax(sc.masked_fill(~mask[None], float("-inf")), -1) @ vs).transpose(0, 1).reshape(S, H * D)
    assert sc[0, 1000, 900].item() - sc[0, 1000, :900].max().item() > 12        # the spike really outgrows the running max
    err = (out.float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), f"forced rescale: {err:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_lm_head_last_vs_fp32(ops, dtype):
    """lmi_lm_head_last at the Llama-3.1 vocabulary: 3 selected rows, fp32 normalised row x 16-bit weights."""
    N, K = 128256, 4096
    g = torch.Generator(device=DEV).manual_seed(80)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.02).to(dtype)
    x = torch.randn(50, K, generator=g, device=DEV) * 3
    gamma = torch.rand(K, generator=g, device=DEV) + 0.5
    rows = torch.tensor([49, 0, 17], device=DEV)
    out = torch.full((3, N), float("nan"), device=DEV)
    ops.lm_head_last(w, x, rows, gamma, 1e-5, out)
    xn = gamma * (x[rows] * torch.rsqrt(x[rows].pow(2).mean(-1, keepdim=True) + 1e-5))
    ref = (xn.double() @ w.double().T)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


# ---- RMSNorm / RoPE fused into the GEMM epilogues (lmi_gemm_ex, lmi_rmsnorm_rope) at the C3 shapes ----------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 1, 2, 5, 7])
@pytest.mark.parametrize("K", [4096, 14336])
def test_gemm_ex_residual_norm_producer_m7187(ops, dtype, cfg, K):
    """o_proj (K = 4096) / down_proj (K = 14336) at M = 7187: x += a.w^T plus the second output T(x * gamma) and the 64 per-row
    partial sums of squares; x must equal the plain residual epilogue bit for bit."""
    M, N = 7187, 4096
    a, w, _, ref = operands(M, N, K, dtype, 0.02)
    g = torch.Generator(device=DEV).manual_seed(91)
    x0 = torch.randn(M, N, generator=g, device=DEV)
    gamma = torch.rand(N, generator=g, device=DEV) + 0.5
    ops.set_option("gemm.config", cfg)
    try:
        x = x0.clone()
        h = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
        sq = torch.full((M, N // 64), float("nan"), device=DEV)
        ops.gemm_ex(a, w, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=gamma, rowsq_out=sq)
        x2 = x0.clone()
        ops.gemm(a, w, x2, epilogue=_lib.EPI_RESIDUAL)
        assert torch.equal(x, x2)
        assert (x - (x0 + ref)).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
        hx = x * gamma
        assert ((h.float() - hx).abs() <= eps(dtype) * hx.abs() + 2.0 ** -24).all()            # one rounding of x * gamma
        sq_ref = x.double().pow(2).view(M, N // 64, 64).sum(-1)
        assert ((sq.double() - sq_ref).abs() / sq_ref).max().item() <= 1e-5
    finally:
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 1, 5, 7, 10])
def test_gemm_ex_swiglu_consumer_m7187(ops, dtype, cfg):
    """gate/up + SwiGLU with the RMSNorm row scale applied to the accumulators (rowsq_in), M = 7187, N = 28672."""
    from leopard_amd.weights import interleave_gate_up
    M, F, K = 7187, 14336, 4096
    a, w, _, ref = operands(M, 2 * F, K, dtype, 0.02)
    wi = interleave_gate_up(w[:F], w[F:])
    g = torch.Generator(device=DEV).manual_seed(92)
    sq = (torch.rand(M, K // 64, generator=g, device=DEV) + 0.5) * 64
    rstd = torch.rsqrt(sq.sum(-1, keepdim=True) / K + 1e-5)
    want = torch.nn.functional.silu(ref[:, :F] * rstd) * (ref[:, F:] * rstd)
    ops.set_option("gemm.config", cfg)
    try:
        out = torch.full((M, F), float("nan"), dtype=dtype, device=DEV)
        ops.gemm_ex(a, wi, out, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
        e = rel_err(out, want)
        assert e <= 3 * eps(dtype), f"SWIGLU + row scale cfg {cfg}: {e:.3e}"
    finally:
        ops.set_option("gemm.config", -1)
        del wi


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 1, 2, 5, 7])
@pytest.mark.parametrize("with_norm", [False, True])
def test_rmsnorm_rope_qkv_s7187(ops, dtype, cfg, with_norm):
    """lmi_rmsnorm_rope at the C3 Llama shape: S = 7187, 32 q + 8 kv heads x 128, K = 4096; llama3-scaled tables at positions
    0..S-1; q / k rotated on the fp32 accumulators, K / V appended to the cache."""
    from leopard_amd.config import RopeScaling
    from leopard_amd.engine import llama3_inv_freq
    from leopard_amd.weights import rope_permute_rows
    S, nq, nkv, D, K = 7187, 32, 8, 128, 4096
    N = (nq + 2 * nkv) * D
    a, w, _, ref = operands(S, N, K, dtype, 0.02)
    w_rope = torch.cat([rope_permute_rows(w[:(nq + nkv) * D]), w[(nq + nkv) * D:]], 0).contiguous()
    inv = llama3_inv_freq(D, 5e5, RopeScaling()).to(DEV)
    ang = torch.arange(S, device=DEV, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    sq, rstd = None, 1.0
    if with_norm:
        g = torch.Generator(device=DEV).manual_seed(93)
        sq = (torch.rand(S, K // 64, generator=g, device=DEV) + 0.5) * 64
        rstd = torch.rsqrt(sq.sum(-1, keepdim=True) / K + 1e-5)
    acc = (ref * rstd).view(S, nq + 2 * nkv, D)
    rot = torch.cat((-acc[..., D // 2:], acc[..., :D // 2]), -1)
    c2, s2 = torch.cat([cos, cos], -1)[:, None], torch.cat([sin, sin], -1)[:, None]
    want = acc.clone()
    want[:, :nq + nkv] = acc[:, :nq + nkv] * c2 + rot[:, :nq + nkv] * s2
    want = want.view(S, N)
    ops.set_option("gemm.config", cfg)
    try:
        qkv = torch.full((S, N), float("nan"), dtype=dtype, device=DEV)
        kc = torch.zeros(S + 5, nkv * D, dtype=dtype, device=DEV)
        vc = torch.zeros_like(kc)
        ops.rmsnorm_rope(a, w_rope, qkv, sq, 1e-5, cos, sin, kc, vc, 2, nq, nkv, D)
        e = rel_err(qkv, want)
        assert e <= 3 * eps(dtype), f"qkv + rope cfg {cfg} norm {with_norm}: {e:.3e}"
        assert torch.equal(kc[2:2 + S], qkv[:, nq * D:(nq + nkv) * D]) and torch.equal(vc[2:2 + S], qkv[:, (nq + nkv) * D:])
        assert kc[:2].abs().max().item() == 0 and kc[2 + S:].abs().max().item() == 0
    finally:
        ops.set_option("gemm.config", -1)
        del w_rope


# ---- fp8 linears (lmi_gemm_fp8 / lmi_quantize_fp8, BASELINE config 5) at the C3 shapes --------------------------------------
F8 = torch.float8_e4m3fn


def test_quantize_fp8_matches_torch_bit_for_bit_on_device(ops):
    g = torch.Generator(device=DEV).manual_seed(95)
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        x = (torch.randn(7187, 4096, generator=g, device=DEV) * torch.logspace(-4, 3, 4096, device=DEV)[None, :]).to(dtype)
        out = torch.zeros(7187, 4096, dtype=torch.uint8, device=DEV)
        ops.quantize_fp8(x, out, 4.0)
        want = (x.float() * 4.0).clamp(-448, 448).to(F8).view(torch.uint8)
        same = (out == want) | (((out & 0x7F) == 0) & ((want & 0x7F) == 0))
        assert bool(same.all())


@pytest.mark.parametrize("out_dtype", DTYPES)
@pytest.mark.parametrize("shape", [("qkv", 7187, 6144, 4096), ("o_proj", 7187, 4096, 4096), ("down", 7187, 4096, 14336), ("qkv_mid", 566, 6144, 4096),
                                   ("vit_qkv", 28392, 3456, 1152), ("vit_out", 28392, 1152, 1152), ("vit_fc2", 28392, 1152, 4352)],
                         ids=lambda s: s[0])
def test_gemm_fp8_store_and_residual(ops, out_dtype, shape):
    """fp8 x fp8 -> fp32 accumulate on every production geometry the chooser picks: products of fp8 values are exact in fp32, so the
    only difference from the fp32 matmul of the dequantised operands is the summation order (and the output rounding)."""
    _, M, N, K = shape
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w8 = (torch.randn(N, K, generator=g, device=DEV) * 0.25).to(F8)
    bias = torch.randn(N, generator=g, device=DEV)
    ref = (a8.float() @ w8.float().T) * 2.0 ** -5
    out = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)

    def store():
        out.fill_(float("nan"))
        ops.gemm_fp8(a8.view(torch.uint8), w8.view(torch.uint8), out, bias=bias, scale_exp=-5)
        return out.clone()
    o = run3(store)
    e = rel_err(o, ref + bias)
    assert e <= 3 * eps(out_dtype), f"fp8 STORE {shape}: {e:.3e}"
    x0 = torch.randn(M, N, generator=g, device=DEV)
    x = x0.clone()
    ops.gemm_fp8(a8.view(torch.uint8), w8.view(torch.uint8), x, bias=bias, epilogue=_lib.EPI_RESIDUAL, scale_exp=-5)
    assert (x - (x0 + ref + bias)).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("out_dtype", DTYPES)
def test_gemm_fp8_swiglu_and_gelu(ops, out_dtype):
    from leopard_amd.weights import interleave_gate_up
    M, F, K = 7187, 14336, 4096
    g = torch.Generator(device=DEV).manual_seed(96)
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w8 = (torch.randn(2 * F, K, generator=g, device=DEV) * 0.25).to(F8)
    ref = (a8.float() @ w8.float().T) * 2.0 ** -6
    want = torch.nn.functional.silu(ref[:, :F]) * ref[:, F:]
    wi = interleave_gate_up(w8.view(torch.uint8)[:F], w8.view(torch.uint8)[F:])
    out = torch.full((M, F), float("nan"), dtype=out_dtype, device=DEV)
    ops.gemm_fp8(a8.view(torch.uint8), wi, out, epilogue=_lib.EPI_SWIGLU, scale_exp=-6)
    assert rel_err(out, want) <= 3 * eps(out_dtype)
    del ref, want, wi, w8, a8
    M, N, K = 28392, 4352, 1152
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w8 = (torch.randn(N, K, generator=g, device=DEV) * 0.25).to(F8)
    bias = torch.randn(N, generator=g, device=DEV)
    ref = (a8.float() @ w8.float().T) * 2.0 ** -4 + bias
    out = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
    ops.gemm_fp8(a8.view(torch.uint8), w8.view(torch.uint8), out, bias=bias, act=_lib.ACT_GELU_TANH, scale_exp=-4)
    assert rel_err(out, torch.nn.functional.gelu(ref, approximate="tanh")) <= 3 * eps(out_dtype)


def _fp8_close(out_u8, want_scaled):
    """fp8 outputs against torch's fp8 conversion of the fp32 result: identical bytes except where the value sits within summation
    noise of a rounding tie (then one e4m3 ulp apart)."""
    want8 = want_scaled.clamp(-448, 448).to(F8)
    got, ref = out_u8.view(F8).float(), want8.float()
    assert float((out_u8 == want8.view(torch.uint8)).float().mean()) >= 0.99
    assert bool(((got - ref).abs() <= 0.126 * ref.abs() + 2.0 ** -9).all())


def test_gemm_fp8_fp8_outputs_at_production_shapes(ops):
    """gate/up -> down and fc1 -> fc2 hand-overs: the SwiGLU / GELU epilogues write the next GEMM's fp8 operand directly."""
    from leopard_amd.weights import interleave_gate_up
    M, F, K = 7187, 14336, 4096
    g = torch.Generator(device=DEV).manual_seed(97)
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w8 = (torch.randn(2 * F, K, generator=g, device=DEV) * 0.25).to(F8)
    ref = (a8.float() @ w8.float().T) * 2.0 ** -6
    want = torch.nn.functional.silu(ref[:, :F]) * ref[:, F:] * 32.0
    wi = interleave_gate_up(w8.view(torch.uint8)[:F], w8.view(torch.uint8)[F:])
    out = torch.zeros(M, F, dtype=torch.uint8, device=DEV)

    def swiglu():
        out.zero_()
        ops.gemm_fp8(a8.view(torch.uint8), wi, out, epilogue=_lib.EPI_SWIGLU, scale_exp=-6, out_scale=32.0)
        return out.clone()
    _fp8_close(run3(swiglu), want)
    del ref, want, wi, w8, a8
    M, N, K = 28392, 4352, 1152
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w8 = (torch.randn(N, K, generator=g, device=DEV) * 0.25).to(F8)
    bias = torch.randn(N, generator=g, device=DEV)
    ref = torch.nn.functional.gelu((a8.float() @ w8.float().T) * 2.0 ** -4 + bias, approximate="tanh")
    out = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
    ops.gemm_fp8(a8.view(torch.uint8), w8.view(torch.uint8), out, bias=bias, act=_lib.ACT_GELU_TANH, scale_exp=-4, out_scale=16.0)
    _fp8_close(out, ref * 16.0)


@pytest.mark.parametrize("shape", [(7187, 4096), (28392, 1152)])
def test_norm_fp8_at_production_shapes(ops, shape):
    M, D = shape
    g = torch.Generator(device=DEV).manual_seed(98)
    x = torch.randn(M, D, generator=g, device=DEV) * 3 + 0.5
    w, b = 1 + 0.1 * torch.randn(D, generator=g, device=DEV), 0.1 * torch.randn(D, generator=g, device=DEV)
    out = torch.zeros(M, D, dtype=torch.uint8, device=DEV)
    ops.norm_fp8(x, w, b, out, 1e-6, 16.0)
    _fp8_close(out, 16.0 * torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6))
    ops.norm_fp8(x, w, None, out, 1e-5, 8.0)
    _fp8_close(out, 8.0 * (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * w))


# ---- lmi_patch_embed (fused normalise + im2col + patch conv + bias + pos-emb) at the C3 shape ----------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed_at_production_shape(ops, dtype):
    """42 ViT inputs of 364 x 364 -> 28392 patch rows x 1152: against conv2d over the processor-normalised pixels rounded to the
    operand type; u8 tiles and fp32 pixel_values bit-identical; three launches bit-identical; and against the unfused pair
    (lmi_preprocess_tiles + lmi_gemm) it replaced, timed."""
    from leopard_amd.weights import patch_weight_image_order
    n, S, P, N = 42, 364, 14, 1152
    G = S // P
    g = torch.Generator(device=DEV).manual_seed(99)
    u8 = torch.randint(0, 256, (n, S, S, 3), generator=g, device=DEV, dtype=torch.uint8)
    w = (torch.randn(N, 3, P, P, generator=g, device=DEV) * 0.05).to(dtype)
    bias, pos = torch.randn(N, generator=g, device=DEV), torch.randn(G * G, N, generator=g, device=DEV)
    wf = patch_weight_image_order(w, P)
    out = torch.empty(n * G * G, N, device=DEV)

    def fused():
        out.fill_(float("nan"))
        ops.patch_embed(u8, wf, bias, pos, out, S, P)
        return out.clone()
    o = run3(fused)
    pix = (u8.float() * (1.0 / 255.0) - 0.5) * 2.0                              # == the processor (mul, sub, mul; no contraction in eager torch)
    pix = pix.permute(0, 3, 1, 2).contiguous()
    ref = torch.nn.functional.conv2d(pix.to(dtype).float(), w.float(), bias, stride=P).flatten(2).transpose(1, 2).reshape(n * G * G, N)
    ref = ref + pos.repeat(n, 1)
    assert (o - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    o32 = torch.empty_like(o)
    ops.patch_embed(pix, wf, bias, pos, o32, S, P)
    assert torch.equal(o32, o)
    # the pair it replaced: im2col matrix in HBM + GEMM
    kp = 640
    w2 = torch.zeros(N, kp, dtype=dtype, device=DEV)
    w2[:, :588] = w.reshape(N, -1)
    patches = torch.empty(n * G * G, kp, dtype=dtype, device=DEV)
    o2 = torch.empty_like(o)

    def pair():
        ops.preprocess_tiles(u8, patches, S, P)
        ops.gemm(patches, w2, o2, bias=bias, addmat=pos, epilogue=_lib.EPI_STORE_F32)
    pair()
    assert (o2 - o).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())

    def timed(fn, reps=20):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_f, t_p = timed(lambda: ops.patch_embed(u8, wf, bias, pos, out, S, P)), timed(pair)
    print(f"[patch embed {dtype}] fused {t_f:.1f} us  ({2.0 * n * G * G * N * 588 / t_f / 1e6:.0f} TFLOP/s algorithmic)   unfused pair {t_p:.1f} us")


def test_kv_append_on_device(ops):
    g = torch.Generator(device=DEV).manual_seed(100)
    pk, pv = torch.randn(9000, 1024, generator=g, device=DEV).half(), torch.randn(9000, 1024, generator=g, device=DEV).half()
    kc, vc = torch.zeros(7400, 1024, dtype=torch.float16, device=DEV), torch.zeros(7400, 1024, dtype=torch.float16, device=DEV)
    ops.kv_append(pk[1000:8187], pv[1000:8187], kc, vc, 100)
    assert torch.equal(kc[100:7287], pk[1000:8187]) and torch.equal(vc[100:7287], pv[1000:8187])
    assert bool((kc[:100] == 0).all()) and bool((kc[7287:] == 0).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_folded_rmsnorm_over_six_decades_of_row_scale(ops, dtype):
    """The folded RMSNorm hands over T(x * gamma) WITHOUT the row scale (the consumer applies rstd to its accumulators), so in fp16 the
    operand spans the raw dynamic range of the residual stream instead of O(1) normalised values.  Rows of the stream scaled by 1e-3,
    1 and 1e3 (massive-activation rows next to tiny early-layer ones), producer (o_proj shape) -> consumer (gate/up + SwiGLU):
    the result must be as close to fp32 as the UNFUSED schedule (rmsnorm launch -> plain GEMM) on every row class — the fold may not
    cost precision anywhere in that range — and finite everywhere."""
    from leopard_amd.weights import interleave_gate_up
    M, D, F = 1536, 4096, 2048
    g = torch.Generator(device=DEV).manual_seed(191)
    att = (torch.randn(M, D, generator=g, device=DEV)).to(dtype)
    wo = (torch.randn(D, D, generator=g, device=DEV) * 0.02).to(dtype)
    wgu = (torch.randn(2 * F, D, generator=g, device=DEV) * 0.02).to(dtype)
    wi = interleave_gate_up(wgu[:F], wgu[F:])
    gamma = torch.rand(D, generator=g, device=DEV) + 0.5
    scale = torch.ones(M, 1, device=DEV)
    scale[:512] = 1e-3
    scale[1024:] = 1e3
    x0 = torch.randn(M, D, generator=g, device=DEV) * scale
    att = (att.float() * scale).to(dtype)                          # the o_proj increment follows the row's scale, as in a real stream
    eps_n = 1e-5
    # fused: producer writes T(x * gamma) + row partials, consumer scales its accumulators
    x = x0.clone()
    h = torch.empty(M, D, dtype=dtype, device=DEV)
    sq = torch.empty(M, D // 64, device=DEV)
    ops.gemm_ex(att, wo, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=gamma, rowsq_out=sq)
    fused = torch.empty(M, F, dtype=dtype, device=DEV)
    ops.gemm_ex(h, wi, fused, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=D, norm_eps=eps_n)
    # unfused: residual GEMM, rmsnorm launch, plain GEMM
    x2 = x0.clone()
    ops.gemm(att, wo, x2, epilogue=_lib.EPI_RESIDUAL)
    h2 = torch.empty(M, D, dtype=dtype, device=DEV)
    ops.rmsnorm(x2, gamma, h2, eps_n)
    unfused = torch.empty(M, F, dtype=dtype, device=DEV)
    ops.gemm(h2, wi, unfused, epilogue=_lib.EPI_SWIGLU)
    assert torch.equal(x, x2) and torch.isfinite(h.float()).all() and torch.isfinite(fused.float()).all()
    xr = x0.double() + att.double() @ wo.double().T
    hr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps_n) * gamma.double()
    lin = hr @ wgu.double().T
    ref = (torch.nn.functional.silu(lin[:, :F]) * lin[:, F:]).float()
    for name, rows in (("1e-3", slice(0, 512)), ("1", slice(512, 1024)), ("1e3", slice(1024, M))):
        ef = ((fused[rows].float() - ref[rows]).pow(2).mean().sqrt() / ref[rows].pow(2).mean().sqrt()).item()
        eu = ((unfused[rows].float() - ref[rows]).pow(2).mean().sqrt() / ref[rows].pow(2).mean().sqrt()).item()
        print(f"[folded norm, {dtype}, row scale {name}] rel RMS vs fp64: fused {ef:.3e}, unfused {eu:.3e}")
        assert ef <= 1.5 * eu + 1e-6, (name, ef, eu)


def _fp8_dequant(u8):
    return u8.view(F8).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [("llama", 7187, 32, 8, 128, True), ("siglip", 42 * 676, 16, 16, 72, False)], ids=lambda s: s[0])
def test_attention_fp8_output_vs_attention_then_quantize(ops, dtype, shape):
    """lmi_attn_varlen_fwd_fp8 (the fp8 schedule's o_proj operand written by the attention epilogue) vs the 16-bit attention followed by
    lmi_quantize_fp8: the same values up to the order of the two roundings — every byte within one e4m3 step, >= 90 % identical."""
    _, S, H, KV, hd, causal = shape
    g = torch.Generator(device=DEV).manual_seed(311)
    qkv = (torch.randn(S, (H + 2 * KV) * hd, generator=g, device=DEV)).to(dtype)
    qw, kw = H * hd, KV * hd
    if causal:
        cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
        mx = S
    else:
        cu = torch.arange(0, S + 1, 676, dtype=torch.int32, device=DEV)
        mx = 676
    scale_out = 64.0
    att = torch.empty(S, qw, dtype=dtype, device=DEV)
    ops.attention(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], att, cu, cu, mx, H, KV, hd, hd ** -0.5, causal, True)
    want = torch.zeros(S, qw, dtype=torch.uint8, device=DEV)
    ops.quantize_fp8(att, want, scale_out)
    got = torch.full((S, qw), 0x7F, dtype=torch.uint8, device=DEV)
    ops.attention_fp8out(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], got, scale_out, cu, cu, mx, H, KV, hd, hd ** -0.5, causal)
    torch.cuda.synchronize()
    a, b = _fp8_dequant(got), _fp8_dequant(want)
    assert torch.isfinite(a).all()
    step = torch.maximum(b.abs(), torch.full_like(b, 2.0 ** -6)) * 2.0 ** -3          # one e4m3 mantissa step at that magnitude
    assert bool(((a - b).abs() <= step * 1.001).all())
    assert float((got == want).float().mean()) >= 0.90


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [7187, 566])
def test_rope_qkv_fp8_vs_gemm_fp8_then_rope(ops, dtype, M):
    """lmi_rope_qkv_fp8 (fp8 q|k|v GEMM with RoPE + KV append in the epilogue, rope-permuted weight rows) vs lmi_gemm_fp8 followed by
    lmi_rope_qk: q / k rotated from the fp32 accumulators (one rounding) vs from the rounded 16-bit values (two) — equal within two
    roundings of the output type; v and the V cache bit-identical."""
    from leopard_amd.weights import rope_permute_rows
    H, KV, hd, K = 32, 8, 128, 4096
    N = (H + 2 * KV) * hd
    g = torch.Generator(device=DEV).manual_seed(313 + M)
    a8 = torch.randn(M, K, generator=g, device=DEV).to(F8)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.25).to(F8)
    w_u8 = w.view(torch.uint8)
    w_rope = torch.cat([rope_permute_rows(w_u8[:(H + KV) * hd]), w_u8[(H + KV) * hd:]], dim=0).contiguous()
    f = torch.arange(M, device=DEV).float().reshape(-1, 1) * (1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    ref = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.gemm_fp8(a8.view(torch.uint8), w.view(torch.uint8), ref, scale_exp=-6)
    kc0, vc0 = torch.zeros(M + 8, KV * hd, dtype=dtype, device=DEV), torch.zeros(M + 8, KV * hd, dtype=dtype, device=DEV)
    ops.rope_qk(ref, H, KV, hd, cos, sin, kc0, vc0, 3)
    got = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    kc, vc = torch.zeros_like(kc0), torch.zeros_like(vc0)
    ops.rope_qkv_fp8(a8.view(torch.uint8), w_rope, got, -6, cos, sin, kc, vc, 3, H, KV, hd)
    torch.cuda.synchronize()
    qk = (H + KV) * hd
    assert torch.equal(got[:, qk:], ref[:, qk:]) and torch.equal(vc, vc0)
    e = rel_err(got[:, :qk], ref[:, :qk].float())
    assert e <= 3 * eps(dtype), e
    assert torch.equal(kc[3:3 + M], got[:, H * hd:qk]) and bool((kc[:3] == 0).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_mid_m_complete_tile_m312(ops, dtype):
    """Idefics2's text side (S = 312): gate/up + SwiGLU with the folded RMSNorm row scale on the M-complete 384 x 128 geometry the chooser
    picks for 256 < M <= 384 and wide N, vs fp32 and vs the 64 x 128 geometry (gemm.mid_m = 0): same rows, same row scales (bit-identical
    rstd by construction), results equal within the output rounding."""
    from leopard_amd.weights import interleave_gate_up
    M, F, K = 312, 14336, 4096
    a, w, _, ref = operands(M, 2 * F, K, dtype, 0.02)
    wi = interleave_gate_up(w[:F], w[F:])
    g = torch.Generator(device=DEV).manual_seed(93)
    sq = (torch.rand(M, K // 64, generator=g, device=DEV) + 0.5) * 64
    rstd = torch.rsqrt(sq.sum(-1, keepdim=True) / K + 1e-5)
    want = torch.nn.functional.silu(ref[:, :F] * rstd) * (ref[:, F:] * rstd)
    outs = {}
    try:
        for mid in (1, 0):
            ops.set_option("gemm.mid_m", mid)
            out = torch.full((M, F), float("nan"), dtype=dtype, device=DEV)
            ops.gemm_ex(a, wi, out, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
            outs[mid] = out
            e = rel_err(out, want)
            assert e <= 3 * eps(dtype), f"mid_m {mid}: {e:.3e}"
    finally:
        ops.set_option("gemm.mid_m", 1)
    assert rel_err(outs[1], outs[0].float()) <= 2 * eps(dtype)


# ---- one copy of the LLM weights: the prefill GEMM reading the packed (decode) order ------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 5, 7, 10])
def test_gemm_packed_weights_bit_identical_at_llama_shapes(ops, dtype, cfg):
    """ldw = LMI_LDW_PACKED(K) at the C3 / C2 Llama shapes on every tile geometry: residual + folded-norm producer (o_proj, down_proj),
    SwiGLU + folded-norm consumer (gate/up), q|k|v + RoPE + KV append — every output bit equals the row-major call's (the W image is
    staged chunk-major from the packed order: same fragments, same MFMA order), three launches each."""
    from leopard_amd import _lib
    from leopard_amd.weights import as_packed, interleave_gate_up, rope_permute_rows
    ops.set_option("gemm.config", cfg)
    try:
        for M, N, K in ((7187, 4096, 4096), (1242, 4096, 14336)):
            a, w, _, _ = operands(M, N, K, dtype, 0.02)
            wp = as_packed(w)
            g = torch.Generator(device=DEV).manual_seed(5)
            x0 = torch.randn(M, N, generator=g, device=DEV)
            gam = torch.randn(N, generator=g, device=DEV)
            def resid(wt):
                x, h, sq = x0.clone(), torch.zeros(M, N, dtype=dtype, device=DEV), torch.zeros(M, N // 64, device=DEV)
                ops.gemm_ex(a, wt, x, epilogue=_lib.EPI_RESIDUAL, norm_out=h, norm_gamma=gam, rowsq_out=sq)
                return torch.cat([x, h.float(), sq], 1)
            assert torch.equal(run3(lambda: resid(w)), run3(lambda: resid(wp))), f"residual {M}x{N}x{K} cfg {cfg}"
            del wp
        M, F, K = 7187, 14336, 4096
        g = torch.Generator(device=DEV).manual_seed(6)
        a = torch.randn(M, K, generator=g, device=DEV).to(dtype)
        wi = interleave_gate_up((torch.randn(F, K, generator=g, device=DEV) * 0.02).to(dtype), (torch.randn(F, K, generator=g, device=DEV) * 0.02).to(dtype))
        sq = (torch.rand(M, K // 64, generator=g, device=DEV) + 0.5) * 64
        wip = as_packed(wi)
        def gate_up(wt):
            out = torch.full((M, F), float("nan"), dtype=dtype, device=DEV)
            return ops.gemm_ex(a, wt, out, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
        assert torch.equal(run3(lambda: gate_up(wi)), run3(lambda: gate_up(wip))), f"gate/up cfg {cfg}"
        del wi, wip
        nq, nkv, D = 32, 8, 128
        N = (nq + 2 * nkv) * D
        w = (torch.randn(N, K, generator=g, device=DEV) * 0.02).to(dtype)
        wr = torch.cat([rope_permute_rows(w[:(nq + nkv) * D]), w[(nq + nkv) * D:]], 0).contiguous()
        wrp = as_packed(wr)
        ang = torch.rand(M, D // 2, generator=g, device=DEV) * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        def qkv(wt):
            out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
            kc, vc = torch.zeros(M, nkv * D, dtype=dtype, device=DEV), torch.zeros(M, nkv * D, dtype=dtype, device=DEV)
            ops.rmsnorm_rope(a, wt, out, sq, 1e-5, cos, sin, kc, vc, 0, nq, nkv, D)
            return torch.cat([out, kc, vc], 1)
        assert torch.equal(run3(lambda: qkv(wr)), run3(lambda: qkv(wrp))), f"q|k|v + RoPE cfg {cfg}"
    finally:
        ops.set_option("gemm.config", -1)
