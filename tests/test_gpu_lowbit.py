"""-m gpu: the LOW-BIT CORRECTION PHASE on the device (csrc/lowbit.h, the LO4 instantiations of csrc/gemm.h, through the C ABI).

tests/test_emu_lowbit.py pins the LOGIC on the host emulator (images == the oracle's quantisation rule bit for bit; tile indexing; epilogues).
Here the hardware half: what v_mfma_scale_f32_32x32x64_f8f6f4 does with the fp4 images and their E8M0 block scales (nibble order, lane ->
(row, k-block) ownership of a scale byte, op_sel) — every LO4 GEMM is compared with a plain fp32 product over the DEQUANTISED images, at small
shapes on every geometry and at the production shapes of the C3 step (7187 / 28392 rows), three launches each bit-identical (a DMA / read
race would show) —, the producers against their host definitions, and the engine's lo4 mode at full depth in tests/test_gpu_parity.py."""
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.ops import Lo4Act, Lo4Weight, lo4_k4
from leopard_amd.weights import as_packed, interleave_gate_up, rope_permute_rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from leopard_amd.ops import Ops
    o = Ops()
    yield o
    o.set_option("gemm.config", -1)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def decode_img(img, sc, K, per_row=False):
    """fp4 image [M, K4 / 2] bytes + E8M0 scales -> fp32 [M, K]: element k in nibble k & 1 of byte k >> 1 (the layout the kernels assume)."""
    grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=img.device)
    M, half = img.shape
    b = img.to(torch.int64)
    codes = torch.stack([b & 15, b >> 4], dim=-1).reshape(M, half * 2)
    val = grid[codes & 7] * torch.where((codes & 8) != 0, -1.0, 1.0)
    s = torch.exp2(sc.to(torch.float32) - 127.0)
    s = s[:, None].expand(M, half * 2) if per_row else s[:, :half * 2 // 32].repeat_interleave(32, dim=1)
    return (val * s)[:, :K]


def host_lo_round(x, block):
    """The oracle's MX e2m1 rule (oracle.leopard_oracle._lo_round) — imported: this file is test code."""
    from oracle import leopard_oracle as O
    return O._lo_round(x.float().cpu(), "e2m1", block)


def eps(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7


def act_from(ops, x, dtype):
    act = Lo4Act.empty(x.shape[0], x.shape[1], dtype, DEV)
    ops.split_lo4(x.contiguous(), act)
    return act


def ref_acc(act, w_rowmajor, w4):
    K = act.K
    return act.hi.float() @ w_rowmajor.float().T + decode_img(act.img, act.sc, K) @ decode_img(w4.img, w4.sc, K, per_row=True).T


@pytest.mark.parametrize("dtype", DTYPES)
def test_producers_match_the_host_rule(ops, dtype):
    M, K = 333, 1152
    x = rnd((M, K), torch.float32, 1, 3.0)
    act = Lo4Act.empty(M, K, dtype, DEV)
    act.img.fill_(0xAB); act.sc.fill_(0xCD)
    ops.split_lo4(x, act)
    assert torch.equal(act.hi, x.to(dtype))
    lo = (x - act.hi.float())
    assert torch.equal(decode_img(act.img, act.sc, K).cpu(), host_lo_round(lo, 32))
    assert act.img[:, K // 2:].abs().max() == 0 and act.sc[:, K // 32:].abs().max() == 0
    w = rnd((300, 4096), dtype, 2, 0.02)
    w4 = ops.quantize_w4(w)
    assert torch.equal(decode_img(w4.img, w4.sc, 4096, per_row=True).cpu(), host_lo_round(w, 0))
    # norms: the 16-bit rows are those of the plain kernels, the image removes ~85 % of their rounding
    for rms, D in ((False, 1152), (True, 4096)):
        xs = rnd((777, D), torch.float32, 3, 2.0) + 0.3
        g = torch.rand(D, generator=torch.Generator().manual_seed(4)).to(DEV) + 0.5
        b = None if rms else rnd((D,), torch.float32, 5, 0.2)
        a = Lo4Act.empty(777, D, dtype, DEV)
        ops.norm_lo4(xs, g, b, a, 1e-5)
        plain = torch.empty(777, D, dtype=dtype, device=DEV)
        (ops.rmsnorm(xs, g, plain, 1e-5) if rms else ops.layernorm(xs, g, b, plain, 1e-5))
        assert torch.equal(a.hi, plain)
        y = (xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-5) * g) if rms else torch.nn.functional.layer_norm(xs, (D,), g, b, 1e-5)
        before = (y - a.hi.float()).pow(2).mean().sqrt().item()
        after = (y - a.hi.float() - decode_img(a.img, a.sc, D)).pow(2).mean().sqrt().item()
        assert after < 0.25 * before, (rms, before, after)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 0, 2, 5, 8, 10])
def test_gemm_lo4_small_every_geometry(ops, dtype, cfg):
    M, N, K = 300, 256, 320
    x, w = rnd((M, K), torch.float32, 10, 2.0), rnd((N, K), dtype, 11, 0.1)
    bias = rnd((N,), torch.float32, 12)
    act, w4 = act_from(ops, x, dtype), ops.quantize_w4(w)
    ref = ref_acc(act, w, w4) + bias
    ops.set_option("gemm.config", cfg)
    try:
        outs = []
        for _ in range(3):
            o = torch.empty(M, N, dtype=torch.float32, device=DEV)
            ops.gemm_lo4(act, w, w4, o, bias=bias, epilogue=_lib.EPI_STORE_F32)
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert (outs[0] - ref).abs().max() <= 2e-4 * ref.abs().max(), ((outs[0] - ref).abs().max().item(), ref.abs().max().item())
        exact = x @ w.float().T + bias
        plain = torch.empty(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(act.hi, w, plain, bias=bias, epilogue=_lib.EPI_STORE_F32)
        assert (outs[0] - exact).pow(2).mean().sqrt().item() < 0.3 * (plain - exact).pow(2).mean().sqrt().item()
        # GELU + the image of its own output
        out4 = Lo4Act.empty(M, N, dtype, DEV)
        ops.gemm_lo4(act, w, w4, out4.hi, bias=bias, act=_lib.ACT_GELU_TANH, out4=out4)
        y = torch.nn.functional.gelu(ref, approximate="tanh")
        assert ((out4.hi.float() - y).abs() / (1 + y.abs())).max().item() <= 2 * eps(dtype)
        before = (y - out4.hi.float()).pow(2).mean().sqrt().item()
        after = (y - out4.hi.float() - decode_img(out4.img, out4.sc, N)).pow(2).mean().sqrt().item()
        assert after < 0.3 * before, (before, after)
    finally:
        ops.set_option("gemm.config", -1)


PROD = [  # (name, M, N, K, epilogue) — the lo4 schedule's GEMMs at the C3 sizes
    ("siglip qkv", 28392, 3456, 1152, "store_bias"),
    ("siglip out_proj", 28392, 1152, 1152, "resid_bias"),
    ("siglip fc1", 28392, 4352, 1152, "gelu_out4"),
    ("siglip fc2", 28392, 1152, 4352, "resid_bias"),
    ("llama o_proj", 7187, 4096, 4096, "producer"),
    ("llama gate/up", 7187, 28672, 4096, "swiglu_out4"),
    ("llama down", 7187, 4096, 14336, "producer"),
]


@pytest.mark.parametrize("name,M,N,K,kind", PROD, ids=[p[0].replace(" ", "_").replace("/", "_") for p in PROD])
def test_gemm_lo4_production_shapes(ops, name, M, N, K, kind):
    """fp16, the automatic geometry, packed 16-bit weights where the engine packs them; sampled rows (every tile edge) against fp32."""
    dtype = torch.float16
    x = rnd((M, K), torch.float32, 20, 1.5)
    w = rnd((N, K), dtype, 21, 0.03)
    act, w4 = act_from(ops, x, dtype), ops.quantize_w4(w)
    w_run = as_packed(w) if name.startswith("llama") else w
    rows = torch.unique(torch.cat([torch.arange(0, M, 997), torch.tensor([0, 1, 255, 256, 257, M - 257, M - 256, M - 2, M - 1])]).clamp(0, M - 1)).to(DEV)
    sub = Lo4Act(act.hi[rows].contiguous(), act.img[rows].contiguous(), act.sc[rows].contiguous())
    acc = ref_acc(sub, w, w4)
    bias = rnd((N,), torch.float32, 22, 0.5) if "bias" in kind or "gelu" in kind else None

    def run():
        if kind == "store_bias":
            o = torch.empty(M, N, dtype=dtype, device=DEV)
            ops.gemm_lo4(act, w_run, w4, o, bias=bias)
            return (o,)
        if kind == "resid_bias":
            xs = torch.ones(M, N, dtype=torch.float32, device=DEV)
            ops.gemm_lo4(act, w_run, w4, xs, bias=bias, epilogue=_lib.EPI_RESIDUAL)
            return (xs,)
        if kind == "gelu_out4":
            o4 = Lo4Act.empty(M, N, dtype, DEV)
            ops.gemm_lo4(act, w_run, w4, o4.hi, bias=bias, act=_lib.ACT_GELU_TANH, out4=o4)
            return (o4.hi, o4.img, o4.sc)
        if kind == "producer":
            xs = torch.ones(M, N, dtype=torch.float32, device=DEV)
            h = Lo4Act.empty(M, N, dtype, DEV)
            sq = torch.empty(M, N // 64, dtype=torch.float32, device=DEV)
            ops.gemm_lo4(act, w_run, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq, out4=h)
            return (xs, h.hi, h.img, h.sc, sq)
        o4 = Lo4Act.empty(M, N // 2, dtype, DEV)
        ops.gemm_lo4(act, w_run, w4, o4.hi, epilogue=_lib.EPI_SWIGLU, out4=o4)
        return (o4.hi, o4.img, o4.sc)

    gamma = (torch.rand(N, generator=torch.Generator().manual_seed(23)) + 0.5).to(DEV)
    a, b, c = run(), run(), run()
    for t0, t1, t2 in zip(a, b, c):
        assert torch.equal(t0, t1) and torch.equal(t0, t2), name
    scale = acc.abs().max().item()
    if kind == "store_bias":
        y = acc + bias
        assert ((a[0][rows].float() - y).abs() / (1 + y.abs())).max().item() <= 2 * eps(dtype)
    elif kind == "resid_bias":
        assert (a[0][rows] - (1.0 + acc + bias)).abs().max().item() <= 3e-4 * scale
    elif kind == "gelu_out4":
        y = torch.nn.functional.gelu(acc + bias, approximate="tanh")
        assert ((a[0][rows].float() - y).abs() / (1 + y.abs())).max().item() <= 2 * eps(dtype)
        before = (y - a[0][rows].float()).pow(2).mean().sqrt().item()
        after = (y - a[0][rows].float() - decode_img(a[1][rows], a[2][rows], N)).pow(2).mean().sqrt().item()
        assert after < 0.3 * before, (before, after)
    elif kind == "producer":
        xs = a[0][rows]
        assert (xs - (1.0 + acc)).abs().max().item() <= 3e-4 * scale
        hx = xs * gamma
        hi = a[1][rows].float()
        assert ((hi - hx).abs() <= eps(dtype) * hx.abs() + 2.0 ** -24).all()               # one rounding of x * gamma (the device may fuse the product into the conversion)
        got, want = decode_img(a[2][rows], a[3][rows], N).cpu(), host_lo_round(hx - hi, 32)
        assert (got != want).float().mean().item() < 2e-3                                    # the same rule; a fused multiply-subtract moves a few ties
        before, after = (hx - hi).pow(2).mean().sqrt().item(), (hx - hi - got.to(DEV)).pow(2).mean().sqrt().item()
        assert after < 0.25 * before, (before, after)
    else:
        g = acc.view(-1, N // 64, 2, 32)
        y = (torch.nn.functional.silu(g[:, :, 0]) * g[:, :, 1]).reshape(-1, N // 2)
        assert ((a[0][rows].float() - y).abs() / (1 + y.abs())).max().item() <= 3 * eps(dtype)
        before = (y - a[0][rows].float()).pow(2).mean().sqrt().item()
        after = (y - a[0][rows].float() - decode_img(a[1][rows], a[2][rows], N // 2)).pow(2).mean().sqrt().item()
        assert after < 0.3 * before, (before, after)


@pytest.mark.parametrize("kind", ["gelu_out4", "producer", "swiglu_out4"])
def test_gemm_lo4_rows_and_their_images_do_not_depend_on_their_position(ops, kind):
    """The 16-bit result, the residual image and its block scales of a row are the same bits wherever the row sits in the launch and whichever tile
    geometry handles it (M = 300 / 729 / 3000): hipcc contracts `a * b - T(a * b)` into a fused multiply-subtract or not per unrolled copy of an
    epilogue unless the product is individually rounded — which would move image ties with the row's position (the packed == separate
    bit-identities of the lo4 engine rest on this; the fast-schedule twin is tests/test_gpu_kernels.py)."""
    dtype = torch.float16
    K, N = 1152, (2176 * 2 if kind == "swiglu_out4" else 2304)
    x = rnd((3100, K), torch.float32, 40, 1.5)
    w = rnd((N, K), dtype, 41, 0.03)
    if kind == "swiglu_out4":
        w = interleave_gate_up(w[:N // 2].contiguous(), w[N // 2:].contiguous())
    w4 = ops.quantize_w4(w)
    bias = rnd((N,), torch.float32, 42, 0.5)
    gamma = (torch.rand(N, generator=torch.Generator().manual_seed(43)) + 0.5).to(DEV)

    def run(lo, hi):
        act = act_from(ops, x[lo:hi], dtype)
        M = hi - lo
        if kind == "gelu_out4":
            o4 = Lo4Act.empty(M, N, dtype, DEV)
            ops.gemm_lo4(act, w, w4, o4.hi, bias=bias, act=_lib.ACT_GELU_TANH, out4=o4)
            return (o4.hi, o4.img, o4.sc)
        if kind == "producer":
            xs = torch.ones(M, N, dtype=torch.float32, device=DEV)
            h = Lo4Act.empty(M, N, dtype, DEV)
            sq = torch.empty(M, N // 64, dtype=torch.float32, device=DEV)
            ops.gemm_lo4(act, w, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq, out4=h)
            return (xs, h.hi, h.img, h.sc, sq)
        o4 = Lo4Act.empty(M, N // 2, dtype, DEV)
        ops.gemm_lo4(act, w, w4, o4.hi, epilogue=_lib.EPI_SWIGLU, out4=o4)
        return (o4.hi, o4.img, o4.sc)

    base = run(0, 3000)
    for sh in (1, 7, 32, 33, 64, 100):
        for t, b in zip(run(sh, sh + 3000), base):
            assert torch.equal(t[:3000 - sh], b[sh:]), f"{kind}: rows shifted by {sh} differ"
    for lo, hi in ((0, 300), (41, 341), (0, 729), (729, 1458), (100, 164)):
        for t, b in zip(run(lo, hi), base):
            assert torch.equal(t, b[lo:hi]), f"{kind}: rows {lo}:{hi} alone differ from the same rows inside M = 3000"


def test_gemm_lo4_small_m_ring_at_the_llama_gate_up_shape(ops):
    """The 64 x 128 ring (M < 512: BASELINE config C1, S = 228) and the M-complete 384 x 128 ring (256 < M <= 384: Idefics2's text side) through
    16 fp4 k-tiles at N = 28672 with the folded RMSNorm, against the same rows inside M = 3225 on the staggered 256 x 256 schedule, twice each.
    Round 5 regression: the waves of those two geometries that have no scale piece to fetch issued one VMEM operation fewer per fp4 k-tile than
    the ring's counted vmcnt wait assumes and could read a slot before their own last operand pieces had landed (a few rows off by an ulp, image
    bytes wrong, different from run to run)."""
    dtype = torch.float16
    M, F, K = 3225, 14336, 4096
    x = rnd((M, K), torch.float32, 50, 1.5)
    w = interleave_gate_up(rnd((F, K), dtype, 51, 0.02), rnd((F, K), dtype, 52, 0.02))
    w4, wp = ops.quantize_w4(w), as_packed(w)
    sq = (torch.rand(M, K // 64, generator=torch.Generator().manual_seed(53)) * 64 + 32).to(DEV)

    def run(lo, hi):
        act = act_from(ops, x[lo:hi], dtype)
        o4 = Lo4Act.empty(hi - lo, F, dtype, DEV)
        ops.gemm_lo4(act, wp, w4, o4.hi, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq[lo:hi].contiguous(), norm_dim=K, norm_eps=1e-5, out4=o4)
        return (o4.hi, o4.img, o4.sc)
    base = run(0, M)
    for lo, hi in ((2997, 3225), (0, 228), (100, 412), (566, 878), (0, 566)):
        for rep in range(2):
            for t, b in zip(run(lo, hi), base):
                assert torch.equal(t, b[lo:hi]), f"rows {lo}:{hi} alone (M = {hi - lo}) differ from the same rows inside M = {M} (repetition {rep})"


def test_rmsnorm_rope_lo4_at_the_llama_shape(ops):
    dtype = torch.float16
    S, nq, nkv, D, K = 7187, 32, 8, 128, 4096
    xa = rnd((S, K), torch.float32, 30, 1.5)
    wq, wk, wv = rnd((nq * D, K), dtype, 31, 0.03), rnd((nkv * D, K), dtype, 32, 0.03), rnd((nkv * D, K), dtype, 33, 0.03)
    w_nat = torch.cat([wq, wk, wv], 0)
    w_rope = torch.cat([rope_permute_rows(torch.cat([wq, wk], 0)), wv], 0).contiguous()
    act = act_from(ops, xa, dtype)
    w4_rope, w4_nat = ops.quantize_w4(w_rope), ops.quantize_w4(w_nat)
    pos = torch.arange(S).float()
    inv = 1.0 / (5e5 ** (torch.arange(0, D, 2).float() / D))
    cos, sin = (pos[:, None] * inv[None]).cos().contiguous().to(DEV), (pos[:, None] * inv[None]).sin().contiguous().to(DEV)
    sq = (torch.rand(S, K // 64, generator=torch.Generator().manual_seed(34)) * 64 + 32).to(DEV)
    rstd = torch.rsqrt(sq.sum(-1, keepdim=True) / K + 1e-5)
    qkv = torch.empty(S, (nq + 2 * nkv) * D, dtype=dtype, device=DEV)
    kc, vc = torch.zeros(S, nkv * D, dtype=dtype, device=DEV), torch.zeros(S, nkv * D, dtype=dtype, device=DEV)
    ops.rmsnorm_rope_lo4(act, as_packed(w_rope), w4_rope, qkv, sq, 1e-5, cos, sin, kc, vc, 0, nq, nkv, D)
    rows = torch.cat([torch.arange(0, S, 499), torch.tensor([255, 256, S - 1])]).to(DEV)
    sub = Lo4Act(act.hi[rows].contiguous(), act.img[rows].contiguous(), act.sc[rows].contiguous())
    acc = (ref_acc(sub, w_nat, w4_nat) * rstd[rows]).view(len(rows), nq + 2 * nkv, D)
    h = D // 2
    c, s = torch.cat([cos[rows], cos[rows]], -1)[:, None, :], torch.cat([sin[rows], sin[rows]], -1)[:, None, :]
    qk = acc[:, :nq + nkv]
    rot = torch.cat((-qk[..., h:], qk[..., :h]), dim=-1)
    ref = torch.cat([qk * c + rot * s, acc[:, nq + nkv:]], dim=1).reshape(len(rows), -1)
    assert ((qkv[rows].float() - ref).abs() / (1 + ref.abs())).max().item() <= 2 * eps(dtype)
    assert torch.equal(kc, qkv[:, nq * D:(nq + nkv) * D]) and torch.equal(vc, qkv[:, (nq + nkv) * D:])


@pytest.mark.parametrize("name,lens,H,KV,hd,causal", [("llama", [7187], 32, 8, 128, True), ("siglip", [676] * 42, 16, 16, 72, False)])
def test_attention_lo4_at_the_production_shapes(ops, name, lens, H, KV, hd, causal):
    """lmi_attn_varlen_fwd_lo4 at the C3 shapes: the 16-bit rows == lmi_attn_varlen_fwd bit for bit, the image == the host rule on
    (fp32 output - 16-bit rows) in the per-head padded order (up to a few ties moved by fused multiply-subtracts), three launches bit-identical."""
    from leopard_amd.ops import lo4_head_k4
    dtype = torch.float16
    S = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    qkv = rnd((S, (H + 2 * KV) * hd), dtype, 70)
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:(H + KV) * hd], qkv[:, (H + KV) * hd:]
    plain = torch.empty(S, H * hd, dtype=dtype, device=DEV)
    ops.attention(q, k, v, plain, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
    o32 = torch.empty(S, H * hd, dtype=torch.float32, device=DEV)
    ops.attention_f32out(q, k, v, o32, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
    acts = []
    for _ in range(3):
        act = Lo4Act.empty(S, H * hd, dtype, DEV, k4=lo4_head_k4(H, hd))
        ops.attention_lo4(q, k, v, act, cu, cu, max(lens), H, KV, hd, hd ** -0.5, causal)
        acts.append(act)
    assert all(torch.equal(acts[0].hi, a.hi) and torch.equal(acts[0].img, a.img) and torch.equal(acts[0].sc, a.sc) for a in acts[1:])
    act = acts[0]
    assert torch.equal(act.hi, plain)
    nb = (hd + 31) // 32 * 32
    rows = torch.arange(0, S, 37, device=DEV)
    lo = torch.zeros(len(rows), H, nb, device=DEV)
    lo[:, :, :hd] = (o32[rows] - plain[rows].float()).view(len(rows), H, hd)
    want = host_lo_round(lo.view(len(rows), H * nb), 32)
    got = decode_img(act.img[rows], act.sc[rows], H * nb).cpu()
    assert (got != want).float().mean().item() < 2e-3
    assert ((lo.view(len(rows), -1).cpu() - got).pow(2).mean().sqrt() / lo.pow(2).mean().sqrt().cpu()).item() < 0.25
    if H * nb < act.K4:
        assert act.img[:, H * nb // 2:].abs().max() == 0


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
    return row.to(DEV), unit.view(-1, 64).max(dim=1).values.contiguous().to(DEV), _ranges(row)


def _select(act, sel):
    act.row_sel, act.unit_sel, act.sel_ranges = sel
    keep = sel[0].bool()
    act.img[~keep] = 0
    act.sc[~keep] = 0
    return act


@pytest.mark.parametrize("name,N,K,kind", [("gate/up", 28672, 4096, "swiglu"), ("down", 4096, 14336, "producer"), ("o_proj", 4096, 4096, "producer")])
def test_gemm_lo4_row_selection_at_the_llama_shapes(ops, name, N, K, kind):
    """M = 7187 (the C3 sequence) with the correction on the last 256 rows — rows 6931 .. 7186: the tail of row tile 27 and the 19-row tile 28.
    Selected rows: the bits of the every-row lo4 launch (results AND the image of the output); every other row: the bits of the fast launch; the
    image of an unselected row is never written; three launches bit-identical."""
    dtype = torch.float16
    M = 7187
    x = rnd((M, K), torch.float32, 90, 2.0)
    w = rnd((N, K), dtype, 91, 0.05)
    if kind == "swiglu":
        w = interleave_gate_up(w[:N // 2].contiguous(), w[N // 2:].contiguous())
    w_run = as_packed(w)
    w4 = ops.quantize_w4(w)
    sel = _selection(M, range(M - 256, M))
    keep = sel[0].bool()
    full, part = act_from(ops, x, dtype), _select(act_from(ops, x, dtype), sel)
    gamma = (torch.rand(N, generator=torch.Generator().manual_seed(92)) + 0.5).to(DEV)

    def run(mode):
        a = {"full": full, "sel": part, "fast": full}[mode]
        s = sel if mode == "sel" else None
        if kind == "swiglu":
            o = Lo4Act.empty(M, N // 2, dtype, DEV, sel=s)
            if mode == "fast":
                ops.gemm_ex(a.hi, w_run, o.hi, epilogue=_lib.EPI_SWIGLU)
            else:
                ops.gemm_lo4(a, w_run, w4, o.hi, epilogue=_lib.EPI_SWIGLU, out4=o)
            return (o.hi, o.img, o.sc)
        xs = torch.ones(M, N, dtype=torch.float32, device=DEV)
        h = Lo4Act.empty(M, N, dtype, DEV, sel=s)
        sq = torch.empty(M, N // 64, dtype=torch.float32, device=DEV)
        if mode == "fast":
            ops.gemm_ex(a.hi, w_run, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq)
        else:
            ops.gemm_lo4(a, w_run, w4, xs, epilogue=_lib.EPI_RESIDUAL, norm_out=h.hi, norm_gamma=gamma, rowsq_out=sq, out4=h)
        return (xs, sq, h.hi, h.img, h.sc)

    f, l, s1, s2, s3 = run("fast"), run("full"), run("sel"), run("sel"), run("sel")
    for t1, t2, t3 in zip(s1, s2, s3):
        assert torch.equal(t1, t2) and torch.equal(t1, t3), name
    n_val = len(s1) - 2                                                # the value tensors; the last two are the image and its scales
    for i in range(n_val):
        assert torch.equal(s1[i][keep], l[i][keep]) and torch.equal(s1[i][~keep], f[i][~keep]), (name, i)
    assert not torch.equal(l[0][keep], f[0][keep])
    assert torch.equal(s1[-2][keep], l[-2][keep]) and torch.equal(s1[-1][keep], l[-1][keep])
    assert s1[-2][~keep].abs().max().item() == 0 and s1[-1][~keep].abs().max().item() == 0


def test_norm_attention_and_qkv_lo4_row_selection_at_the_llama_shape(ops):
    dtype = torch.float16
    S, D, H, KV, hd = 7187, 4096, 32, 8, 128
    sel = _selection(S, range(S - 256, S))
    keep = sel[0].bool()
    x = rnd((S, D), torch.float32, 95, 2.0)
    g = (torch.rand(D, generator=torch.Generator().manual_seed(96)) + 0.5).to(DEV)
    full, part = Lo4Act.empty(S, D, dtype, DEV), Lo4Act.empty(S, D, dtype, DEV, sel=sel)
    ops.norm_lo4(x, g, None, full, 1e-5)
    ops.norm_lo4(x, g, None, part, 1e-5)
    assert torch.equal(part.hi, full.hi) and torch.equal(part.img[keep], full.img[keep]) and torch.equal(part.sc[keep], full.sc[keep])
    assert part.img[~keep].abs().max().item() == 0 and part.sc[~keep].abs().max().item() == 0
    # q|k|v + RoPE + KV append consuming the selected operand: selected rows == every-row launch, the others == the fast launch
    w = rnd(((H + 2 * KV) * hd, D), dtype, 97, 0.05)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd], hd), w[(H + KV) * hd:]], 0).contiguous()
    w4 = ops.quantize_w4(w_rope)
    pos = torch.arange(S, device=DEV).float()
    inv = 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))
    cos, sin = (pos[:, None] * inv[None]).cos().contiguous(), (pos[:, None] * inv[None]).sin().contiguous()
    outs = []
    for mode in ("fast", "full", "sel"):
        qkv = torch.empty(S, (H + 2 * KV) * hd, dtype=dtype, device=DEV)
        kc, vc = torch.zeros(S, KV * hd, dtype=dtype, device=DEV), torch.zeros(S, KV * hd, dtype=dtype, device=DEV)
        if mode == "fast":
            ops.rmsnorm_rope(full.hi, as_packed(w_rope), qkv, None, 1e-5, cos, sin, kc, vc, 0, H, KV, hd)
        else:
            ops.rmsnorm_rope_lo4(full if mode == "full" else part, as_packed(w_rope), w4, qkv, None, 1e-5, cos, sin, kc, vc, 0, H, KV, hd)
        outs.append((qkv, kc, vc))
    for i in range(3):
        assert torch.equal(outs[2][i][keep], outs[1][i][keep]) and torch.equal(outs[2][i][~keep], outs[0][i][~keep])
    # attention writing the image of selected rows only
    from leopard_amd.ops import lo4_head_k4
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    qkv = outs[0][0]
    qw, kw = H * hd, KV * hd
    k4 = lo4_head_k4(H, hd)
    a_full, a_part = Lo4Act.empty(S, qw, dtype, DEV, k4=k4), Lo4Act.empty(S, qw, dtype, DEV, k4=k4, sel=sel)
    for a in (a_full, a_part):
        ops.attention_lo4(qkv[:, :qw], qkv[:, qw:qw + kw], qkv[:, qw + kw:], a, cu, cu, S, H, KV, hd, hd ** -0.5, True)
    assert torch.equal(a_part.hi, a_full.hi) and torch.equal(a_part.img[keep], a_full.img[keep]) and torch.equal(a_part.sc[keep], a_full.sc[keep])
    assert a_part.img[~keep].abs().max().item() == 0 and a_part.sc[~keep].abs().max().item() == 0
