"""-m gpu: every kernel of libleopard_amd.so on a real MI355X, through the C ABI, at production shapes, against
plain PyTorch fp32 of the same op (computed on the device in fp32 as the checker) — plus bit-exact checks for
the integer / byte work (synthetic fill, im2col gather, merge gather, KV-cache copy)."""
import numpy as np
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.synth import KIND_BIAS, KIND_NORM, KIND_WEIGHT, name_seed, synth_array

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from leopard_amd.ops import Ops
    assert torch.cuda.is_available()
    return Ops()


DEV = "cuda:0"


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def eps(dtype):
    return 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7          # one rounding of the 16-bit type


def check(out, ref, dtype, k=4.0, what=""):
    out, ref = out.float(), ref.float()
    err = ((out - ref).abs() / (1.0 + ref.abs())).max().item()
    assert err <= k * eps(dtype), f"{what}: rel err {err:.3e} > {k * eps(dtype):.3e}"


def test_native_library_loaded(ops):
    import os
    assert os.path.exists(_lib.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libleopard_amd.so" in maps


def test_fill_synthetic_bit_exact(ops):
    for kind in (KIND_WEIGHT, KIND_BIAS, KIND_NORM):
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            out = torch.empty(100003, dtype=dt, device=DEV)
            ops.fill_synthetic(out, name_seed("language_model.lm_head.weight"), kind)
            ref = torch.from_numpy(synth_array("language_model.lm_head.weight", (100003,), kind))
            assert torch.equal(out.float().cpu(), ref)
    big = torch.empty((1 << 32) + 4096, dtype=torch.float16, device=DEV)      # exercises the 64-bit index path
    ops.fill_synthetic(big, 123, KIND_WEIGHT)
    from leopard_amd.synth import hash_bytes, values_from_bytes
    tail = values_from_bytes(hash_bytes(123, (1 << 32), 4096), KIND_WEIGHT)
    assert torch.equal(big[-4096:].float().cpu(), torch.from_numpy(tail))
    del big


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1000, 1152, 1152), (676 * 3, 3456, 1152), (333, 4352, 1152), (700, 1152, 4352),
                                   (257, 4096, 4096), (128, 128, 64), (1, 256, 128)])
def test_gemm_store_bias(ops, dtype, shape):
    M, N, K = shape
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, 0.05)
    bias = rnd((N,), torch.float32, 3)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.gemm(a, w, out, bias=bias)
    check(out, a.float() @ w.float().T + bias, dtype, what=f"gemm {shape}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_activations(ops, dtype):
    M, N, K = 500, 4352, 1152
    a, w = rnd((M, K), dtype, 4), rnd((N, K), dtype, 5, 0.05)
    bias = rnd((N,), torch.float32, 6)
    ref = a.float() @ w.float().T + bias
    out = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.gemm(a, w, out, bias=bias, act=_lib.ACT_GELU_TANH)
    check(out, torch.nn.functional.gelu(ref, approximate="tanh"), dtype, what="gelu_tanh")
    ops.gemm(a, w, out, bias=bias, act=_lib.ACT_GELU_ERF)
    check(out, torch.nn.functional.gelu(ref), dtype, what="gelu_erf")


def test_gemm_transpose_detecting(ops):
    M, N, K = 256, 256, 128
    a = torch.zeros(M, K, dtype=torch.float16, device=DEV)
    a[torch.arange(M), torch.arange(M) % K] = 1
    w = ((torch.arange(N * K).reshape(N, K) % 97).to(torch.float16) / 16).to(DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a, w, out)
    assert torch.equal(out.float(), a.float() @ w.float().T)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_residual_storef32_addmat_rowmap(ops, dtype):
    M, N, K = 1400, 1152, 640
    a, w = rnd((M, K), dtype, 7), rnd((N, K), dtype, 8, 0.05)
    bias = rnd((N,), torch.float32, 9)
    ref = a.float() @ w.float().T + bias
    x = rnd((M, N), torch.float32, 10)
    x0 = x.clone()
    ops.gemm(a, w, x, bias=bias, epilogue=_lib.EPI_RESIDUAL)
    assert (x - (x0 + ref)).abs().max() <= 2e-4
    pos = rnd((676, N), torch.float32, 11)
    out = torch.empty(M, N, device=DEV)
    ops.gemm(a, w, out, bias=bias, addmat=pos, epilogue=_lib.EPI_STORE_F32)
    assert (out - (ref + pos[torch.arange(M, device=DEV) % 676])).abs().max() <= 2e-4
    perm = torch.randperm(M + 50, generator=torch.Generator().manual_seed(12))[:M].to(torch.int32).to(DEV)
    big = torch.zeros(M + 50, N, device=DEV)
    ops.gemm(a, w, big, bias=bias, row_map=perm, epilogue=_lib.EPI_STORE_F32)
    assert (big[perm.long()] - ref).abs().max() <= 2e-4


@pytest.mark.parametrize("case", ["gelu_tanh", "gelu_erf", "swiglu", "plain"])
def test_gemm_epilogue_rows_do_not_depend_on_their_position(ops, case):
    """A row's result must not depend on where it sits in the launch (which unrolled copy of the epilogue, which tile geometry handles it): the
    packed == separate bit-identities of the engine rest on it.  hipcc contracts plain `*` / `+` chains differently in different unrolled copies
    (round 5: a rewritten tanh-GELU flipped one fp16 tie in ~3e5 elements until its multiplies were individually rounded), so every activation
    epilogue is checked here: the same rows at shifted positions of the same geometry, and through three geometries (M = 300 / 729 / 6000)."""
    from leopard_amd.weights import interleave_gate_up
    dtype = torch.float16
    K = 1152
    N = 4352 if case != "swiglu" else 2 * 2176
    a = rnd((6100, K), dtype, 31, 0.5)
    w = rnd((N, K), dtype, 32, 0.05)
    bias = None if case == "swiglu" else rnd((N,), torch.float32, 33, 0.1)
    if case == "swiglu":
        w = interleave_gate_up(w[:N // 2].contiguous(), w[N // 2:].contiguous())
    kw = {"gelu_tanh": dict(act=_lib.ACT_GELU_TANH), "gelu_erf": dict(act=_lib.ACT_GELU_ERF), "swiglu": dict(epilogue=_lib.EPI_SWIGLU), "plain": {}}[case]

    def run(rows):
        out = torch.empty(rows.shape[0], N // 2 if case == "swiglu" else N, dtype=dtype, device=DEV)
        ops.gemm(rows, w, out, bias=bias, **kw)
        return out
    base = run(a[:6000])
    for sh in (1, 7, 32, 33, 64, 100):
        assert torch.equal(run(a[sh:sh + 6000])[:6000 - sh], base[sh:]), f"{case}: rows shifted by {sh} differ"
    for lo, hi in ((0, 300), (41, 341), (0, 729), (729, 1458), (100, 164)):
        assert torch.equal(run(a[lo:hi]), base[lo:hi]), f"{case}: rows {lo}:{hi} alone differ from the same rows inside M = 6000"


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_swiglu_llama_shape(ops, dtype):
    from leopard_amd.weights import interleave_gate_up
    M, F, K = 300, 14336, 4096
    a = rnd((M, K), dtype, 13)
    gate, up = rnd((F, K), dtype, 14, 0.02), rnd((F, K), dtype, 15, 0.02)
    out = torch.empty(M, F, dtype=dtype, device=DEV)
    ops.gemm(a, interleave_gate_up(gate, up), out, epilogue=_lib.EPI_SWIGLU)
    ref = torch.nn.functional.silu(a.float() @ gate.float().T) * (a.float() @ up.float().T)
    check(out, ref, dtype, what="swiglu")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_pixel_shuffle_projector_shape(ops, dtype):
    from oracle.leopard_oracle import pixel_shuffle
    tiles, G, C, N = 3, 26, 1152, 4096
    x = rnd((tiles * G * G, C), dtype, 16)
    w = rnd((N, 4 * C), dtype, 17, 0.02)
    bias = rnd((N,), torch.float32, 18)
    out = torch.empty(tiles * 169, N, dtype=dtype, device=DEV)
    ops.gemm(x, w, out, bias=bias, act=_lib.ACT_GELU_ERF, a_mode=_lib.A_PIXEL_SHUFFLE, ps_grid=G, M=tiles * 169)
    shuf = pixel_shuffle(x.float().cpu().view(tiles, G * G, C)).reshape(tiles * 169, 4 * C).to(DEV)
    check(out, torch.nn.functional.gelu(shuf @ w.float().T + bias), dtype, what="pixel-shuffle gemm")


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms(ops, dtype):
    for D in (1152, 4096):
        M = 3001
        x = rnd((M, D), torch.float32, 20) * 3 + 0.5
        w = torch.from_numpy(synth_array("w", (D,), KIND_NORM)).to(DEV)
        b = torch.from_numpy(synth_array("b", (D,), KIND_BIAS)).to(DEV)
        out = torch.empty(M, D, dtype=dtype, device=DEV)
        ops.layernorm(x, w, b, out, 1e-6)
        check(out, torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6), dtype, k=2, what="layernorm")
        ops.rmsnorm(x, w, out, 1e-5)
        check(out, w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5)), dtype, k=2, what="rmsnorm")


@pytest.mark.parametrize("dtype", DTYPES)
def test_rope_and_kv_cache(ops, dtype):
    from leopard_amd.config import RopeScaling
    from oracle.leopard_oracle import rope_tables, rotate_half
    S, nq, nkv, D = 777, 32, 8, 128
    qkv = rnd((S, (nq + 2 * nkv) * D), dtype, 30)
    orig = qkv.clone()
    cos, sin = rope_tables(torch.arange(3000, 3000 + S), D, 5e5, RopeScaling())
    kc = torch.zeros(S + 9, nkv * D, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    ops.rope_qk(qkv, nq, nkv, D, cos[:, :D // 2].contiguous().to(DEV), sin[:, :D // 2].contiguous().to(DEV), kc, vc, 4)
    x = orig.float().cpu().view(S, nq + 2 * nkv, D)
    rot = x[:, :nq + nkv] * cos[:, None, :] + rotate_half(x[:, :nq + nkv]) * sin[:, None, :]
    got = qkv.float().cpu().view(S, nq + 2 * nkv, D)
    check(got[:, :nq + nkv], rot, dtype, k=2, what="rope")
    assert torch.equal(got[:, nq + nkv:], x[:, nq + nkv:])
    assert torch.equal(kc[4:4 + S].view(S, nkv, D), qkv.view(S, -1, D)[:, nq:nq + nkv])
    assert torch.equal(vc[4:4 + S].view(S, nkv, D), orig.view(S, -1, D)[:, nq + nkv:])
    assert kc[:4].abs().max() == 0 and kc[4 + S:].abs().max() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_merge_bit_exact(ops, dtype):
    from leopard_amd.engine import plan_merge
    D, V, tpt, n_img = 4096, 5000, 169, 3
    table = rnd((V, D), dtype, 40)
    ids = torch.randint(0, V - 1, (60,), generator=torch.Generator().manual_seed(41))
    ids[[5, 20, 21]] = V - 1
    feats = rnd((n_img * tpt, D), torch.float32, 42)
    src = torch.from_numpy(plan_merge(ids.numpy(), V - 1, n_img * tpt, tpt))
    out = torch.empty(src.numel(), D, device=DEV)
    ops.embed_merge(ids.to(DEV), src.to(DEV), table, feats, out)
    src_d = src.to(DEV)
    ref = torch.where((src_d >= 0)[:, None], table[ids.to(DEV)[src_d.clamp(min=0)]].float(), feats[(-src_d - 1).clamp(min=0)])
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_preprocess_tiles_bit_exact(ops, dtype):
    from leopard_amd.tiler import siglip_normalize
    n, S, P, ldo = 3, 364, 14, 640
    u8 = np.random.default_rng(0).integers(0, 256, (n, S, S, 3), dtype=np.uint8)
    out = torch.full((n * 676, ldo), 7.0, dtype=dtype, device=DEV)
    ops.preprocess_tiles(torch.from_numpy(u8).to(DEV), out, S, P)
    pix = torch.from_numpy(siglip_normalize(u8))
    ref = torch.nn.functional.unfold(pix, kernel_size=P, stride=P).transpose(1, 2).reshape(n * 676, 588)
    assert torch.equal(out[:, :588].float().cpu(), ref.to(dtype).float())
    assert out[:, 588:].abs().max() == 0
    out2 = torch.empty_like(out)
    ops.preprocess_tiles(pix.contiguous().to(DEV), out2, S, P)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemv_shapes(ops, dtype):
    from leopard_amd.weights import interleave_gate_up
    x = rnd((4096,), dtype, 51)
    w = rnd((8192, 4096), dtype, 50, 0.02)
    ref = w.float() @ x.float()
    out = torch.empty(8192, device=DEV)
    ops.gemv(w, x, out)
    assert (out - ref).abs().max() <= 2e-3
    o16 = torch.empty(8192, dtype=dtype, device=DEV)
    ops.gemv(w, x, o16, epilogue=1)
    check(o16, ref, dtype, k=2, what="gemv store T")
    acc = torch.ones(8192, device=DEV)
    ops.gemv(w, x, acc, epilogue=2)
    assert (acc - 1 - ref).abs().max() <= 2e-3
    gate, up = rnd((14336, 4096), dtype, 52, 0.02), rnd((14336, 4096), dtype, 53, 0.02)
    sw = torch.empty(14336, dtype=dtype, device=DEV)
    ops.gemv(interleave_gate_up(gate, up), x, sw, epilogue=3)
    check(sw, torch.nn.functional.silu(gate.float() @ x.float()) * (up.float() @ x.float()), dtype, k=2, what="gemv swiglu")
    wd, xd = rnd((4096, 14336), dtype, 54, 0.02), rnd((14336,), dtype, 55)
    od = torch.zeros(4096, device=DEV)
    ops.gemv(wd, xd, od, epilogue=2)
    assert (od - wd.float() @ xd.float()).abs().max() <= 5e-3


def attn_ref(q, k, v, cu_q, cu_k, H, KV, D, scale, causal):
    out = torch.zeros(q.shape[0], H * D, device=q.device)
    for s in range(len(cu_q) - 1):
        qs = q[cu_q[s]:cu_q[s + 1]].float().reshape(-1, H, D).transpose(0, 1)
        ks = k[cu_k[s]:cu_k[s + 1]].float().reshape(-1, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        vs = v[cu_k[s]:cu_k[s + 1]].float().reshape(-1, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        sc = qs @ ks.transpose(-1, -2) * scale
        if causal:
            lq, lk = qs.shape[1], ks.shape[1]
            m = torch.arange(lk, device=q.device)[None, :] <= torch.arange(lq, device=q.device)[:, None] + (lk - lq)
            sc = sc.masked_fill(~m, float("-inf"))
        out[cu_q[s]:cu_q[s + 1]] = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(-1, H * D)
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_tr", [False, True])
def test_attention_llama_causal_gqa(ops, dtype, use_tr):
    H, KV, D = 32, 8, 128
    cu = [0, 1000, 1000 + 77, 1000 + 77 + 333]
    T = cu[-1]
    qkv = rnd((T, (H + 2 * KV) * D), dtype, 60)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    out = torch.full((T, H * D), float("nan"), dtype=dtype, device=DEV)
    cu_t = torch.tensor(cu, dtype=torch.int32, device=DEV)
    ops.attention(q, k, v, out, cu_t, cu_t, 1000, H, KV, D, D ** -0.5, True, use_tr)
    ref = attn_ref(q, k, v, cu, cu, H, KV, D, D ** -0.5, True)
    err = (out.float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), f"use_tr={use_tr}: {err}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_tr", [False, True])
def test_attention_siglip_noncausal_d72(ops, dtype, use_tr):
    H, D, n = 16, 72, 3
    cu = [676 * i for i in range(n + 1)]
    T = cu[-1]
    qkv = rnd((T, 3 * H * D), dtype, 61)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    out = torch.full((T, H * D), float("nan"), dtype=dtype, device=DEV)
    cu_t = torch.tensor(cu, dtype=torch.int32, device=DEV)
    ops.attention(q, k, v, out, cu_t, cu_t, 676, H, H, D, D ** -0.5, False, use_tr)
    ref = attn_ref(q, k, v, cu, cu, H, H, D, D ** -0.5, False)
    err = (out.float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), f"use_tr={use_tr}: {err}"


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_pipelined_variants_on_device(ops, dtype, variant):
    """attention64.h (opt-in, attn.rows64 = 1 | 2): the software-pipelined kernels on the device — ragged causal GQA sequences spanning several
    256-row workgroups against fp32, three launches bit-identical (the ring, the counted waits and, for variant 1, the inline-asm MFMA hazard
    contract are what the CPU emulator cannot check), and for variant 2 the exactness property of a row that sees a single key."""
    H, KV, D = 8, 2, 128
    lens = [1500, 40, 700]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    T = cu[-1]
    qkv = rnd((T, (H + 2 * KV) * D), dtype, 66)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    cu_t = torch.tensor(cu, dtype=torch.int32, device=DEV)
    ref = attn_ref(q, k, v, cu, cu, H, KV, D, D ** -0.5, True)
    ops.set_option("attn.rows64", variant)
    ops.set_option("attn.rows64_min", 0)
    try:
        outs = []
        for _ in range(3):
            out = torch.full((T, H * D), float("nan"), dtype=dtype, device=DEV)
            ops.attention(q, k, v, out, cu_t, cu_t, max(lens), H, KV, D, D ** -0.5, True, True)
            outs.append(out)
        torch.cuda.synchronize()
    finally:
        ops.set_option("attn.rows64", 0)
        ops.set_option("attn.rows64_min", 1024)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = (outs[0].float() - ref).abs().max().item()
    assert err <= 3 * eps(dtype), err
    if variant == 2:
        for b in cu[:-1]:
            assert torch.equal(outs[0][b].view(H, D), v[b].view(KV, D).repeat_interleave(H // KV, 0))


def test_attention_decode_and_chunk_shapes(ops):
    H, KV, D = 32, 8, 128
    dtype = torch.float16
    k, v = rnd((700, KV * D), dtype, 62), rnd((700, KV * D), dtype, 63)
    for lq in (1, 33, 200):
        q = rnd((lq, H * D), dtype, 64)
        out = torch.empty(lq, H * D, dtype=dtype, device=DEV)
        ops.attention(q, k, v, out, torch.tensor([0, lq], dtype=torch.int32, device=DEV),
                      torch.tensor([0, 700], dtype=torch.int32, device=DEV), lq, H, KV, D, D ** -0.5, True, True)
        ref = attn_ref(q, k, v, [0, lq], [0, 700], H, KV, D, D ** -0.5, True)
        assert (out.float() - ref).abs().max() <= 3 * eps(dtype)


def test_attention_full_size_properties(ops):
    """C3-size causal sequence (S=7187): size-independent properties — with V == 1 every output is exactly 1
    (softmax rows sum to one), row 0 returns V[0], and two launches are bit-identical."""
    H, KV, D, S = 32, 8, 128, 7187
    dtype = torch.bfloat16
    qkv = rnd((S, (H + 2 * KV) * D), dtype, 70)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    out = torch.empty(S, H * D, dtype=dtype, device=DEV)
    ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    out2 = torch.empty_like(out)
    ops.attention(q, k, v, out2, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    assert torch.equal(out, out2)
    assert torch.equal(out[0].view(H, D), v[0].view(KV, D).repeat_interleave(H // KV, 0))
    ones = torch.ones(S, KV * D, dtype=dtype, device=DEV)
    ops.attention(q, k, ones, out, cu, cu, S, H, KV, D, D ** -0.5, True, True)
    assert (out.float() - 1).abs().max() <= 2.0 ** -7


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_decode_split_kv_long_cache(ops, dtype):
    """lmi_attn_decode_fwd at the C3 context: one (and five) query rows against 7187 keys, launch geometry fixed by a
    larger cache capacity (as under the captured decode graph), against the fp32 reference and the single-pass kernel."""
    H, KV, D, lk, cap = 32, 8, 128, 7187, 7187 + 128
    k, v = rnd((cap, KV * D), dtype, 65), rnd((cap, KV * D), dtype, 66)
    for lq in (1, 5):
        q = rnd((lq, H * D), dtype, 67)
        out = torch.full((lq, H * D), float("nan"), dtype=dtype, device=DEV)
        ws = torch.empty(ops.decode_workspace_elems(lq, H, D, cap), dtype=torch.float32, device=DEV)
        cu_q = torch.tensor([0, lq], dtype=torch.int32, device=DEV)
        cu_k = torch.tensor([0, lk], dtype=torch.int32, device=DEV)
        ops.attention_decode(q, k, v, out, cu_q, cu_k, lq, cap, H, KV, D, D ** -0.5, ws)
        ref = attn_ref(q, k[:lk], v[:lk], [0, lq], [0, lk], H, KV, D, D ** -0.5, True)
        assert (out.float() - ref).abs().max() <= 3 * eps(dtype)
        one = torch.empty_like(out)
        ops.attention(q, k, v, one, cu_q, cu_k, lq, H, KV, D, D ** -0.5, True, True)
        assert (out.float() - one.float()).abs().max() <= 3 * eps(dtype)


@pytest.mark.parametrize("lk,cap", [(7187, 7424), (228, 512), (1, 64), (4100, 16384)])
def test_attention_decode_every_split_size(ops, lk, cap):
    """The split-KV geometry (attn.decode_split_tiles: 64 ... 512 keys per workgroup; 0 = chosen from the launch shape) never changes
    the answer beyond the summation order: every setting within 3 eps of the fp32 reference, splits past the sequence's end (the
    capacity fixes the grid, not the length) contributing nothing; repeated launches agree bit for bit."""
    dtype, H, KV, D = torch.float16, 32, 8, 128
    k, v = rnd((cap, KV * D), dtype, 71), rnd((cap, KV * D), dtype, 72)
    q = rnd((1, H * D), dtype, 73)
    cu_q = torch.tensor([0, 1], dtype=torch.int32, device=DEV)
    cu_k = torch.tensor([0, lk], dtype=torch.int32, device=DEV)
    ref = attn_ref(q, k[:lk], v[:lk], [0, 1], [0, lk], H, KV, D, D ** -0.5, True)
    ws = torch.full((ops.decode_workspace_elems(1, H, D, cap),), float("nan"), dtype=torch.float32, device=DEV)
    try:
        for st in (0, 1, 2, 4, 8):
            ops.set_option("attn.decode_split_tiles", st)
            outs = [torch.full((1, H * D), float("nan"), dtype=dtype, device=DEV) for _ in range(2)]
            for o in outs:
                ops.attention_decode(q, k, v, o, cu_q, cu_k, 1, cap, H, KV, D, D ** -0.5, ws)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[1]), st
            assert (outs[0].float() - ref).abs().max() <= 3 * eps(dtype), st
    finally:
        ops.set_option("attn.decode_split_tiles", 0)


def test_gemv_rmsnorm_equals_rmsnorm_then_gemv(ops):
    dtype = torch.float16
    K, N = 4096, 6144
    w = rnd((N, K), dtype, 110, 0.02)
    x = rnd((1, K), torch.float32, 111, 3.0)
    g = rnd((K,), torch.float32, 112).abs() + 0.5
    h = torch.empty(1, K, dtype=dtype, device=DEV)
    ops.rmsnorm(x, g, h, 1e-5)
    a, b = torch.empty(N, dtype=dtype, device=DEV), torch.empty(N, dtype=dtype, device=DEV)
    ops.gemv(w, h[0], a, epilogue=1)
    ops.gemv_rmsnorm(w, x[0], g, 1e-5, b, epilogue=1)
    assert torch.equal(a, b)
    ref = w.float() @ h[0].float()
    check(a, ref, dtype, k=2, what="gemv_rmsnorm")


# ---- batched decode: skinny-M projections, per-row RoPE into a pooled cache, pooled split-KV attention ----------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 3, 8, 16])
def test_gemm_skinny_llama_decode_shapes(ops, dtype, M):
    """lmi_gemm_skinny at the four projection shapes of a Llama-3.1-8B decode step, every epilogue, vs fp32 on the device; three
    launches each must agree bit for bit (the partial tiles are summed in a fixed order: no atomics)."""
    for name, N, K, epi in (("qkv", 6144, 4096, 0), ("o", 4096, 4096, 1), ("gate_up", 28672, 4096, 2), ("down", 4096, 14336, 1), ("head", 8192, 4096, 3)):
        x, w = rnd((M, K), dtype, 11, 1.0), rnd((N, K), dtype, 12, 0.02)
        lin = x.float() @ w.float().T
        if epi == 2:
            lv = lin.view(M, N // 64, 2, 32)
            ref = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, N // 2)
            outs = [torch.zeros(M, N // 2, dtype=dtype, device=DEV) for _ in range(3)]
        elif epi == 1:
            base = rnd((M, N), torch.float32, 13)
            ref = base + lin
            outs = [base.clone() for _ in range(3)]
        else:
            ref = lin
            outs = [torch.zeros(M, N, dtype=dtype if epi == 0 else torch.float32, device=DEV) for _ in range(3)]
        from leopard_amd.weights import skinny_pack
        wp = skinny_pack(w)
        for j, o in enumerate(outs):                               # launches 0, 1: row-major weights; launch 2: the packed copy — same bits
            ops.gemm_skinny(wp if j == 2 else w, x, o, epi, packed=(j == 2))
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), name
        check(outs[0], ref, dtype if epi in (0, 2) else torch.float16, k=4.0, what=f"skinny {name} M={M}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 8, 16])
def test_rope_qkv_skinny_llama_shape(ops, dtype, M):
    """lmi_rope_qkv_skinny at the Llama-3.1-8B decode shape (32 + 8 + 8 heads of 128, K = 4096, C3-sized cache slots): vs fp32 on the
    device, packed == row-major weights bit for bit, K / V rows land in slot m at pos[m] and nowhere else."""
    from leopard_amd.weights import rope_permute_rows, skinny_pack
    H, KV, hd, K, cap = 32, 8, 128, 4096, 7424
    w, x = rnd(((H + 2 * KV) * hd, K), dtype, 31, 0.02), rnd((M, K), dtype, 32)
    f = torch.arange(cap, device=DEV).float().reshape(-1, 1) * (1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = ((torch.arange(M, dtype=torch.int64) * 977 + 7186) % cap).to(torch.int32).to(DEV)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    wp = skinny_pack(w_rope)
    res = []
    for packed in (False, True):
        kp, vp = torch.zeros(M * cap, KV * hd, dtype=dtype, device=DEV), torch.zeros(M * cap, KV * hd, dtype=dtype, device=DEV)
        got = torch.zeros(M, (H + 2 * KV) * hd, dtype=dtype, device=DEV)
        ops.rope_qkv_skinny(wp if packed else w_rope, x, got, H, KV, hd, cos, sin, kp, vp, cap, pos, packed=packed)
        res.append((got, kp, vp))
    torch.cuda.synchronize()
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    got, kp, vp = res[0]
    lin = (x.float() @ w.float().T).view(M, H + 2 * KV, hd)
    c, sn = cos[pos.long()].unsqueeze(1), sin[pos.long()].unsqueeze(1)
    ref = lin.clone()
    a, b = lin[:, :H + KV, :64], lin[:, :H + KV, 64:]
    ref[:, :H + KV, :64] = a * c - b * sn
    ref[:, :H + KV, 64:] = b * c + a * sn
    check(got, ref.view(M, -1), dtype, k=4.0, what=f"rope_qkv_skinny M={M}")
    rows = torch.arange(M, device=DEV) * cap + pos.long()
    assert torch.equal(kp[rows], got[:, H * hd:(H + KV) * hd]) and torch.equal(vp[rows], got[:, (H + KV) * hd:])
    mask = torch.ones(M * cap, dtype=torch.bool, device=DEV)
    mask[rows] = False
    assert not kp[mask].any() and not vp[mask].any()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 8, 16])
def test_gemm_skinny_folded_rmsnorm_llama_shapes(ops, dtype, M):
    """lmi_gemm_skinny_ex at the Llama-3.1-8B decode shapes: the o_proj producer (residual + T(x gamma) + 256 row-square partials per row)
    feeding the gate/up SwiGLU consumer and the q|k|v + RoPE consumer, against the norm launch + plain projections and fp32 on the
    device; launches agree bit for bit."""
    from leopard_amd.weights import rope_permute_rows, skinny_pack
    D, FF, H, KV, hd, cap, eps_n = 4096, 14336, 32, 8, 128, 512, 1e-5
    a, w_o = rnd((M, D), dtype, 81), skinny_pack(rnd((D, D), dtype, 82, 0.02))
    x0, gamma = rnd((M, D), torch.float32, 83, 2.0), rnd((D,), torch.float32, 84) + 1.0
    runs = []
    for _ in range(2):
        x = x0.clone()
        h, sq = torch.zeros(M, D, dtype=dtype, device=DEV), torch.zeros(M, D // 16, device=DEV)
        ops.gemm_skinny(w_o, a, x, 1, True, norm_out=h, norm_gamma=gamma, rowsq_out=sq)
        runs.append((x, h, sq))
    torch.cuda.synchronize()
    assert all(torch.equal(p, q) for p, q in zip(runs[0], runs[1]))
    x, h, sq = runs[0]
    check(h, x * gamma, dtype, k=1.0, what="producer norm_out")    # one rounding of x * gamma (the device may round the product once, torch twice)
    want_sq = (x.double() ** 2).view(M, D // 16, 16).sum(-1)
    assert (sq.double() - want_sq).abs().max() <= 1e-5 * want_sq.abs().max()
    rstd = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + eps_n)
    hn = torch.zeros(M, D, dtype=dtype, device=DEV)
    ops.rmsnorm(x, gamma, hn, eps_n)
    # gate / up
    w_gu = rnd((2 * FF, D), dtype, 85, 0.02)
    wp = skinny_pack(w_gu)
    gu, gu2 = torch.zeros(M, FF, dtype=dtype, device=DEV), torch.zeros(M, FF, dtype=dtype, device=DEV)
    ops.gemm_skinny(wp, h, gu, 2, True, rowsq_in=sq, norm_dim=D, norm_eps=eps_n)
    ops.gemm_skinny(wp, hn, gu2, 2, True)
    lv = ((h.double() @ w_gu.double().T) * rstd).view(M, 2 * FF // 64, 2, 32)
    want = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, FF).float()
    check(gu, want, dtype, k=4.0, what="folded gate/up")
    check(gu, gu2, dtype, k=6.0, what="folded vs norm launch, gate/up")
    # q | k | v + RoPE
    w = rnd(((H + 2 * KV) * hd, D), dtype, 86, 0.02)
    w_rope = skinny_pack(torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous())
    f = torch.arange(cap, device=DEV).float().reshape(-1, 1) * (1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = ((torch.arange(M, dtype=torch.int64) * 97 + 300) % cap).to(torch.int32).to(DEV)
    outs = []
    for xin, rs in ((h, sq), (hn, None)):
        kp, vp = torch.zeros(M * cap, KV * hd, dtype=dtype, device=DEV), torch.zeros(M * cap, KV * hd, dtype=dtype, device=DEV)
        got = torch.zeros(M, (H + 2 * KV) * hd, dtype=dtype, device=DEV)
        ops.rope_qkv_skinny(w_rope, xin, got, H, KV, hd, cos, sin, kp, vp, cap, pos, packed=True, rowsq_in=rs, norm_eps=eps_n)
        outs.append((got, kp, vp))
    for g, u in zip(outs[0], outs[1]):
        check(g, u, dtype, k=6.0, what="folded vs norm launch, q|k|v")
    lin = ((h.double() @ w.double().T) * rstd).float().view(M, H + 2 * KV, hd)
    c, sn = cos[pos.long()].unsqueeze(1), sin[pos.long()].unsqueeze(1)
    ref = lin.clone()
    a_, b_ = lin[:, :H + KV, :64], lin[:, :H + KV, 64:]
    ref[:, :H + KV, :64] = a_ * c - b_ * sn
    ref[:, :H + KV, 64:] = b_ * c + a_ * sn
    check(outs[0][0], ref.view(M, -1), dtype, k=4.0, what="folded q|k|v")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemv_rmsnorm_rope_llama_shape(ops, dtype):
    """lmi_gemv_rmsnorm_rope at the Llama-3.1-8B decode shape vs fp32 on the device and vs lmi_gemv_rmsnorm + lmi_rope_qk_at (V bit for
    bit — no rotation; q / k within a rounding: the fused launch rotates the unrounded sums); launches agree bit for bit; only cache row
    *pos is written."""
    from leopard_amd.weights import rope_permute_rows
    H, KV, hd, K, cap = 32, 8, 128, 4096, 7424
    N = (H + 2 * KV) * hd
    w, x, g = rnd((N, K), dtype, 51, 0.02), rnd((1, K), torch.float32, 52, 2.0), rnd((K,), torch.float32, 53) + 1.0
    f = torch.arange(cap, device=DEV).float().reshape(-1, 1) * (1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    P = 7186
    pos = torch.tensor([P], dtype=torch.int32, device=DEV)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    outs = []
    for _ in range(2):
        kc, vc = torch.zeros(cap, KV * hd, dtype=dtype, device=DEV), torch.zeros(cap, KV * hd, dtype=dtype, device=DEV)
        got = torch.zeros(1, N, dtype=dtype, device=DEV)
        ops.gemv_rmsnorm_rope(w_rope, x[0], g, 1e-5, got[0], H, KV, hd, cos, sin, kc, vc, pos)
        outs.append((got, kc, vc))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    got, kc, vc = outs[0]
    xn = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * g).to(dtype).float()
    lin = (xn @ w.float().T).view(H + 2 * KV, hd)
    ref = lin.clone()
    a, b = lin[:H + KV, :64], lin[:H + KV, 64:]
    ref[:H + KV, :64] = a * cos[P] - b * sin[P]
    ref[:H + KV, 64:] = b * cos[P] + a * sin[P]
    check(got, ref.view(1, -1), dtype, k=4.0, what="gemv_rmsnorm_rope")
    two = torch.zeros(1, N, dtype=dtype, device=DEV)
    k2, v2 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.gemv_rmsnorm(w, x[0], g, 1e-5, two[0], epilogue=1)
    ops.rope_qk_at(two, H, KV, hd, cos, sin, k2, v2, pos)
    assert torch.equal(got[0, (H + KV) * hd:], two[0, (H + KV) * hd:]) and torch.equal(vc, v2)
    check(got, two, dtype, k=4.0, what="gemv_rmsnorm_rope vs two launches")
    assert torch.equal(kc[P], got[0, H * hd:(H + KV) * hd]) and torch.equal(vc[P], got[0, (H + KV) * hd:])
    kc[P] = 0
    assert not kc.any()


@pytest.mark.parametrize("M", [1, 8, 16, 32])
def test_norm_small_m(ops, M):
    """RMSNorm D = 4096 / LayerNorm D = 1152 at decode row counts (one workgroup per row) vs fp64 on the device."""
    for rms, D in ((True, 4096), (False, 1152)):
        x, g, b = rnd((M, D), torch.float32, 41, 2.0) + 0.3, rnd((D,), torch.float32, 42) + 1.0, rnd((D,), torch.float32, 43)
        out = torch.zeros(M, D, dtype=torch.float16, device=DEV)
        xd = x.double()
        if rms:
            ops.rmsnorm(x, g, out, 1e-5)
            ref = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-5) * g.double()
        else:
            ops.layernorm(x, g, b, out, 1e-6)
            ref = torch.nn.functional.layer_norm(xd, (D,), g.double(), b.double(), 1e-6)
        check(out, ref.float(), torch.float16, k=2.0, what=f"norm rows M={M} D={D}")


def test_rope_rows_and_decode_pool_vs_per_sequence(ops):
    """lmi_rope_qk_rows / lmi_attn_decode_pool (B sequences in slots of one pooled cache) == lmi_rope_qk_at / lmi_attn_decode_fwd
    sequence by sequence, bit for bit, at the Llama head geometry and a C3-sized slot."""
    dtype, H, KV, hd, cap = torch.float16, 32, 8, 128, 7424
    lens = [7187, 1, 300, 7423, 64]
    B = len(lens)
    qkv = rnd((B, (H + 2 * KV) * hd), dtype, 21)
    f = torch.arange(cap, device=DEV).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=DEV).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    kp, vp = rnd((B * cap, KV * hd), dtype, 22), rnd((B * cap, KV * hd), dtype, 23)
    kp1, vp1 = kp.clone(), vp.clone()
    pos = torch.tensor([l - 1 for l in lens], dtype=torch.int32, device=DEV)
    got = qkv.clone()
    ops.rope_qk_rows(got, H, KV, hd, cos, sin, kp, vp, cap, pos)
    out = torch.zeros(B, H * hd, dtype=dtype, device=DEV)
    ws = torch.zeros(ops.decode_workspace_elems(B, H, hd, cap), device=DEV)
    cu_q = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    k_begin = (torch.arange(B, dtype=torch.int32) * cap).to(DEV)
    ops.attention_decode_pool(got[:, :H * hd], kp, vp, out, cu_q, k_begin, torch.tensor(lens, dtype=torch.int32, device=DEV), cap, H, KV, hd,
                              hd ** -0.5, ws)
    for s, L in enumerate(lens):
        one = qkv[s:s + 1].clone()
        kc, vc = kp1[s * cap:(s + 1) * cap], vp1[s * cap:(s + 1) * cap]
        ops.rope_qk_at(one, H, KV, hd, cos, sin, kc, vc, pos[s:s + 1].clone())
        assert torch.equal(one[0], got[s]) and torch.equal(kc, kp[s * cap:(s + 1) * cap]) and torch.equal(vc, vp[s * cap:(s + 1) * cap])
        o1 = torch.zeros(1, H * hd, dtype=dtype, device=DEV)
        ws1 = torch.zeros(ops.decode_workspace_elems(1, H, hd, cap), device=DEV)
        ops.attention_decode(one[:, :H * hd], kc, vc, o1, torch.tensor([0, 1], dtype=torch.int32, device=DEV),
                             torch.tensor([0, L], dtype=torch.int32, device=DEV), 1, cap, H, KV, hd, hd ** -0.5, ws1)
        assert torch.equal(o1[0], out[s]), s


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_fp8_arithmetic_at_the_llama_shape(ops, dtype):
    """lmi_attn_prep_fp8 + lmi_attn_fp8_fwd at the C3 Llama shape (S = 7187 + a ragged second sequence, 32 q / 8 kv heads x 128, causal): q8 is
    exactly e4m3(q * scale); the attention equals the 16-bit kernel run on the SAME e4m3 operand values (exact in 16 bits) up to the e4m3
    rounding of P; three launches are bit-identical; the fp8 output form agrees with the 16-bit one."""
    H, KV, D = 32, 8, 128
    lens = [7187, 300]
    S = sum(lens)
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = (torch.randn(S, (H + 2 * KV) * D, generator=g, device=DEV) * 1.5).to(dtype)
    cu = torch.tensor([0, lens[0], S], dtype=torch.int32, device=DEV)
    tiles = [(L + 63) // 64 for L in lens]
    tb = torch.tensor([0, tiles[0], sum(tiles)], dtype=torch.int32, device=DEV)
    NT = sum(tiles)
    sq, sk, sv = 32.0, 32.0, 64.0
    q8 = torch.zeros(S, H * D, dtype=torch.uint8, device=DEV)
    k_img = torch.zeros(KV * NT * 8192, dtype=torch.uint8, device=DEV)
    v_img = torch.zeros(KV * NT * 8192, dtype=torch.uint8, device=DEV)
    ops.attn_prep_fp8(qkv, cu, tb, NT, H, KV, D, sq, sk, sv, q8, k_img, v_img)
    e4 = lambda x, s_: (x.float() * s_).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q8.view(torch.float8_e4m3fn).float(), e4(qkv[:, :H * D], sq).float())
    deq = torch.cat([e4(qkv[:, :H * D], sq).float() / sq, e4(qkv[:, H * D:(H + KV) * D], sk).float() / sk,
                     e4(qkv[:, (H + KV) * D:], sv).float() / sv], 1).to(dtype).contiguous()         # e4m3 values are exact in both 16-bit types
    ref = torch.zeros(S, H * D, dtype=dtype, device=DEV)
    ops.attention(deq[:, :H * D], deq[:, H * D:(H + KV) * D], deq[:, (H + KV) * D:], ref, cu, cu, max(lens), H, KV, D, D ** -0.5, True)
    outs = []
    for _ in range(3):
        out = torch.full((S, H * D), float("nan"), dtype=dtype, device=DEV)
        ops.attention_fp8(q8, k_img, v_img, out, cu, tb, NT, max(lens), H, KV, D, D ** -0.5, sq, sk, sv, causal=True)
        outs.append(out)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = (outs[0].float() - ref.float()).abs()
    scale = ref.float().abs().max().item()
    print(f"[fp8 attention {dtype}] vs the 16-bit kernel on the same e4m3 operands: max {err.max().item() / scale:.3e} of max|O|, rms {err.pow(2).mean().sqrt().item() / ref.float().pow(2).mean().sqrt().item():.3e}")
    assert not torch.isnan(outs[0].float()).any() and err.max().item() <= 0.05 * scale
    o8 = torch.zeros(S, H * D, dtype=torch.uint8, device=DEV)
    ops.attention_fp8(q8, k_img, v_img, o8, cu, tb, NT, max(lens), H, KV, D, D ** -0.5, sq, sk, sv, causal=True, out_fp8_scale=64.0, dtype=dtype)
    assert (o8.view(torch.float8_e4m3fn).float() / 64.0 - outs[0].float()).abs().max().item() <= 0.07 * scale
