"""Kernel LOGIC tests on CPU: the HIP kernel sources compiled against tools/hipemu (a lock-step fibre emulator
of workgroups / wave64 collectives / MFMA fragment layouts) and driven through the same C ABI + Python
wrappers as the GPU build.  These check tile indexing, LDS swizzles, masks, epilogues and the fragment
bookkeeping of the attention kernel against plain PyTorch fp32 — not hardware semantics (those are pinned by
the -m gpu tests on the real chip).  Shapes are small: the emulator runs ~1e5x slower than the GPU."""
import numpy as np
import pytest
import torch

from leopard_amd import _lib
from leopard_amd.synth import KIND_BIAS, KIND_NORM, KIND_WEIGHT, name_seed, synth_array
from tests.emu_util import emu_ops

DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    return emu_ops()


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def tol(dtype):
    return 2e-3 if dtype == torch.float16 else 1.6e-2


def _declared_symbols():
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "leopard_amd.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t|const char\*)\s+(lmi_\w+)\(", hdr, flags=re.M)))


def test_abi_exports_every_declared_symbol(ops):
    """Every entry point include/leopard_amd.h declares is exported by the emulator build AND by the product library
    (libleopard_amd.so, built by `make`; no compute call is made on it here), and has a ctypes prototype in _lib."""
    import ctypes
    import os
    declared = _declared_symbols()
    assert len(declared) >= 18 and "lmi_gemm" in declared and "lmi_attn_decode_fwd" in declared
    bound = set(_lib.SIGNATURES) | {"lmi_last_error", "lmi_attn_decode_workspace_bytes", "lmi_llm_prefill_workspace_bytes", "lmi_vit_workspace_bytes"}
    assert set(declared) == bound, set(declared) ^ bound
    for name in declared:
        assert hasattr(ops.lib, name), name
    assert ops.lib.lmi_abi_version() == 1
    if os.path.exists(_lib.LIB_PATH):
        prod = ctypes.CDLL(_lib.LIB_PATH)
        for name in declared:
            assert hasattr(prod, name), f"libleopard_amd.so does not export {name}"
    else:
        pytest.skip("libleopard_amd.so not built (run `make`)")


def test_fill_synthetic_bit_exact(ops):
    for kind in (KIND_WEIGHT, KIND_BIAS, KIND_NORM):
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            out = torch.empty(5000, dtype=dt)
            ops.fill_synthetic(out, name_seed("some.param"), kind)
            ref = torch.from_numpy(synth_array("some.param", (5000,), kind))
            assert torch.equal(out.float(), ref), (kind, dt)       # exactly representable in every dtype


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_store_bias_act_ragged_m(ops, dtype):
    M, N, K = 200, 256, 192                                # 2 row tiles (second ragged), 2 col tiles, 3 k-tiles
    a, w = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, 0.1)
    bias = rnd((N,), torch.float32, 3)
    ref = a.float() @ w.float().T + bias
    for act, f in ((_lib.ACT_NONE, lambda x: x), (_lib.ACT_GELU_TANH, lambda x: torch.nn.functional.gelu(x, approximate="tanh")),
                   (_lib.ACT_GELU_ERF, torch.nn.functional.gelu)):
        out = torch.full((M, N), float("nan"), dtype=dtype)
        ops.gemm(a, w, out, bias=bias, act=act)
        assert (out.float() - f(ref)).abs().max() <= tol(dtype) * max(1.0, f(ref).abs().max().item())


def test_gemm_transpose_detecting(ops):
    """A = one-hot rows, asymmetric W: catches swapped row/col in the accumulator write-out."""
    M, N, K = 128, 128, 64
    a = torch.zeros(M, K, dtype=torch.float16)
    a[torch.arange(M), torch.arange(M) % K] = 1
    w = (torch.arange(N * K).reshape(N, K) % 97).to(torch.float16) / 16
    out = torch.empty(M, N, dtype=torch.float16)
    ops.gemm(a, w, out)
    assert torch.equal(out.float(), a.float() @ w.float().T)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_residual_f32_storef32_addmat_rowmap(ops, dtype):
    M, N, K = 130, 128, 128
    a, w = rnd((M, K), dtype, 4), rnd((N, K), dtype, 5, 0.1)
    bias = rnd((N,), torch.float32, 6)
    ref = a.float() @ w.float().T + bias
    x = rnd((M, N), torch.float32, 7)
    x0 = x.clone()
    ops.gemm(a, w, x, bias=bias, epilogue=_lib.EPI_RESIDUAL)
    assert (x - (x0 + ref)).abs().max() <= 1e-4
    pos = rnd((13, N), torch.float32, 8)
    out = torch.empty(M, N)
    ops.gemm(a, w, out, bias=bias, addmat=pos, epilogue=_lib.EPI_STORE_F32)
    assert (out - (ref + pos[torch.arange(M) % 13])).abs().max() <= 1e-4
    perm = torch.randperm(M + 20, generator=torch.Generator().manual_seed(9))[:M].to(torch.int32)
    big = torch.zeros(M + 20, N)
    ops.gemm(a, w, big, bias=bias, row_map=perm, epilogue=_lib.EPI_STORE_F32)
    assert (big[perm.long()] - ref).abs().max() <= 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_swiglu_interleaved(ops, dtype):
    M, F, K = 70, 128, 64                                   # F gate rows + F up rows -> N = 256
    a = rnd((M, K), dtype, 10)
    gate, up = rnd((F, K), dtype, 11, 0.2), rnd((F, K), dtype, 12, 0.2)
    w = torch.stack([gate.view(F // 32, 32, K), up.view(F // 32, 32, K)], dim=1).reshape(2 * F, K).contiguous()
    out = torch.empty(M, F, dtype=dtype)
    ops.gemm(a, w, out, epilogue=_lib.EPI_SWIGLU)
    g, u = a.float() @ gate.float().T, a.float() @ up.float().T
    ref = torch.nn.functional.silu(g) * u
    assert (out.float() - ref).abs().max() <= tol(dtype) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_pixel_shuffle_gather(ops, dtype):
    """projector linear_1 with the 2x2 pixel shuffle folded into the A-row gather (EVAL:165-192)."""
    from oracle.leopard_oracle import pixel_shuffle
    tiles, G, Cc, N = 3, 4, 64, 128                          # 16 ViT tokens / tile -> 4 shuffled rows of 256
    x = rnd((tiles * G * G, Cc), dtype, 13)
    w = rnd((N, 4 * Cc), dtype, 14, 0.1)
    bias = rnd((N,), torch.float32, 15)
    out = torch.empty(tiles * 4, N, dtype=dtype)
    ops.gemm(x, w, out, bias=bias, act=_lib.ACT_GELU_ERF, a_mode=_lib.A_PIXEL_SHUFFLE, ps_grid=G, M=tiles * 4)
    shuf = pixel_shuffle(x.float().view(tiles, G * G, Cc)).reshape(tiles * 4, 4 * Cc)
    ref = torch.nn.functional.gelu(shuf @ w.float().T + bias)
    assert (out.float() - ref).abs().max() <= tol(dtype) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_every_tile_geometry(ops, cfg):
    """All GemmCfg geometries (128x128 .. 256x256, 2- and 3-slot LDS rings) on ragged M, an N that is not a multiple
    of the tile width, and enough k-tiles to wrap the ring several times."""
    dtype = torch.float16
    ops.set_option("gemm.config", cfg)
    try:
        M, N, K = 300, 384, 448
        a, w = rnd((M, K), dtype, 70 + cfg), rnd((N, K), dtype, 80 + cfg, 0.1)
        bias = rnd((N,), torch.float32, 90)
        out = torch.full((M, N), float("nan"), dtype=dtype)
        ops.gemm(a, w, out, bias=bias)
        ref = a.float() @ w.float().T + bias
        assert (out.float() - ref).abs().max() <= tol(dtype) * max(1.0, ref.abs().max().item())
        F = 192
        gate, up = rnd((F, K), dtype, 11, 0.2), rnd((F, K), dtype, 12, 0.2)
        wi = torch.stack([gate.view(F // 32, 32, K), up.view(F // 32, 32, K)], dim=1).reshape(2 * F, K).contiguous()
        o2 = torch.full((M, F), float("nan"), dtype=dtype)
        ops.gemm(a, wi, o2, epilogue=_lib.EPI_SWIGLU)
        r2 = torch.nn.functional.silu(a.float() @ gate.float().T) * (a.float() @ up.float().T)
        assert (o2.float() - r2).abs().max() <= tol(dtype) * max(1.0, r2.abs().max().item())
        x = rnd((M, N), torch.float32, 7)
        x0 = x.clone()
        ops.gemm(a, w, x, bias=bias, epilogue=_lib.EPI_RESIDUAL)
        assert (x - (x0 + ref)).abs().max() <= 2e-4
    finally:
        ops.set_option("gemm.config", -1)


def test_gemm_rejects_bad_shapes(ops):
    a, w, out = torch.zeros(8, 64, dtype=torch.float16), torch.zeros(100, 64, dtype=torch.float16), torch.zeros(8, 100, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="128"):
        ops.gemm(a, w, out)


@pytest.mark.parametrize("dtype", DTYPES)
def test_norms(ops, dtype):
    for D in (1152, 4096, 64):
        M = 7
        x = rnd((M, D), torch.float32, 20) * 3 + 0.5
        w = torch.from_numpy(synth_array("w", (D,), KIND_NORM))
        b = torch.from_numpy(synth_array("b", (D,), KIND_BIAS))
        out = torch.empty(M, D, dtype=dtype)
        ops.layernorm(x, w, b, out, 1e-6)
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
        assert (out.float() - ref).abs().max() <= tol(dtype) * 4
        ops.rmsnorm(x, w, out, 1e-5)
        ref = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5))
        assert (out.float() - ref).abs().max() <= tol(dtype) * 4


@pytest.mark.parametrize("dtype", DTYPES)
def test_rope_and_kv_cache(ops, dtype):
    from oracle.leopard_oracle import rope_tables, rotate_half
    from leopard_amd.config import RopeScaling
    S, nq, nkv, D = 9, 4, 2, 128
    qkv = rnd((S, (nq + 2 * nkv) * D), dtype, 30)
    orig = qkv.clone()
    pos = torch.arange(100, 100 + S)
    cos, sin = rope_tables(pos, D, 5e5, RopeScaling())
    kc, vc = torch.zeros(20, nkv * D, dtype=dtype), torch.zeros(20, nkv * D, dtype=dtype)
    ops.rope_qk(qkv, nq, nkv, D, cos[:, :D // 2].contiguous(), sin[:, :D // 2].contiguous(), kc, vc, cache_pos0=5)
    x = orig.float().view(S, nq + 2 * nkv, D)
    rot = x[:, :nq + nkv] * cos[:, None, :] + rotate_half(x[:, :nq + nkv]) * sin[:, None, :]
    got = qkv.float().view(S, nq + 2 * nkv, D)
    assert (got[:, :nq + nkv] - rot).abs().max() <= tol(dtype) * 4
    assert torch.equal(got[:, nq + nkv:], x[:, nq + nkv:])
    assert torch.equal(kc[5:5 + S].view(S, nkv, D), qkv.view(S, -1, D)[:, nq:nq + nkv])
    assert torch.equal(vc[5:5 + S].view(S, nkv, D), orig.view(S, -1, D)[:, nq + nkv:])
    assert kc[:5].abs().max() == 0 and kc[5 + S:].abs().max() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_merge(ops, dtype):
    from oracle.leopard_oracle import merge_plan
    D, V, tpt = 64, 50, 4
    table = rnd((V, D), dtype, 40)
    ids = torch.tensor([3, 49, 7, 49, 49, 1])
    feats = rnd((3 * tpt, D), torch.float32, 41)
    src = torch.from_numpy(merge_plan(ids.numpy(), 49, 3 * tpt, tpt))
    out = torch.empty(src.numel(), D)
    ops.embed_merge(ids, src, table, feats, out)
    ref = torch.stack([table[ids[s]].float() if s >= 0 else feats[-s - 1] for s in src.tolist()])
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_preprocess_tiles(ops, dtype):
    from leopard_amd.tiler import siglip_normalize
    n, S, P, ldo = 2, 28, 14, 640
    u8 = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (n, S, S, 3), dtype=np.uint8))
    out = torch.full((n * 4, ldo), 7.0, dtype=dtype)
    ops.preprocess_tiles(u8, out, S, P)
    pix = torch.from_numpy(siglip_normalize(u8.numpy()))                        # [n,3,S,S]
    ref = torch.nn.functional.unfold(pix, kernel_size=P, stride=P).transpose(1, 2).reshape(n * 4, 3 * P * P)
    assert torch.equal(out[:, :588].float(), ref.to(dtype).float())
    assert out[:, 588:].abs().max() == 0
    out2 = torch.empty_like(out)
    ops.preprocess_tiles(pix.contiguous(), out2, S, P)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemv(ops, dtype):
    N, K = 192, 1152
    w, x = rnd((N, K), dtype, 50, 0.1), rnd((K,), dtype, 51)
    ref = w.float() @ x.float()
    out = torch.empty(N)
    ops.gemv(w, x, out)
    assert (out - ref).abs().max() <= 1e-3
    o16 = torch.empty(N, dtype=dtype)
    ops.gemv(w, x, o16, epilogue=1)
    assert (o16.float() - ref).abs().max() <= tol(dtype) * 4
    acc = torch.ones(N)
    ops.gemv(w, x, acc, epilogue=2)
    assert (acc - 1 - ref).abs().max() <= 1e-3
    wi = torch.stack([w[:96].view(3, 32, K), w[96:].view(3, 32, K)], dim=1).reshape(N, K).contiguous()
    sw = torch.empty(96, dtype=dtype)
    ops.gemv(wi, x, sw, epilogue=3)
    r = torch.nn.functional.silu(ref[:96]) * ref[96:]
    assert ((sw.float() - r).abs() / (1 + r.abs())).max() <= tol(dtype)


def attn_ref(q, k, v, cu_q, cu_k, H, KV, D, scale, causal):
    out = torch.zeros(q.shape[0], H * D)
    for s in range(len(cu_q) - 1):
        qs = q[cu_q[s]:cu_q[s + 1]].float().view(-1, H, D).transpose(0, 1)
        ks = k[cu_k[s]:cu_k[s + 1]].float().view(-1, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        vs = v[cu_k[s]:cu_k[s + 1]].float().view(-1, KV, D).transpose(0, 1).repeat_interleave(H // KV, 0)
        sc = qs @ ks.transpose(-1, -2) * scale
        if causal:
            lq, lk = qs.shape[1], ks.shape[1]
            m = torch.arange(lk)[None, :] <= torch.arange(lq)[:, None] + (lk - lq)
            sc = sc.masked_fill(~m, float("-inf"))
        o = torch.softmax(sc, -1) @ vs
        out[cu_q[s]:cu_q[s + 1]] = o.transpose(0, 1).reshape(-1, H * D)
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_tr", [True, False])
def test_attention_llama_causal_gqa(ops, dtype, use_tr):
    H, KV, D = 2, 1, 128
    lens = [150, 40]                                          # 2 q-blocks (ragged) + a short sequence
    cu = [0, 150, 190]
    T = cu[-1]
    qkv = rnd((T, (H + 2 * KV) * D), dtype, 60)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    out = torch.full((T, H * D), float("nan"), dtype=dtype)
    cu_t = torch.tensor(cu, dtype=torch.int32)
    ops.attention(q, k, v, out, cu_t, cu_t, max(lens), H, KV, D, D ** -0.5, True, use_tr)
    ref = attn_ref(q, k, v, cu, cu, H, KV, D, D ** -0.5, True)
    assert (out.float() - ref).abs().max() <= tol(dtype) * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_siglip_noncausal_d72(ops, dtype):
    H, D = 2, 72
    cu = [0, 100, 170]                                        # two "tiles" with ragged key tails
    T = cu[-1]
    qkv = rnd((T, 3 * H * D), dtype, 61)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    out = torch.full((T, H * D), float("nan"), dtype=dtype)
    cu_t = torch.tensor(cu, dtype=torch.int32)
    ops.attention(q, k, v, out, cu_t, cu_t, 100, H, H, D, D ** -0.5, False, True)
    ref = attn_ref(q, k, v, cu, cu, H, H, D, D ** -0.5, False)
    assert (out.float() - ref).abs().max() <= tol(dtype) * 2


def test_attention_decode_shape_bottom_right_causal(ops):
    """len_q = 1 against a longer key cache (the decode step) and a chunked-prefill shape (len_q < len_k)."""
    H, KV, D = 2, 2, 128
    dtype = torch.float16
    k, v = rnd((70, KV * D), dtype, 62), rnd((70, KV * D), dtype, 63)
    for lq in (1, 33):
        q = rnd((lq, H * D), dtype, 64)
        out = torch.empty(lq, H * D, dtype=dtype)
        ops.attention(q, k, v, out, torch.tensor([0, lq], dtype=torch.int32), torch.tensor([0, 70], dtype=torch.int32),
                      lq, H, KV, D, D ** -0.5, True, True)
        ref = attn_ref(q, k, v, [0, lq], [0, 70], H, KV, D, D ** -0.5, True)
        assert (out.float() - ref).abs().max() <= 4e-3


def test_gemm_lds_swizzle_is_conflict_free():
    """The ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH LDS table) must hit 16 distinct 16-byte slots of
    the 256-byte bank row for the fragment reads of gemm.h (rows fr, chunk 2ks+fh, swizzle c ^ ((r>>1)&7))."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for ks in range(4):
        for base in (0, 32, 64, 96):
            for g in groups:
                slots = set()
                for lane in g:
                    r, lc = base + (lane & 31), 2 * ks + (lane >> 5)
                    off = r * 128 + ((lc ^ ((r >> 1) & 7)) << 4)
                    slots.add((off % 256) // 16)
                assert len(slots) == 16


def test_gemm_packed_w_image_is_conflict_free_and_its_dma_is_coalesced():
    """gemm.h gemm_lds_off_wp (W staged from the packed weight order): (a) the ds_read_b128 lane groups hit 16 distinct slots of the bank
    row; (b) every lane quad of an LDS-DMA instruction fetches 64 contiguous bytes and every octet one aligned 128-byte line of the
    packed matrix — the traffic pattern of the row-major image; (c) slot <-> (row, chunk) is a bijection of the 1-KiB piece."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    off_wp = lambda r, lc: (r >> 3) * 1024 + ((((lc ^ ((r >> 3) & 1)) << 3) + (r & 7)) << 4)
    for ks in range(4):
        for base in (0, 32, 64, 96):
            for g in groups:
                assert len({(off_wp(base + (lane & 31), 2 * ks + (lane >> 5)) % 256) // 16 for lane in g}) == 16
    K = 512
    for piece in range(4):                                               # pieces of a pass: rows 8 piece .. + 7
        src, dst = [], set()
        for l in range(64):
            rp = piece * 8 + (l & 7)
            c = (l >> 3) ^ ((rp >> 3) & 1)
            n = 48 + rp                                                  # some tile origin (multiple of 16 rows)
            src.append((n >> 4) * 32 * K + (c >> 2) * 1024 + ((c & 3) * 16 + (n & 15)) * 16)
            assert off_wp(rp, c) == piece * 1024 + l * 16                # LDS-DMA is lane-linear: lane l lands on slot l of the piece
            dst.add((rp, c))
        assert len(dst) == 64
        for q in range(0, 64, 4):
            assert [src[q + i] - src[q] for i in range(4)] == [0, 16, 32, 48] and src[q] % 64 == 0
        for o in range(0, 64, 8):
            assert src[o] % 128 == 0 and src[o + 7] - src[o] == 112


def test_attention_d72_v_image_is_conflict_free():
    """attention.h att_vpos72: ds_read_b64_tr_b16 serves the two 32-lane halves of a wave in one LDS cycle each when the 32
    eight-byte pieces cover the 256-byte bank row exactly.  A half reads, for one 32-wide d-block, the four 16-byte chunks of
    four consecutive key rows (144-byte rows at d = 72); with the per-row chunk permutation those 16 chunks must hit 16
    distinct slots, and the rows' last chunk (d 64..71, shared by the padded lanes through broadcast) four distinct ones."""
    pos = [[(p >> (4 * c)) & 15 for c in range(9)] for p in (0x087654321, 0x876543210, 0x087216543, 0x876105432)]
    inv = [[(p >> (4 * q)) & 15 for q in range(9)] for p in (0x765432108, 0x876543210, 0x763210548, 0x876321054)]
    for r4 in range(4):
        assert sorted(pos[r4]) == list(range(9)) and all(inv[r4][pos[r4][c]] == c for c in range(9))
    for r0 in range(0, 64, 4):                       # every 4-row group of a 64-key tile
        for db in range(3):
            slots = [((r0 + j) * 144 + pos[j][min(c, 8)] * 16) // 16 % 16 for j in range(4) for c in range(4 * db, 4 * db + 4)]
            assert len(set(slots)) == (16 if db < 2 else 4)


def test_gemm_epilogue_image_reads_are_conflict_free():
    """gemm.h gemm_epilogue, plain epilogues at 64-column wave tiles: lane l reads two 16-byte pieces of row l/8 of the fp32
    image (row stride 272 bytes) at column group ((l + 7*((l/8 >> 1) & 1)) & 7); each ds_read_b128 lane group must hit 16
    distinct 16-byte slots (without the rotation rows two apart collide)."""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    RS = 64 * 4 + 16
    for it in range(4):                              # 8 rows per instruction, 32 rows per image
        for half in (0, 16):
            for g in groups:
                slots = set()
                for lane in g:
                    r_in = lane // 8
                    oc = ((lane + 7 * ((r_in >> 1) & 1)) & 7) * 8
                    slots.add(((it * 8 + r_in) * RS + oc * 4 + half) % 256 // 16)
                assert len(slots) == 16
    # write side: ds_write_b128 is served in contiguous 8-lane groups over 32 banks: rows fr..fr+7 at one column
    for fr0 in range(0, 32, 8):
        assert len({((fr0 + i) * RS) % 128 // 16 for i in range(8)}) == 8


def test_attention_sliding_window(ops):
    """Mistral window: query i sees keys j with i - j < window (bottom-right aligned)."""
    H, KV, D, S, Wn = 2, 1, 128, 150, 40
    dtype = torch.float16
    qkv = rnd((S, (H + 2 * KV) * D), dtype, 65)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    out = torch.empty(S, H * D, dtype=dtype)
    cu = torch.tensor([0, S], dtype=torch.int32)
    ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True, window=Wn)
    qs = q.float().view(S, H, D).transpose(0, 1)
    ks = k.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H, 0)
    vs = v.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H, 0)
    i, j = torch.arange(S)[:, None], torch.arange(S)[None, :]
    sc = (qs @ ks.transpose(-1, -2) * D ** -0.5).masked_fill(~((j <= i) & (i - j < Wn)), float("-inf"))
    ref = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(S, H * D)
    assert (out.float() - ref).abs().max() <= 4e-3


@pytest.mark.parametrize("use_tr", [True, False])
def test_attention_perceiver_d96_cross(ops, use_tr):
    """Idefics2 perceiver shape: 64 latent queries against [context; latents] keys, head_dim 96, GQA 4:1, non-causal."""
    H, KV, D = 4, 1, 96
    dtype = torch.float16
    cu_q, cu_k = [0, 64, 128], [0, 64 + 75, 64 + 75 + 64 + 130]
    q = rnd((128, H * D), dtype, 66)
    kv = rnd((cu_k[-1], 2 * KV * D), dtype, 67)
    out = torch.full((128, H * D), float("nan"), dtype=dtype)
    ops.attention(q, kv[:, :KV * D], kv[:, KV * D:], out, torch.tensor(cu_q, dtype=torch.int32), torch.tensor(cu_k, dtype=torch.int32),
                  64, H, KV, D, D ** -0.5, False, use_tr)
    ref = attn_ref(q, kv[:, :KV * D], kv[:, KV * D:], cu_q, cu_k, H, KV, D, D ** -0.5, False)
    assert (out.float() - ref).abs().max() <= 4e-3


def test_gemm_addmat_row_index(ops):
    """NaViT position ids: row m adds addmat[add_rows[m]]."""
    M, N, K = 50, 128, 64
    a, w = rnd((M, K), torch.float16, 90), rnd((N, K), torch.float16, 91, 0.1)
    table = rnd((16, N), torch.float32, 92)
    idx = torch.randint(0, 16, (M,), generator=torch.Generator().manual_seed(93)).to(torch.int32)
    out = torch.empty(M, N)
    ops.gemm(a, w, out, addmat=table, add_rows=idx, epilogue=_lib.EPI_STORE_F32)
    assert (out - (a.float() @ w.float().T + table[idx.long()])).abs().max() <= 1e-4


def test_preprocess_rectangular_image(ops):
    n, H, W, P, ldo = 2, 45, 61, 14, 640                       # 3 x 4 patches, remainder pixels dropped
    pix = rnd((n, 3, H, W), torch.float32, 94)
    out = torch.full((n * 12, ldo), 7.0, dtype=torch.float16)
    ops.preprocess_images(pix.contiguous(), out, P)
    ref = torch.nn.functional.unfold(pix[:, :, :42, :56], kernel_size=P, stride=P).transpose(1, 2).reshape(n * 12, 588)
    assert torch.equal(out[:, :588].float(), ref.to(torch.float16).float()) and out[:, 588:].abs().max() == 0
    u8 = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (1, H, W, 3), dtype=np.uint8))
    out2 = torch.empty(12, ldo, dtype=torch.float16)
    ops.preprocess_images(u8, out2, P)
    from leopard_amd.tiler import siglip_normalize
    ref2 = torch.nn.functional.unfold(torch.from_numpy(siglip_normalize(u8.numpy()))[:, :, :42, :56], kernel_size=P, stride=P)
    assert torch.equal(out2[:, :588].float(), ref2.transpose(1, 2).reshape(12, 588).to(torch.float16).float())


def test_gemm_tile_order_round_robin(ops):
    """gemm.order = 1 (XCDs take 32-tile patches round-robin; surplus workgroups of the rounded-up grid exit)."""
    ops.set_option("gemm.order", 1)
    try:
        for cfg in (0, 7):
            ops.set_option("gemm.config", cfg)
            M, N, K = 700, 640, 128
            a, w = rnd((M, K), torch.float16, 95), rnd((N, K), torch.float16, 96, 0.1)
            out = torch.full((M, N), float("nan"), dtype=torch.float16)
            ops.gemm(a, w, out)
            ref = a.float() @ w.float().T
            assert (out.float() - ref).abs().max() <= 2e-3 * max(1.0, ref.abs().max().item())
    finally:
        ops.set_option("gemm.order", 0)
        ops.set_option("gemm.config", -1)


@pytest.mark.parametrize("dma", [1, 0])
def test_attention_kernel_variants(ops, dma):
    """Both attention kernels behind lmi_attn_varlen_fwd (LDS-DMA = production, register-staged = cross-check) on a causal
    GQA case with ragged blocks and on the 72-wide SigLIP head."""
    ops.set_option("attn.dma", dma)
    try:
        H, KV, D = 2, 1, 128
        cu = [0, 300, 341]
        T = cu[-1]
        qkv = rnd((T, (H + 2 * KV) * D), torch.float16, 70)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
        out = torch.full((T, H * D), float("nan"), dtype=torch.float16)
        cu_t = torch.tensor(cu, dtype=torch.int32)
        ops.attention(q, k, v, out, cu_t, cu_t, 300, H, KV, D, D ** -0.5, True, True)
        ref = attn_ref(q, k, v, cu, cu, H, KV, D, D ** -0.5, True)
        assert (out.float() - ref).abs().max() <= 4e-3
        H, D = 2, 72
        cu = [0, 270]
        qkv = rnd((270, 3 * H * D), torch.float16, 71)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
        out = torch.full((270, H * D), float("nan"), dtype=torch.float16)
        cu_t = torch.tensor(cu, dtype=torch.int32)
        ops.attention(q, k, v, out, cu_t, cu_t, 270, H, H, D, D ** -0.5, False, True)
        ref = attn_ref(q, k, v, cu, cu, H, H, D, D ** -0.5, False)
        assert (out.float() - ref).abs().max() <= 4e-3
    finally:
        ops.set_option("attn.dma", 1)



def test_rope_at_device_position_matches_host_position(ops):
    """lmi_rope_qk_at (position from device memory, full tables) = lmi_rope_qk (host position, table slice)."""
    from oracle.leopard_oracle import rope_tables
    from leopard_amd.config import RopeScaling
    nq, nkv, D, cap = 4, 2, 128, 40
    cos, sin = rope_tables(torch.arange(cap), D, 5e5, RopeScaling())
    cos, sin = cos[:, :D // 2].contiguous(), sin[:, :D // 2].contiguous()
    for pos, S in ((0, 1), (17, 1), (30, 3)):
        qkv = rnd((S, (nq + 2 * nkv) * D), torch.float16, 80 + pos)
        a, b = qkv.clone(), qkv.clone()
        kc1, vc1 = torch.zeros(cap, nkv * D, dtype=torch.float16), torch.zeros(cap, nkv * D, dtype=torch.float16)
        kc2, vc2 = torch.zeros_like(kc1), torch.zeros_like(vc1)
        ops.rope_qk(a, nq, nkv, D, cos[pos:pos + S].contiguous(), sin[pos:pos + S].contiguous(), kc1, vc1, cache_pos0=pos)
        ops.rope_qk_at(b, nq, nkv, D, cos, sin, kc2, vc2, torch.tensor([pos], dtype=torch.int32))
        assert torch.equal(a, b) and torch.equal(kc1, kc2) and torch.equal(vc1, vc2)


@pytest.mark.parametrize("lq,lk,cap", [(1, 70, 70), (1, 700, 2048), (5, 1300, 1300), (1, 1, 64)])
def test_attention_decode_split_kv(ops, lq, lk, cap):
    """Split-KV decode attention (partials + merge) = the reference attention, for a launch geometry fixed by `cap`."""
    H, KV, D = 4, 2, 128
    dtype = torch.float16
    q = rnd((lq, H * D), dtype, 90)
    k, v = rnd((lk, KV * D), dtype, 91), rnd((lk, KV * D), dtype, 92)
    out = torch.full((lq, H * D), float("nan"), dtype=dtype)
    ws = torch.empty(ops.decode_workspace_elems(lq, H, D, cap), dtype=torch.float32)
    cu_q, cu_k = torch.tensor([0, lq], dtype=torch.int32), torch.tensor([0, lk], dtype=torch.int32)
    ops.attention_decode(q, k, v, out, cu_q, cu_k, lq, cap, H, KV, D, D ** -0.5, ws)
    ref = attn_ref(q, k, v, [0, lq], [0, lk], H, KV, D, D ** -0.5, True)
    assert (out.float() - ref).abs().max() <= 4e-3
    one = torch.empty_like(out)
    ops.attention(q, k, v, one, cu_q, cu_k, lq, H, KV, D, D ** -0.5, True, True)
    assert (out.float() - one.float()).abs().max() <= 2e-3


@pytest.mark.parametrize("K,N", [(4096, 200), (14336, 70)])
def test_gemv_split_k_kernel(ops, K, N):
    """The decode GEMV that splits K over the waves of a workgroup (hidden sizes 4096 / 14336), every epilogue."""
    dtype = torch.float16
    w, x = rnd((N, K), dtype, 100, 0.05), rnd((K,), dtype, 101)
    ref = w.float() @ x.float()
    out = torch.empty(N)
    ops.gemv(w, x, out)
    assert ((out - ref).abs() / (1 + ref.abs())).max() <= 2e-3
    o16 = torch.empty(N, dtype=dtype)
    ops.gemv(w, x, o16, epilogue=1)
    assert ((o16.float() - ref).abs() / (1 + ref.abs())).max() <= 4e-3
    acc = torch.ones(N)
    ops.gemv(w, x, acc, epilogue=2)
    assert ((acc - 1 - ref).abs() / (1 + ref.abs())).max() <= 2e-3
    if K == 4096:
        N2 = 192
        w2 = rnd((N2, K), dtype, 102, 0.05)
        r2 = w2.float() @ x.float()
        wi = torch.stack([w2[:96].view(3, 32, K), w2[96:].view(3, 32, K)], dim=1).reshape(N2, K).contiguous()
        sw = torch.empty(96, dtype=dtype)
        ops.gemv(wi, x, sw, epilogue=3)
        r = torch.nn.functional.silu(r2[:96]) * r2[96:]
        assert ((sw.float() - r).abs() / (1 + r.abs())).max() <= 4e-3


def test_gemv_rmsnorm_equals_rmsnorm_then_gemv(ops):
    dtype = torch.float16
    K, N = 4096, 136
    w = rnd((N, K), dtype, 110, 0.05)
    x = rnd((1, K), torch.float32, 111, 3.0)
    g = rnd((K,), torch.float32, 112).abs() + 0.5
    h = torch.empty(1, K, dtype=dtype)
    ops.rmsnorm(x, g, h, 1e-5)
    a, b = torch.empty(N, dtype=dtype), torch.empty(N, dtype=dtype)
    ops.gemv(w, h[0], a, epilogue=1)
    ops.gemv_rmsnorm(w, x[0], g, 1e-5, b, epilogue=1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N,n_rows", [(4096, 300, 3), (128, 256, 1), (64, 1000, 2)])
def test_lm_head_last_keeps_the_normalised_row_in_fp32(ops, dtype, K, N, n_rows):
    """lmi_lm_head_last: out = W . rmsnorm(x[rows]) with fp32 activations — exact to fp32 summation order against the fp32
    definition (tolerance far below one 16-bit rounding of the normalised row), ragged N, row selection, and no-norm mode."""
    g = torch.Generator().manual_seed(K + N)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    x = torch.randn(9, K, generator=g) * 3
    gamma = torch.rand(K, generator=g) + 0.5
    rows = torch.tensor([7, 0, 4][:n_rows], dtype=torch.int64)
    out = torch.full((n_rows, N + 8), 7.0)
    ops.lm_head_last(w, x, rows, gamma, 1e-5, out[:, :N])
    xn = gamma * (x[rows] * torch.rsqrt(x[rows].pow(2).mean(-1, keepdim=True) + 1e-5))
    ref = xn.double() @ w.double().T
    assert (out[:, :N].double() - ref).abs().max() <= 2e-5 * ref.abs().max()
    assert torch.all(out[:, N:] == 7.0)
    out2 = torch.empty(n_rows, N)
    ops.lm_head_last(w, x, None, None, 0.0, out2)                     # rows 0..n-1, no normalisation
    ref2 = x[:n_rows].double() @ w.double().T
    assert (out2.double() - ref2).abs().max() <= 2e-5 * ref2.abs().max()


@pytest.mark.parametrize("dtype_pair", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 32, 128), (5, 48, 512), (8, 64, 1792), (16, 16, 1152 + 128)])
def test_gemm_skinny_all_epilogues(ops, dtype_pair, M, N, K):
    """lmi_gemm_skinny (M <= 16 rows against streamed weights: batched decode): every epilogue vs fp32, K steps that do not divide
    evenly over the 8 waves, rows >= M never touched."""
    dtype = dtype_pair
    x, w = rnd((M, K), dtype, 3), rnd((N, K), dtype, 4, 0.1)
    ref = x.float() @ w.float().T
    t = tol(dtype) * max(1.0, ref.abs().max().item())
    out = torch.full((M + 1, N), 7.0, dtype=dtype)
    ops.gemm_skinny(w, x, out[:M], 0)
    assert (out[:M].float() - ref).abs().max() <= t and bool((out[M] == 7.0).all())
    o32 = torch.zeros(M, N)
    ops.gemm_skinny(w, x, o32, 3)
    assert (o32 - ref).abs().max() <= 1e-3 * max(1.0, ref.abs().max().item())
    acc = rnd((M, N), torch.float32, 5)
    acc0 = acc.clone()
    ops.gemm_skinny(w, x, acc, 1)
    assert (acc - (acc0 + ref)).abs().max() <= 1e-3 * max(1.0, ref.abs().max().item())
    from leopard_amd.weights import skinny_pack
    wp = skinny_pack(w)                                            # the packed (MFMA operand order) copy gives the same bits
    outp = torch.zeros(M, N, dtype=dtype)
    ops.gemm_skinny(wp, x, outp, 0, packed=True)
    assert torch.equal(outp, out[:M])
    if N % 64 == 0:
        F = N // 2
        lv = ref.view(M, N // 64, 2, 32)
        want = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, F)
        o = torch.zeros(M, F, dtype=dtype)
        ops.gemm_skinny(w, x, o, 2)
        assert (o.float() - want).abs().max() <= tol(dtype) * max(1.0, want.abs().max().item())
        op = torch.zeros(M, F, dtype=dtype)
        ops.gemm_skinny(wp, x, op, 2, packed=True)
        assert torch.equal(op, o)


def test_rope_rows_equals_rope_at_per_row(ops):
    """lmi_rope_qk_rows (row s at its own device position, K / V into slot s of a pooled cache) == lmi_rope_qk_at row by row."""
    dtype, H, KV, hd, cap, B = torch.float16, 2, 1, 128, 12, 3
    qkv = rnd((B, (H + 2 * KV) * hd), dtype, 1)
    f = torch.arange(cap).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = torch.tensor([4, 0, 11], dtype=torch.int32)
    kp, vp = torch.zeros(B * cap, KV * hd, dtype=dtype), torch.zeros(B * cap, KV * hd, dtype=dtype)
    got = qkv.clone()
    ops.rope_qk_rows(got, H, KV, hd, cos, sin, kp, vp, cap, pos)
    for s in range(B):
        one = qkv[s:s + 1].clone()
        kc, vc = torch.zeros(cap, KV * hd, dtype=dtype), torch.zeros(cap, KV * hd, dtype=dtype)
        ops.rope_qk_at(one, H, KV, hd, cos, sin, kc, vc, pos[s:s + 1].clone())
        assert torch.equal(one[0], got[s])
        assert torch.equal(kp[s * cap:(s + 1) * cap], kc) and torch.equal(vp[s * cap:(s + 1) * cap], vc)


@pytest.mark.parametrize("packed", [False, True])
def test_rope_qkv_skinny_equals_projection_then_rope_rows(ops, packed):
    """lmi_rope_qkv_skinny (q|k|v projection with RoPE + KV append in the epilogue, rope-permuted weight rows) vs the fp32 statement
    of projection -> rotate-half RoPE at each row's own position; the V columns and the pooled caches; within one 16-bit rounding
    of lmi_gemm_skinny + lmi_rope_qk_rows (which rounds the projection before rotating)."""
    from leopard_amd.weights import rope_permute_rows, skinny_pack
    dtype, H, KV, hd, cap, B, K = torch.float16, 2, 1, 128, 12, 3, 384
    w, x = rnd(((H + 2 * KV) * hd, K), dtype, 7, 0.1), rnd((B, K), dtype, 8)
    f = torch.arange(cap).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = torch.tensor([4, 0, 11], dtype=torch.int32)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    kp, vp = torch.zeros(B * cap, KV * hd, dtype=dtype), torch.zeros(B * cap, KV * hd, dtype=dtype)
    got = torch.full((B + 1, (H + 2 * KV) * hd), 7.0, dtype=dtype)
    ops.rope_qkv_skinny(skinny_pack(w_rope) if packed else w_rope, x, got[:B], H, KV, hd, cos, sin, kp, vp, cap, pos, packed=packed)
    assert bool((got[B] == 7.0).all())
    lin = x.float() @ w.float().T
    ref = lin.clone()
    for s in range(B):
        c, sn = cos[pos[s]], sin[pos[s]]
        for h in range(H + KV):
            a, b = lin[s, h * hd:h * hd + 64], lin[s, h * hd + 64:(h + 1) * hd]
            ref[s, h * hd:h * hd + 64] = a * c - b * sn
            ref[s, h * hd + 64:(h + 1) * hd] = b * c + a * sn
    t = tol(dtype) * max(1.0, ref.abs().max().item())
    assert (got[:B].float() - ref).abs().max() <= t
    two = torch.zeros(B, (H + 2 * KV) * hd, dtype=dtype)
    k2, v2 = torch.zeros_like(kp), torch.zeros_like(vp)
    ops.gemm_skinny(w, x, two, 0)
    ops.rope_qk_rows(two, H, KV, hd, cos, sin, k2, v2, cap, pos)
    assert (got[:B].float() - two.float()).abs().max() <= 2 * t
    assert torch.equal(got[:B, (H + KV) * hd:], two[:, (H + KV) * hd:])                # V: no rotation, the same rounding
    for s in range(B):
        r = s * cap + int(pos[s])
        assert torch.equal(kp[r], got[s, H * hd:(H + KV) * hd]) and torch.equal(vp[r], got[s, (H + KV) * hd:])
    touched = torch.zeros(B * cap, dtype=torch.bool)
    touched[[s * cap + int(pos[s]) for s in range(B)]] = True
    assert not kp[~touched].any() and not vp[~touched].any()


@pytest.mark.parametrize("packed,M,D", [(False, 5, 256), (True, 5, 256), (True, 16, 640), (False, 1, 384)])
def test_gemm_skinny_folded_rmsnorm_producer_and_consumers(ops, packed, M, D):
    """lmi_gemm_skinny_ex: the residual projection also emits T(x * gamma) and per-16-column sums of squares (producer); the SwiGLU
    projection and lmi_rope_qkv_skinny scale their accumulator rows by rstd from those partials (consumers) — against the norm launch
    followed by the plain projections, and against fp32."""
    from leopard_amd.weights import rope_permute_rows, skinny_pack
    dtype, K0, eps = torch.float16, 384, 1e-5                        # D / 16 partials per row: fewer than, equal to and more than the 32 lanes that sum them
    pk = skinny_pack if packed else (lambda w: w)
    a, w_o = rnd((M, K0), dtype, 1), rnd((D, K0), dtype, 2, 0.1)
    x0, gamma = rnd((M, D), torch.float32, 3, 2.0), rnd((D,), torch.float32, 4) + 1.0
    # producer
    x = x0.clone()
    h = torch.full((M + 1, D), 7.0, dtype=dtype)
    sq = torch.full((M + 1, D // 16), 7.0)
    ops.gemm_skinny(pk(w_o), a, x, 1, packed, norm_out=h[:M], norm_gamma=gamma, rowsq_out=sq[:M])
    x_ref = x0.clone()
    ops.gemm_skinny(pk(w_o), a, x_ref, 1, packed)
    assert torch.equal(x, x_ref)                                     # the residual itself is untouched by the extra outputs
    assert torch.equal(h[:M], (x * gamma).to(dtype)) and bool((h[M] == 7.0).all()) and bool((sq[M] == 7.0).all())
    want_sq = (x.double() ** 2).view(M, D // 16, 16).sum(-1)
    assert (sq[:M].double() - want_sq).abs().max() <= 1e-5 * want_sq.abs().max()
    rstd = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + eps)
    # consumer, SwiGLU
    F = 128
    w_gu = rnd((2 * F, D), dtype, 5, 0.1)
    gu = torch.zeros(M, F, dtype=dtype)
    ops.gemm_skinny(pk(w_gu), h[:M], gu, 2, packed, rowsq_in=sq[:M], norm_dim=D, norm_eps=eps)
    lin = (h[:M].double() @ w_gu.double().T) * rstd
    lv = lin.view(M, 2 * F // 64, 2, 32)
    want = (torch.nn.functional.silu(lv[:, :, 0]) * lv[:, :, 1]).reshape(M, F)
    assert (gu.double() - want).abs().max() <= tol(dtype) * max(1.0, want.abs().max().item())
    hn = torch.zeros(M, D, dtype=dtype)                               # the unfused schedule: norm launch, plain projection
    ops.rmsnorm(x, gamma, hn, eps)
    gu2 = torch.zeros(M, F, dtype=dtype)
    ops.gemm_skinny(pk(w_gu), hn, gu2, 2, packed)
    assert (gu.float() - gu2.float()).abs().max() <= 2 * tol(dtype) * max(1.0, want.abs().max().item())
    # consumer, q|k|v + RoPE
    H, KV, hd, cap = 2, 1, 128, 12
    w = rnd(((H + 2 * KV) * hd, D), dtype, 6, 0.1)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    f = torch.arange(cap).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = torch.tensor([4, 0, 11, 3, 7, 1, 2, 5, 6, 8, 9, 10, 0, 11, 4, 3][:M], dtype=torch.int32)
    outs = []
    for xin, rs in ((h[:M], sq[:M]), (hn, None)):
        kp, vp = torch.zeros(M * cap, KV * hd, dtype=dtype), torch.zeros(M * cap, KV * hd, dtype=dtype)
        got = torch.zeros(M, (H + 2 * KV) * hd, dtype=dtype)
        ops.rope_qkv_skinny(pk(w_rope), xin, got, H, KV, hd, cos, sin, kp, vp, cap, pos, packed=packed, rowsq_in=rs, norm_eps=eps)
        outs.append((got, kp, vp))
    scale = max(1.0, outs[1][0].float().abs().max().item())
    for g, u in zip(outs[0], outs[1]):
        assert (g.float() - u.float()).abs().max() <= 2 * tol(dtype) * scale
    lin = ((h[:M].double() @ w.double().T) * rstd).float()
    ref = lin.clone()
    for s_ in range(M):
        c, sn = cos[pos[s_]], sin[pos[s_]]
        for hh in range(H + KV):
            a_, b_ = lin[s_, hh * hd:hh * hd + 64], lin[s_, hh * hd + 64:(hh + 1) * hd]
            ref[s_, hh * hd:hh * hd + 64] = a_ * c - b_ * sn
            ref[s_, hh * hd + 64:(hh + 1) * hd] = b_ * c + a_ * sn
    assert (outs[0][0].float() - ref).abs().max() <= tol(dtype) * scale


def test_gemv_rmsnorm_rope_equals_norm_projection_rope(ops):
    """lmi_gemv_rmsnorm_rope (batch-1 decode: RMSNorm + q|k|v projection + RoPE + KV append in one launch) vs the fp32 statement, and
    within one 16-bit rounding of lmi_gemv_rmsnorm + lmi_rope_qk_at; cache rows other than *pos untouched."""
    from leopard_amd.weights import rope_permute_rows
    dtype, H, KV, hd, cap, K = torch.float16, 2, 1, 128, 9, 4096
    N = (H + 2 * KV) * hd
    w, x, g = rnd((N, K), dtype, 7, 0.03), rnd((1, K), torch.float32, 8, 2.0), rnd((K,), torch.float32, 9) + 1.0
    f = torch.arange(cap).float().reshape(-1, 1) * (1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))).reshape(1, -1)
    cos, sin = f.cos().contiguous(), f.sin().contiguous()
    pos = torch.tensor([6], dtype=torch.int32)
    w_rope = torch.cat([rope_permute_rows(w[:(H + KV) * hd]), w[(H + KV) * hd:]]).contiguous()
    kc, vc = torch.zeros(cap, KV * hd, dtype=dtype), torch.zeros(cap, KV * hd, dtype=dtype)
    got = torch.zeros(1, N, dtype=dtype)
    ops.gemv_rmsnorm_rope(w_rope, x[0], g, 1e-5, got[0], H, KV, hd, cos, sin, kc, vc, pos)
    xn = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * g).to(dtype).float()
    lin = xn @ w.float().T
    ref = lin.clone()
    c, sn = cos[6], sin[6]
    for h in range(H + KV):
        a, b = lin[0, h * hd:h * hd + 64], lin[0, h * hd + 64:(h + 1) * hd]
        ref[0, h * hd:h * hd + 64] = a * c - b * sn
        ref[0, h * hd + 64:(h + 1) * hd] = b * c + a * sn
    t = tol(dtype) * max(1.0, ref.abs().max().item())
    assert (got.float() - ref).abs().max() <= t
    two = torch.zeros(1, N, dtype=dtype)
    k2, v2 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.gemv_rmsnorm(w, x[0], g, 1e-5, two[0], epilogue=1)
    ops.rope_qk_at(two, H, KV, hd, cos, sin, k2, v2, pos)
    assert (got.float() - two.float()).abs().max() <= 2 * t
    assert torch.equal(got[0, (H + KV) * hd:], two[0, (H + KV) * hd:]) and torch.equal(vc, v2)
    assert torch.equal(kc[6], got[0, H * hd:(H + KV) * hd]) and torch.equal(vc[6], got[0, (H + KV) * hd:])
    kc[6] = 0
    vc[6] = 0
    assert not kc.any() and not vc.any()


@pytest.mark.parametrize("rms", [True, False])
@pytest.mark.parametrize("M,D", [(1, 1152), (3, 4096), (32, 256), (33, 4096)])
def test_norm_small_m_rows_kernel(ops, rms, M, D):
    """RMSNorm / LayerNorm with a handful of rows (M <= 32: one workgroup per row, norm_rows_kernel) and just above (one wave per row)
    vs fp64; rows >= M untouched."""
    dtype = torch.float16
    x, g, b = rnd((M, D), torch.float32, 1, 2.0) + 0.3, rnd((D,), torch.float32, 2) + 1.0, rnd((D,), torch.float32, 3)
    out = torch.full((M + 1, D), 7.0, dtype=dtype)
    xd = x.double()
    if rms:
        ops.rmsnorm(x, g, out[:M], 1e-5)
        ref = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-5) * g.double()
    else:
        ops.layernorm(x, g, b, out[:M], 1e-6)
        ref = torch.nn.functional.layer_norm(xd, (D,), g.double(), b.double(), 1e-6)
    assert (out[:M].double() - ref).abs().max() <= tol(dtype) * max(1.0, ref.abs().max().item())
    assert bool((out[M] == 7.0).all())


def test_attention_decode_pool_equals_per_sequence_decode(ops):
    """lmi_attn_decode_pool (B sequences in slots of one pooled cache, lengths in a device array) == lmi_attn_decode_fwd per sequence."""
    dtype, H, KV, hd, cap = torch.float16, 4, 2, 128, 200
    lens = [130, 1, 77]
    B = len(lens)
    q = rnd((B, H * hd), dtype, 2)
    kp, vp = rnd((B * cap, KV * hd), dtype, 3), rnd((B * cap, KV * hd), dtype, 4)
    ws = torch.zeros(ops.decode_workspace_elems(B, H, hd, cap))
    out = torch.zeros(B, H * hd, dtype=dtype)
    cu_q = torch.arange(B + 1, dtype=torch.int32)
    k_begin = (torch.arange(B, dtype=torch.int32) * cap)
    ops.attention_decode_pool(q, kp, vp, out, cu_q, k_begin, torch.tensor(lens, dtype=torch.int32), cap, H, KV, hd, hd ** -0.5, ws)
    for s, L in enumerate(lens):
        one = torch.zeros(1, H * hd, dtype=dtype)
        ws1 = torch.zeros(ops.decode_workspace_elems(1, H, hd, cap))
        ops.attention_decode(q[s:s + 1], kp[s * cap:(s + 1) * cap], vp[s * cap:(s + 1) * cap], one, torch.tensor([0, 1], dtype=torch.int32),
                             torch.tensor([0, L], dtype=torch.int32), 1, cap, H, KV, hd, hd ** -0.5, ws1)
        assert torch.equal(one[0], out[s]), s


def test_attention_decode_gqa_pack_is_bit_identical_to_head_per_block(ops):
    """Decode blocks packed by kv head (4 waves = the 4 query heads that share it; attn.gqa_pack, the default when n_heads == 4 n_kv_heads)
    give the same bits as one block per query head: the same per-row arithmetic, only the wave that runs it changes."""
    dtype, H, KV, hd, cap = torch.float16, 8, 2, 128, 260
    lens, B = [259, 70], 2
    q = rnd((B, H * hd), dtype, 5)
    kp, vp = rnd((B * cap, KV * hd), dtype, 6), rnd((B * cap, KV * hd), dtype, 7)
    cu_q = torch.arange(B + 1, dtype=torch.int32)
    k_begin = (torch.arange(B, dtype=torch.int32) * cap)
    outs = []
    try:
        for pack in (1, 0):
            ops.set_option("attn.gqa_pack", pack)
            ws = torch.zeros(ops.decode_workspace_elems(B, H, hd, cap))
            out = torch.zeros(B, H * hd, dtype=dtype)
            ops.attention_decode_pool(q, kp, vp, out, cu_q, k_begin, torch.tensor(lens, dtype=torch.int32), cap, H, KV, hd, hd ** -0.5, ws)
            outs.append(out)
    finally:
        ops.set_option("attn.gqa_pack", 1)
    assert torch.equal(outs[0], outs[1])
    for s, L in enumerate(lens):                                   # and both are right
        qs = q[s].float().view(H, hd)
        ks = kp[s * cap:s * cap + L].float().view(L, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        vs = vp[s * cap:s * cap + L].float().view(L, KV, hd).transpose(0, 1).repeat_interleave(H // KV, 0)
        ref = (torch.softmax((qs[:, None, :] @ ks.transpose(-1, -2)) * hd ** -0.5, -1) @ vs).reshape(-1)
        assert (outs[0][s].float() - ref).abs().max() <= 3e-3



@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_split_hi_lo_and_fp32_producers(ops, dtype):
    """lmi_split_hi_lo: [hi | lo] with hi = T(x), lo = T(x - hi) bit for bit; a GEMM over [hi | lo] against [W | W] equals the fp32
    product to ~2^-19; the fp32 hand-over forms of the producers: attention (lmi_attn_varlen_fwd_f32), fc1 + GELU -> fp32, gate/up -> fp32."""
    M, K, N = 70, 128, 128
    x = rnd((M, K), torch.float32, 51)
    out = torch.zeros(M, 2 * K, dtype=dtype)
    ops.split_hi_lo(x, out)
    hi = x.to(dtype)
    assert torch.equal(out[:, :K], hi) and torch.equal(out[:, K:], (x - hi.float()).to(dtype))
    w = rnd((N, K), dtype, 52, 0.1)
    y = torch.zeros(M, N)
    ops.gemm(out, torch.cat([w, w], dim=1).contiguous(), y, epilogue=_lib.EPI_STORE_F32)
    ref = x.double() @ w.double().T
    plain = torch.zeros(M, N)
    ops.gemm(hi, w, plain, epilogue=_lib.EPI_STORE_F32)
    e_split, e_plain = (y.double() - ref).abs().max().item(), (plain.double() - ref).abs().max().item()
    assert e_split <= 0.02 * e_plain + 1e-6, (e_split, e_plain)
    # fc1 + GELU with an fp32 destination == the 16-bit destination before its rounding
    g32 = torch.zeros(M, N)
    ops.gemm(hi, w, g32, act=_lib.ACT_GELU_TANH, epilogue=_lib.EPI_STORE_F32)
    g16 = torch.zeros(M, N, dtype=dtype)
    ops.gemm(hi, w, g16, act=_lib.ACT_GELU_TANH)
    assert torch.equal(g32.to(dtype), g16)
    # gate/up + SwiGLU with an fp32 destination
    s32 = torch.zeros(M, N // 2)
    ops.gemm(hi, w, s32, epilogue=_lib.EPI_SWIGLU_F32)
    s16 = torch.zeros(M, N // 2, dtype=dtype)
    ops.gemm(hi, w, s16, epilogue=_lib.EPI_SWIGLU)
    assert torch.equal(s32.to(dtype), s16)
    # attention with an fp32 destination
    H, hd, S = 2, 128, 70
    qkv = rnd((S, 3 * H * hd), dtype, 53)
    cu = torch.tensor([0, 30, S], dtype=torch.int32)
    o16 = torch.zeros(S, H * hd, dtype=dtype)
    ops.attention(qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:], o16, cu, cu, 40, H, H, hd, hd ** -0.5, True, True)
    o32 = torch.zeros(S, H * hd)
    ops.attention_f32out(qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:], o32, cu, cu, 40, H, H, hd, hd ** -0.5, True)
    assert torch.equal(o32.to(dtype), o16)


# ---- attention64.h: 64 query rows per wave, 32-key software-pipelined steps (the long-prefill kernel at head_dim 128) -----------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("variant", [1, 2])
def test_attention_rows64_vs_fp32(ops, dtype, causal, variant):
    """attn_fwd_r64_kernel against fp32 on ragged shapes: several 256-row workgroups with a tail block (waves with no rows, waves with
    fewer tiles than the workgroup), a sequence shorter than one wave, one of exactly one tile, GQA 2:1; every output row is written."""
    H, KV, D = 2, 1, 128
    lens = [600, 40, 64, 257]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    T = cu[-1]
    qkv = rnd((T, (H + 2 * KV) * D), dtype, 160)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KV) * D], qkv[:, (H + KV) * D:]
    cu_t = torch.tensor(cu, dtype=torch.int32)
    ref = attn_ref(q, k, v, cu, cu, H, KV, D, D ** -0.5, causal)
    ops.set_option("attn.rows64_min", 0)
    try:
        ops.set_option("attn.rows64", variant)                     # 1: two 32-row blocks per wave; 2: one block per wave, 8 waves
        out = torch.full((T, H * D), float("nan"), dtype=dtype)
        ops.attention(q, k, v, out, cu_t, cu_t, max(lens), H, KV, D, D ** -0.5, causal, True)
        ops.set_option("attn.rows64", 0)
        base = torch.full((T, H * D), float("nan"), dtype=dtype)
        ops.attention(q, k, v, base, cu_t, cu_t, max(lens), H, KV, D, D ** -0.5, causal, True)
    finally:
        ops.set_option("attn.rows64", 0)
        ops.set_option("attn.rows64_min", 1024)
    assert torch.isfinite(out.float()).all()
    err, err_base = (out.float() - ref).abs().max(), (base.float() - ref).abs().max()
    assert err <= tol(dtype) * 2, (err, err_base)


@pytest.mark.parametrize("variant", [1, 2])
def test_attention_rows64_sliding_window_and_reference_moves(ops, variant):
    """The deferred softmax reference under the pipelined schedule: scores that grow by far more than 2^8 along the keys (the reference
    must move several times, each time flushing the pending probabilities first), and the Mistral sliding window (leading tiles fully
    masked for late rows: the reference stays -inf there)."""
    H, KV, D, S, Wn = 2, 1, 128, 700, 300
    dtype = torch.float16
    qkv = rnd((S, (H + 2 * KV) * D), dtype, 161)
    q, k, v = qkv[:, :H * D].clone(), qkv[:, H * D:(H + KV) * D].clone(), qkv[:, (H + KV) * D:].clone()
    k *= torch.linspace(0.2, 6.0, S).reshape(S, 1).to(dtype)       # later keys score higher and higher: frequent reference moves
    cu = torch.tensor([0, S], dtype=torch.int32)
    qs = q.float().view(S, H, D).transpose(0, 1)
    ks = k.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H, 0)
    vs = v.float().view(S, KV, D).transpose(0, 1).repeat_interleave(H, 0)
    i, j = torch.arange(S)[:, None], torch.arange(S)[None, :]
    ops.set_option("attn.rows64_min", 0)
    ops.set_option("attn.rows64", variant)
    try:
        for window in (0, Wn):
            out = torch.full((S, H * D), float("nan"), dtype=dtype)
            ops.attention(q, k, v, out, cu, cu, S, H, KV, D, D ** -0.5, True, True, window=window)
            vis = (j <= i) & ((i - j < window) if window else torch.ones_like(j <= i))
            sc = (qs @ ks.transpose(-1, -2) * D ** -0.5).masked_fill(~vis, float("-inf"))
            ref = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(S, H * D)
            assert torch.isfinite(out.float()).all()
            assert (out.float() - ref).abs().max() <= 6e-3, (window, (out.float() - ref).abs().max())
            if variant == 2 and window == 0:                       # masks before the maxima: a row that sees one key returns its V row exactly
                assert torch.equal(out[0].view(H, D), v[0].view(KV, D).repeat_interleave(H // KV, 0))
    finally:
        ops.set_option("attn.rows64", 0)
        ops.set_option("attn.rows64_min", 1024)


def test_prefill_workspace_api(ops):
    """SURVEY.md 8b "caller owns every buffer incl. workspace (size from lmi_*_workspace_bytes)": the two prefill-stage size functions return
    256-byte aligned, non-overlapping buffers in the LMI_WS_* order whose sizes are the activation buffers of one pass, and reject bad
    arguments with -1 + lmi_last_error."""
    S, D, H, KV, hd, ff = 7187, 4096, 32, 8, 128, 14336
    total, offs = ops.llm_prefill_workspace(S, D, H, KV, hd, ff, torch.float16)
    sizes = [S * D * 2, S * (H + 2 * KV) * hd * 2, S * H * hd * 2, S * ff * 2, S * (D // 64) * 4, S * (D // 64) * 4]
    assert offs[0] == 0 and all(o % 256 == 0 for o in offs) and total % 256 == 0
    for i in range(6):
        end = offs[i + 1] if i + 1 < 6 else total
        assert sizes[i] <= end - offs[i] < sizes[i] + 256
    M, Dv = 42 * 676, 1152
    total_v, offs_v = ops.vit_workspace(M, Dv, 3456, 4352, torch.bfloat16)
    assert offs_v == sorted(offs_v) and total_v >= M * (2 * Dv + 3456 + 4352) * 2 and total_v < M * (2 * Dv + 3456 + 4352) * 2 + 4 * 256
    assert ops.lib.lmi_llm_prefill_workspace_bytes(-1, D, H, KV, hd, ff, 0, None) == -1 and b"workspace" in ops.lib.lmi_last_error()
    assert ops.lib.lmi_vit_workspace_bytes(10, Dv, 3456, 4352, 2, None) == -1                  # LMI_F32 is not a compute type


def test_decode_advance_argmax_stop_rule_and_history(ops):
    """lmi_decode_advance: per-row argmax (lowest index on ties, suppressed ids excluded), history ring, stop rule (eos ids, budget) and the
    frozen state of stopped sequences, against a plain restatement."""
    B, V, ld, H = 5, 1003, 1024, 3                                  # 250 vector loads + a 3-element tail per row
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(B, ld, generator=g)
    logits[:, V:] = 100.0                                           # padding columns past the vocabulary must be ignored
    logits[1, 17] = logits[1, 400] = 50.0                           # a tie: the lower index wins
    logits[2, 5] = 60.0                                             # suppressed: the runner-up must be chosen
    suppress = torch.tensor([5], dtype=torch.int64)
    want = []
    for b in range(B):
        row = logits[b, :V].clone()
        row[5] = float("-inf")
        want.append(int(row.argmax()))
    want[1] = 17
    tok = torch.zeros(B, dtype=torch.int64)
    pos = torch.tensor([10, 20, 30, 40, 50], dtype=torch.int32)
    k_len = pos + 1
    live = torch.tensor([1, 1, 0, 1, 1], dtype=torch.int32)         # sequence 2 already stopped
    budget = torch.tensor([5, 1, 9, 7, 3], dtype=torch.int32)       # sequence 1 produces its last token now
    eos = torch.tensor([want[3], -1], dtype=torch.int64)            # sequence 3 hits its eos
    hist = torch.full((H, B), -7, dtype=torch.int64)
    hist_pos = torch.tensor([0, 1, 2, 3, 4], dtype=torch.int32)
    ops.decode_advance(logits, V, tok, pos, k_len=k_len, live=live, budget=budget, eos=eos, hist=hist, hist_pos=hist_pos, suppress=suppress)
    assert tok.tolist() == want
    assert live.tolist() == [1, 0, 0, 0, 1] and budget.tolist() == [4, 0, 9, 6, 2]
    assert pos.tolist() == [11, 20, 30, 40, 51] and k_len.tolist() == [12, 21, 31, 41, 52]
    assert hist_pos.tolist() == [1, 2, 3, 4, 5]
    for b in range(B):
        assert int(hist[b % H, b]) == want[b]
    # the batch-1 form: no live / budget / history
    t1, p1, k1 = torch.zeros(1, dtype=torch.int64), torch.tensor([7], dtype=torch.int32), torch.tensor([8], dtype=torch.int32)
    ops.decode_advance(logits[4:5], V, t1, p1, k_len=k1)
    assert int(t1) == int(logits[4, :V].argmax()) and int(p1) == 8 and int(k1) == 9


@pytest.mark.parametrize("cfg", [-1, 0, 3, 6, 10])
def test_gemm_reads_packed_weights_bit_identically(ops, cfg):
    """ldw = LMI_LDW_PACKED(K): the prefill GEMM stages the same LDS image from the operand order the decode kernels stream
    (weights.skinny_pack), so every output bit equals the row-major call's — plain, SwiGLU, folded-norm producer and q|k|v + RoPE epilogues,
    ragged M, an N that is not a multiple of the tile width; and the skinny kernel reads the very same tensor."""
    from leopard_amd.weights import as_packed, as_row_major, interleave_gate_up, is_packed, rope_permute_rows
    dtype = torch.float16
    if cfg >= 0:
        ops.set_option("gemm.config", cfg)
    try:
        M, N, K = 300, 384, 512
        a, w = rnd((M, K), dtype, 170 + cfg), rnd((N, K), dtype, 180 + cfg, 0.1)
        wp = as_packed(w)
        assert is_packed(wp) and not is_packed(w) and torch.equal(as_row_major(wp), w) and not torch.equal(wp, w)
        bias = rnd((N,), torch.float32, 90)
        o1, o2 = torch.full((M, N), float("nan"), dtype=dtype), torch.full((M, N), float("nan"), dtype=dtype)
        ops.gemm(a, w, o1, bias=bias)
        ops.gemm(a, wp, o2, bias=bias)
        assert torch.equal(o1, o2)
        # SwiGLU + consumer side of the folded norm
        F = 192
        wi = interleave_gate_up(rnd((F, K), dtype, 11, 0.2), rnd((F, K), dtype, 12, 0.2))
        sq = rnd((M, K // 64), torch.float32, 5).abs() + 0.5
        g1, g2 = torch.full((M, F), float("nan"), dtype=dtype), torch.full((M, F), float("nan"), dtype=dtype)
        ops.gemm_ex(a, wi, g1, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
        ops.gemm_ex(a, as_packed(wi), g2, epilogue=_lib.EPI_SWIGLU, rowsq_in=sq, norm_dim=K, norm_eps=1e-5)
        assert torch.equal(g1, g2)
        # residual + producer side
        x1 = rnd((M, N), torch.float32, 7)
        x2 = x1.clone()
        gam = rnd((N,), torch.float32, 8)
        h1, h2 = torch.zeros(M, N, dtype=dtype), torch.zeros(M, N, dtype=dtype)
        s1, s2 = torch.zeros(M, N // 64), torch.zeros(M, N // 64)
        ops.gemm_ex(a, w, x1, epilogue=_lib.EPI_RESIDUAL, norm_out=h1, norm_gamma=gam, rowsq_out=s1)
        ops.gemm_ex(a, wp, x2, epilogue=_lib.EPI_RESIDUAL, norm_out=h2, norm_gamma=gam, rowsq_out=s2)
        assert torch.equal(x1, x2) and torch.equal(h1, h2) and torch.equal(s1, s2)
        # q|k|v + RoPE + KV append (2 q heads, 1 kv head of 128)
        H, KV, hd = 2, 1, 128
        wq = rnd(((H + 2 * KV) * hd, K), dtype, 21, 0.1)
        wr = torch.cat([rope_permute_rows(wq[:(H + KV) * hd]), wq[(H + KV) * hd:]], dim=0).contiguous()
        ang = torch.rand(M, hd // 2, generator=torch.Generator().manual_seed(3)) * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        outs = []
        for wt in (wr, as_packed(wr)):
            qkv = torch.full((M, (H + 2 * KV) * hd), float("nan"), dtype=dtype)
            kc, vc = torch.zeros(M + 8, KV * hd, dtype=dtype), torch.zeros(M + 8, KV * hd, dtype=dtype)
            ops.rmsnorm_rope(a, wt, qkv, None, 1e-5, cos, sin, kc, vc, 4, H, KV, hd)
            outs.append((qkv, kc, vc))
        assert all(torch.equal(p, q) for p, q in zip(*outs))
        # the decode kernel streams the same tensor
        xs = rnd((5, K), dtype, 31)
        d1, d2 = torch.zeros(5, N, dtype=dtype), torch.zeros(5, N, dtype=dtype)
        ops.gemm_skinny(w, xs, d1, 0)
        ops.gemm_skinny(wp, xs, d2, 0)
        assert torch.equal(d1, d2)
    finally:
        ops.set_option("gemm.config", -1)
    # rejected: a K that is not a multiple of 128, a stride that is not -K
    with pytest.raises(RuntimeError, match="packed W"):
        bad = torch.zeros(128, 192, dtype=dtype)
        bad._lmi_packed = True
        ops.gemm(torch.zeros(8, 192, dtype=dtype), bad, torch.zeros(8, 128, dtype=dtype))
