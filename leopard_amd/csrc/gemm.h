// gemm.h — the MFMA GEMM family:  out[M,N] = epilogue( A[M,K] . W[N,K]^T )     (nn.Linear layout)
//
// Reference ops served: SigLIP patch-embed (im2col rows), q/k/v/out, fc1/fc2; the
// projector's linear_1 (with the 2x2 pixel-shuffle folded into the A-row gather, EVAL:165-192) and
// linear_2; Llama q/k/v/o, gate/up (SwiGLU fused in the epilogue), down; all-position lm_head.
//
// One kernel template, several tile geometries (GemmCfg): BM x BN output tile, BK = 64, WAVES_M x WAVES_N
// waves, each wave owning a (BM/WAVES_M) x (BN/WAVES_N) sub-tile as MI x NI v_mfma_f32_32x32x16 accumulators.
//   * Both operands are K-contiguous and go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round trip)
//     into a ring of STAGES k-tile slots.  Loads for tile t+STAGES-1 are issued, interleaved between the MFMAs,
//     while tile t is being multiplied; the only synchronisation per k-tile is one counted `s_waitcnt vmcnt(N)`
//     (never 0 in steady state for STAGES >= 3) + one raw s_barrier.
//   * The LDS image is lane-linear per wave instruction (hardware rule), so the bank-conflict swizzle is applied
//     on the SOURCE address and again on the ds_read_b128 address: 16-byte chunk c of row r is stored at chunk
//     c ^ ((r>>1)&7).  With 128-byte rows every 16-lane ds_read_b128 group hits 16 distinct 16-byte slots of the
//     256-byte bank row (checked in tests/test_emu_kernels.py).
//   * The MFMA is issued "swapped": A-operand = W rows (n), B-operand = A rows (m), so that a lane owns ONE
//     output row m and 4 consecutive n per accumulator quad; the fused epilogue (bias, GELU, residual add into the
//     fp32 stream, SwiGLU on interleaved gate/up blocks, position-table add, row scatter) turns each wave's tile
//     through the idle LDS so that every store instruction covers whole row segments (gemm_epilogue).
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs each get a contiguous slab of the tile space, walked in
//     groups of GROUP_M row-tiles so that the k-tiles a slab is working on stay in that XCD's private 4 MiB L2.
// Why big tiles: at full MFMA rate a 128x128 tile needs 64 B/clk/CU of L2->LDS traffic — the whole per-CU vector
// memory path; 256x256 halves it.
// Requirements (met by weight preparation, leopard_amd/weights.py): N % 128 == 0, K % 64 == 0, 16-byte aligned
// rows.  M is arbitrary and N may be a non-multiple of BN (tail rows / columns are clamped on load and masked
// on store).
#pragma once
#include "lmi_device.h"
#include <type_traits>

namespace lmi {

enum { EPI_STORE_T = 0, EPI_RESID_F32 = 1, EPI_STORE_F32 = 2, EPI_SWIGLU_T = 3, EPI_QKV_ROPE_T = 4 };
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2 };
enum { AMODE_PLAIN = 0, AMODE_PIXSHUF = 1 };

struct GemmArgs {
    const void* A;        // [M, K] (plain) or ViT output [tiles*G*G, C] (pixel-shuffle mode)
    const void* W;        // [N, K]
    void* out;            // T or fp32, leading dimension ldo
    const float* bias;    // [N] or null (for SwiGLU: unused)
    const float* addmat;  // optional [add_period, N] fp32 matrix added row-periodically (SigLIP pos-emb) ...
    const int* add_rows;  // ... or, when non-null, row add_rows[m] of addmat (NaViT bucketised position ids)
    const int* row_map;   // optional: output row of logical row m (scatter into the merged sequence)
    int M, N, K;
    int lda, ldw, ldo;
    int add_period;
    int ps_grid;          // pixel-shuffle: G (26); output tokens per tile = (G/2)^2; C = K/4
    int group_m;          // row-tiles per L2 group in the XCD-aware tile order
    int order;            // 0 = each XCD owns a contiguous slab of the grouped order; 1 = XCDs take 32-tile chunks round-robin
    unsigned a_bytes, w_bytes;   // extents of A and W (buffer resources of the LDS-DMA; both < 4 GiB)
    int w_packed;         // W is stored in the operand order of lmi_gemm_skinny (weights.skinny_pack) instead of row-major: see GemmStager
    // ---- RMSNorm folded into the GEMMs around it (Llama / Mistral layers) ---------------------------------------------------
    // producer (EPI_RESID_F32): besides x += acc, write norm_out[m, n] = T(x[m, n] * norm_gamma[n]) — the NEXT RMSNorm's gain
    // applied, its row scale still missing — and rowsq_out[m, n / 64] = sum of x[m, n..n+63]^2 (one partial per wave column
    // range: no atomics, bit-reproducible).  consumer (any epilogue): rowsq_in != null -> the accumulator row m is multiplied
    // by rstd[m] = rsqrt(sum_j rowsq_in[m, j] / norm_dim + norm_eps) first, which completes the RMSNorm the producer started:
    // (x * gamma) . W^T * rstd == (gamma * x * rstd) . W^T.
    void* norm_out;
    const float* norm_gamma;
    float* rowsq_out;
    int ld_norm;
    const float* rowsq_in;
    int rowsq_parts;             // partials per row of rowsq_in (= norm_dim / 64)
    int norm_dim;
    float norm_eps;
    // ---- EPI_QKV_ROPE_T: q | k | v projection with RoPE and the KV-cache append in the epilogue ------------------------------
    // W rows of every q / k head are stored in the order d = [0..31, 64..95, 32..63, 96..127] (weights.rope_permute_rows) so that
    // a wave's 64 output columns hold 32 "first half" elements and their 32 rotate-half partners; the epilogue un-permutes on
    // store.  cos / sin: fp32 [M, head_dim / 2] per packed row.  Columns [0, rope_q) are q heads, [rope_q, rope_q + rope_k)
    // k heads, the rest v (copied; appended to v_cache).  k_cache / v_cache rows cache_pos0 + m (nullable).
    const float* rope_cos;
    const float* rope_sin;
    void* k_cache;
    void* v_cache;
    int ld_cache, cache_pos0, rope_q, rope_k;
    // ---- fp8 operands (lmi_gemm_fp8): the accumulators come out multiplied by 2^(scale_e8m0 - 127) (E8M0 block scale of the MFMA,
    // every byte the same), which undoes the power-of-two scales the operands were quantised with
    int swiglu_f32;              // EPI_SWIGLU_T with an fp32 destination (split-operand precision mode: the product goes to lmi_split_hi_lo)
    int scale_e8m0;
    float out_scale;             // fp8 OUTPUTS (T = fp8_t: GELU / SwiGLU results handed to the next fp8 GEMM): value * out_scale, then e4m3
    // ---- low-bit correction phase (LO4 instantiations; DESIGN.md 2.1 "precision mode") ------------------------------------------------------
    // The 16-bit A operand is T(x); the distance of the HIP path from the fp32 reference is that one rounding per hand-over.  The producer
    // also hands over the residual x - T(x) as an MX fp4 image A4 (e2m1, one E8M0 scale per 32 k: a4_scale[m][k / 32]) and the weight has an
    // fp4 image W4 (one E8M0 scale per row: w4_scale[n]); after the K / 64 16-bit k-tiles the SAME accumulators take K4 / 256 k-tiles of
    // v_mfma_scale_f32_32x32x64_f8f6f4 on the two images (4 x the 16-bit rate: + 25 % matrix time), which removes ~80 % of the rounding.
    // Images are row-major, K4 = K rounded up to 256 elements wide (zero codes in the padding), lda4 / ldw4 in BYTES.
    const void* A4;
    const void* W4;
    const uint8_t* a4_scale;
    const uint8_t* w4_scale;
    int lda4, ldw4, lds4, K4;
    unsigned a4_bytes, w4_bytes, a4s_bytes;
    // producer side (LO4 instantiations): the fp4 image + block scales of the residual of THIS launch's 16-bit output — `out` for the
    // STORE (+ activation) and SwiGLU epilogues, `norm_out` for the RESIDUAL producer mode — written next to it (null = not wanted)
    void* out4;
    uint8_t* out4_scale;
    int ld_out4, ld_out4s;       // bytes per row of out4 / of out4_scale
    // ---- row selection of the correction phase (round 6; LeopardEngine.lo4_rows) ---------------------------------------------
    // The logits of a row are dominated by the hand-over roundings of THAT row's own path through the layers; the other rows' roundings reach it
    // only through the softmax average over ~S keys.  So only the rows whose logits are read (the last rows of every sequence) need the
    // correction.  row_sel[m] != 0 <=> row m's operands carry a residual image: producers write the image of selected rows only (the buffers are
    // zero-filled once per pass, so an unselected row's image is all zero codes: its correction term is exactly 0 and its bits are the fast
    // schedule's wherever it sits in a tile), and a tile none of whose rows is selected skips the fp4 k-tiles altogether.  unit_sel[u] = OR of
    // row_sel over rows [64 u, 64 u + 64) (what a tile tests: a handful of scalar loads).  Both null = every row selected (round 5's schedule).
    const uint8_t* row_sel;
    const uint8_t* unit_sel;
    // Tile order under a row selection.  Workgroups are dispatched in id order, id b to XCD b % 8, and the dispatcher stalls on the first id
    // whose XCD has no free CU: the eight XCDs advance through their slabs in lock-step, so a few 1.25 x longer (corrected) tiles inside ONE XCD's
    // slab slow the whole launch to their pace (measured: gate/up 1.38 -> 1.73 ms with 2 of 29 row tiles corrected, profiles/r06_lo4_rows_launch.txt).
    // The row tiles that hold selected rows (sel_n ranges of row-tile indices for THIS launch's BM, filled by the launcher from the caller's host-side
    // row ranges) are therefore taken FIRST, round-robin over the XCDs in 32-tile patches, ahead of the other row tiles in the usual order.
    int sel_n, sel_total;                 // ranges (0 = plain order), row tiles in them
    int sel_tm_first[8], sel_tm_count[8];
};

// does any row of the tile [m0, m0 + bm) carry a residual image?  (m0 % 64 == 0 for every geometry; workgroup-uniform scalar loads)
LMI_DEV bool gemm_tile_selected(const GemmArgs& p, int m0, int bm) {
    if (!p.unit_sel) return true;
    const int u1 = ((m0 + bm < p.M ? m0 + bm : p.M) + 63) >> 6;
    unsigned any = 0;
    for (int u = m0 >> 6; u < u1; ++u) any |= p.unit_sel[u];
    return any != 0;
}

constexpr int GEMM_BK = 64;
constexpr int GEMM_GROUP_M = 4;                 // end-to-end sweep 2..16 on the C3 prefill: 4-6 best (-0.5 % vs 8), 16 +4 %; round 6 (profiles/r06_group_m_ab.txt): 4 x 8
                                                // tiles per XCD patch is also the HBM-side fetch minimum (1.37 GB per GEMM launch; 3: 1.41, 5: 1.47, 8: 1.67) — 5 is 0.3 - 0.5 %
                                                // faster on the lo4 step and moves 7.5 % more bytes: not taken

template <int BM_, int BN_, int WAVES_M_, int WAVES_N_, int STAGES_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, STAGES = STAGES_;
    static constexpr int NT = 64 * WAVES_M * WAVES_N;
    static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    static constexpr int MI = WTM / 32, NI = WTN / 32;
    static constexpr int ROWS_PER_PASS = NT / 8;               // 8 lanes cover one 128-byte row
    static constexpr int A_PASSES = BM / ROWS_PER_PASS, W_PASSES = BN / ROWS_PER_PASS;
    static constexpr int G = A_PASSES + W_PASSES;              // LDS-DMA instructions per thread per k-tile
    static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int SMEM = STAGES * STAGE_BYTES;
    static constexpr int SMEM_TOTAL = SMEM + BM * 4;             // + the row scales of a folded RMSNorm (GemmRowScale)
    // LO4: 8 E8M0 block scales per A row and fp4 k-tile, one slot per ring stage — rounded up to a whole number of 4-byte pieces per THREAD: every
    // wave issues the same number of VMEM operations per k-tile (the counted vmcnt waits of the ring rely on it); the surplus pieces read rows past
    // the tile (or past M: 0 from the bounds-checked buffer) into the slot's padding
    static constexpr int SC_BYTES = (2 * BM + NT - 1) / NT * NT * 4;
    static constexpr int SMEM_LO4 = SMEM_TOTAL + STAGES * SC_BYTES;
    static constexpr int D = STAGES - 1;                       // prefetch distance in k-tiles (>= 1)
    static_assert(STAGES >= 2, "ring needs at least two slots");
    static_assert(BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile rows vs threads");
    static_assert(WTM % 32 == 0 && WTN % 64 == 0, "wave tile must hold gate/up block pairs");
    static_assert(G * D <= 63, "vmcnt range");
};

// ---- fast activations (v_exp_f32 / v_rcp_f32; abs error ~1e-7, far below the 16-bit output rounding) ----------
LMI_DEV float fast_erf(float x) {                              // Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
    const float ax = fabsf(x);
    const float t = fast_rcp(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * fast_exp2(-ax * ax * 1.4426950408889634f);
    return x < 0.f ? -r : r;
}
LMI_DEV float fast_silu(float g) { return g * fast_rcp(1.0f + fast_exp2(-g * 1.4426950408889634f)); }

LMI_DEV float act_apply(float x, int act) {
    // 0.5 x (1 + tanh(u)) = x / (1 + e^(-2u)), u = sqrt(2/pi) (x + 0.044715 x^3): the exponent is x (c1 + c2 x^2) with the log2(e) folded in —
    // 5 VALU + 2 transcendental operations per element instead of 10 + 2 (the SigLIP fc1 epilogue is ~13 % of its launch); x -> -inf gives -0.
    // The multiplies are individually rounded (mul_rn): written as plain `*`, hipcc contracted them differently in different unrolled copies of
    // the epilogue, and a row's last fp32 bit — one fp16 tie in ~3e5 elements — depended on its position in the tile (caught by the packed ==
    // separate bit-identity tests).
    if (act == ACT_GELU_TANH) {
        const float p = __builtin_fmaf(mul_rn(x, x), -0.10294324f, -2.3022082f);
        return mul_rn(x, fast_rcp(1.0f + fast_exp2(mul_rn(x, p))));
    }
    if (act == ACT_GELU_ERF) return 0.5f * x * (1.0f + fast_erf(x * 0.7071067811865476f));
    return x;
}

// byte offset of logical 16-byte chunk `lc` of row `r` inside a [rows][64] 16-bit tile (128-byte rows)
LMI_DEV int gemm_lds_off(int r, int lc) { return r * 128 + ((lc ^ ((r >> 1) & 7)) << 4); }

// MFMA operand fragment of one lane for one k-step, per operand element type TA.  16-bit: 8 elements (16 bytes) of a 16-deep
// k-step, 4 k-steps per 128-byte k-tile row, v_mfma_f32_32x32x16.  fp8: 32 elements (32 bytes = two swizzled 16-byte chunks) of a
// 64-deep k-step, 2 k-steps per k-tile (128 k-elements), v_mfma_scale_f32_32x32x64_f8f6f4 — the same LDS image and DMA staging,
// twice the K per tile at twice the MFMA rate.
template <typename TA> struct GemmFrag { typedef typename vec_of<TA>::x8 type; static constexpr int KS = 4; };
template <> struct GemmFrag<fp8_t> { typedef v8i type; static constexpr int KS = 2; };
template <typename TA>
LMI_DEV typename GemmFrag<TA>::type gemm_frag_load(const char* tile, int row, int ks, int fh) {
    if constexpr (sizeof(TA) == 1) {
        const int c0 = ks * 4 + fh * 2;
        const u32x4 lo = *(const u32x4*)(tile + gemm_lds_off(row, c0)), hi = *(const u32x4*)(tile + gemm_lds_off(row, c0 + 1));
        return v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    } else {
        return *(const typename GemmFrag<TA>::type*)(tile + gemm_lds_off(row, ks * 2 + fh));
    }
}
// W operand staged from the PACKED weight order (GemmArgs::w_packed, 16-bit operands).  The packed order keeps the 16-byte pieces of 16
// consecutive rows with the same k-piece adjacent in memory (what the decode kernel's lanes want), so a DMA lane quad that fetches FOUR
// CHUNKS OF ONE ROW — the row-major image's lane order — touches four cache lines instead of one (measured: C3 step 150.0 ms against 134.0).
// The W image therefore changes with the source: every 1-KiB piece (8 rows x 8 chunks) is stored chunk-major — slot (c ^ (piece & 1)) * 8 + j
// holds chunk c of the piece's row j — so that a lane quad fetches the same chunk of four consecutive rows (64 contiguous bytes, an octet
// one full 128-byte line, exactly the row-major traffic pattern).  Reads: a 16-lane ds_read_b128 group takes rows 0..7 of an even and
// of an odd piece at one chunk: 8 consecutive 16-byte slots each, in opposite 128-byte halves of the 256-byte bank row (the parity
// flip) — conflict-free.  Same fragments, same MFMA order, same bits as the row-major call.
LMI_DEV int gemm_lds_off_wp(int r, int lc) { return (r >> 3) * 1024 + ((((lc ^ ((r >> 3) & 1)) << 3) + (r & 7)) << 4); }
// Per-lane W fragment offsets, fixed before the main loop so that the layout choice is a select of three loop invariants and never
// control flow inside it: for row R0 + fr (R0 a multiple of 32) and chunk lc = 2 ks + fh both images are
//     R0 * 128 + lane_base + ((lc ^ X) << SH)      row-major: lane_base = fr * 128,                         X = (fr >> 1) & 7, SH = 4
//                                                  packed:    lane_base = (fr >> 3) * 1024 + (fr & 7) * 16, X = (fr >> 3) & 1, SH = 7
// (gemm_lds_off / gemm_lds_off_wp with the R0 terms folded: R0 / 2 is a multiple of 8, R0 / 8 is even).
template <typename TA> struct GemmWOff {
    static constexpr int KS = GemmFrag<TA>::KS;
    int ko[KS];
    LMI_DEV void init(int w_packed, int fr, int fh) {
        const bool wp = sizeof(TA) == 2 && w_packed;
        const int lane_base = wp ? (fr >> 3) * 1024 + (fr & 7) * 16 : fr * 128;
        const int x = wp ? (fr >> 3) & 1 : (fr >> 1) & 7;
        const int sh = wp ? 7 : 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ko[ks] = lane_base + (((ks * 2 + fh) ^ x) << sh);
    }
};
template <typename TA>
LMI_DEV typename GemmFrag<TA>::type gemm_wfrag_load(const char* tile, int r0, int fr, int ks, int fh, const GemmWOff<TA>& wo) {
    if constexpr (sizeof(TA) == 1) return gemm_frag_load<TA>(tile, r0 + fr, ks, fh);
    else return *(const typename GemmFrag<TA>::type*)(tile + r0 * 128 + wo.ko[ks]);
}
LMI_DEV f32x16 gemm_mma(f16x8 a, f16x8 b, f32x16 c, int) { return mfma32(a, b, c); }
LMI_DEV f32x16 gemm_mma(bf16x8 a, bf16x8 b, f32x16 c, int) { return mfma32(a, b, c); }
LMI_DEV f32x16 gemm_mma(v8i a, v8i b, f32x16 c, int scale_e8m0) { return mfma32_fp8(a, b, c, scale_e8m0); }
// the 16 bytes of a 16-bit fragment as the fp4 operand of the correction phase (the v8i overload only keeps fp8 instantiations well-formed)
LMI_DEV u32x4 frag_bits(f16x8 f) { return __builtin_bit_cast(u32x4, f); }
LMI_DEV u32x4 frag_bits(bf16x8 f) { return __builtin_bit_cast(u32x4, f); }
LMI_DEV u32x4 frag_bits(v8i f) { return u32x4{(uint32_t)f[0], (uint32_t)f[1], (uint32_t)f[2], (uint32_t)f[3]}; }
template <int N, typename F> LMI_DEV void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// ---- low-bit correction phase: block scales of the A residual image -----------------------------------------------------------------------
// Per fp4 k-tile (256 k = 8 blocks) every A row has 8 scale bytes.  They travel like the operands: one 4-byte LDS-DMA piece per thread
// (thread t -> dword t of the tile's [BM][2] dword image; rows past M read zeros through the buffer's range check = scale 2^-127) into a
// per-stage slot behind the row scales of the folded norm.  A lane (fr, fh) reads its row's two dwords once per k-tile; block 2 ks + fh
// of the tile is byte (ks & 1) * 2 + fh of dword ks >> 1, so the odd half-wave shifts its dwords right by 8 and both halves select bytes
// 0 / 2 with the instruction's op_sel.  The weight image has ONE scale per row: two bytes per lane, loaded before the main loop.
template <typename C>
struct GemmLo4 {
    static constexpr int SP = (2 * C::BM + C::NT - 1) / C::NT;    // 4-byte pieces per thread per k-tile
    BufRsrc s_buf;
    unsigned s_src[SP];
    char* sc_base;            // slot 0 of the scale ring
    int wave_off;
    int w_sc[C::NI];
    LMI_DEV void init(const GemmArgs& p, int m0, int n0, int tid, int wave, int wn, int fr, char* smem) {
        s_buf = make_buf(p.a4_scale, p.a4s_bytes);
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            const int idx = j * C::NT + tid;                      // dword of the [BM][2] image: row idx >> 1, half idx & 1
            s_src[j] = (unsigned)((long)(m0 + (idx >> 1)) * p.lds4 + (idx & 1) * 4);
        }
        sc_base = smem + C::SMEM_TOTAL;
        wave_off = wave * 256;
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni) w_sc[ni] = (int)p.w4_scale[imin(n0 + wn * C::WTN + ni * 32 + fr, p.N - 1)];
    }
    LMI_DEV void issue(int kt, int slot) const {
        // every wave issues all SP pieces, also those past the tile's 2 BM dwords (64 x 128: waves 2 - 3; 384 x 128: waves 4 - 7 of the second
        // piece): a wave that skipped them had fewer VMEM operations in flight per fp4 k-tile than the counted wait of k_tile() assumes, which let
        // it read a ring slot before its own last operand pieces had landed (round 5: found by the packed == separate bit-identity at M = 228)
#pragma unroll
        for (int j = 0; j < SP; ++j) glds4_buf(s_buf, s_src[j], (unsigned)kt * 8u, sc_base + slot * C::SC_BYTES + j * C::NT * 4 + wave_off);
    }
    // scale dwords of row `row` of the tile for this lane's k-half: lo = blocks 0..3, hi = blocks 4..7 (bytes 0 / 2 after the shift)
    LMI_DEV void read(int slot, int row, int fh, int& lo, int& hi) const {
        const u32x2 d = *(const u32x2*)(sc_base + slot * C::SC_BYTES + row * 8);
        lo = (int)(d[0] >> (fh * 8));
        hi = (int)(d[1] >> (fh * 8));
    }
};

// XCD-aware, grouped tile order.  Tiles are linearised in groups of `group_m` row-tiles, row-tile fastest, so that 32
// consecutive ids form a (group_m x 32/group_m)-tile patch whose current k-tiles one XCD's 4 MiB L2 can hold.  Workgroup b runs on XCD
// b % 8 (observed dispatch; speed only).  order 0: XCD x owns the x-th contiguous eighth of the linear order (bijective
// for any tile count).  order 1: the XCDs take 32-tile patches round-robin, so at any time all eight work on
// neighbouring patches of the SAME row-tile group and share its A panel through the Infinity Cache; the grid is rounded
// up to a multiple of 256 workgroups and surplus workgroups exit.  Returns false for a surplus workgroup.
LMI_DEV bool gemm_tile_coords(int bid, int tiles_m, int tiles_n, int group_m, int order, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = bid & 7, idx = bid >> 3;
    int swz;
    if (order == 1) {
        swz = ((idx >> 5) * 8 + xcd) * 32 + (idx & 31);
        if (swz >= nwg) return false;
    } else {
        const int q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = group_m * tiles_n;
    const int g = swz / per_group;
    const int first_m = g * group_m;
    const int gsize = imin(group_m, tiles_m - first_m);
    const int in_g = swz - g * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
    return true;
}

// tile order with the selected row tiles first (GemmArgs::sel_*); grid = gemm_grid_size(...) workgroups
LMI_DEV bool gemm_tile_coords_sel(const GemmArgs& p, int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    if (p.sel_n == 0) return gemm_tile_coords(bid, tiles_m, tiles_n, p.group_m, p.order, tm, tn);
    // prefix: XCD x (= bid % 8) takes the columns x, x + 8, ... of the selected row tiles, the selected row tiles of one column back to back (they
    // share the column's W / W4 k-tiles in that XCD's L2); evenly spread whatever the count (round-robin 32-tile patches put 112 tiles on 4 XCDs)
    const int pad = ((tiles_n + 7) >> 3) * 8 * p.sel_total;
    if (bid < pad) {
        const int xcd = bid & 7, q = bid >> 3;
        tn = (q / p.sel_total) * 8 + xcd;
        if (tn >= tiles_n) return false;
        int j = q % p.sel_total;
        tm = 0;
        for (int r = 0; r < p.sel_n; ++r) {
            if (j < p.sel_tm_count[r]) { tm = p.sel_tm_first[r] + j; break; }
            j -= p.sel_tm_count[r];
        }
        return true;
    }
    int tmu;
    gemm_tile_coords(bid - pad, tiles_m - p.sel_total, tiles_n, p.group_m, 0, tmu, tn);   // the other row tiles, XCD slabs of the grouped order
    for (int r = 0; r < p.sel_n; ++r)
        if (tmu >= p.sel_tm_first[r]) tmu += p.sel_tm_count[r];                            // ranges ascending: step over every selected range at or below
    tm = tmu;
    return true;
}
inline int g_sel_keep_ragged_last = 0;
inline int gemm_grid_size(const GemmArgs& p, int bm, int bn) {
    const int tiles_m = (p.M + bm - 1) / bm, tiles_n = (p.N + bn - 1) / bn;
    if (p.sel_n > 0) return ((tiles_n + 7) >> 3) * 8 * p.sel_total + (tiles_m - p.sel_total) * tiles_n;
    const int tiles = tiles_m * tiles_n;
    return p.order == 1 ? (tiles + 255) / 256 * 256 : tiles;
}
// row ranges [begin, end) (host memory, ascending, disjoint) -> ranges of row tiles of height bm; more than 8 after merging: plain order
inline void gemm_fill_sel(GemmArgs& a, const int* ranges, int n, int bm) {
    a.sel_n = a.sel_total = 0;
    if (!ranges || n <= 0 || !a.unit_sel) return;
    const int tiles_m = (a.M + bm - 1) / bm;
    const int full_tiles = g_sel_keep_ragged_last ? a.M / bm : tiles_m;     // A/B knob (lmi_set_option "gemm.sel_ragged_last"): a ragged last row tile stays last
    int cnt = 0, first[8], count[8], prev_end = -1;
    for (int i = 0; i < n; ++i) {
        int b = ranges[2 * i], e = ranges[2 * i + 1];
        if (b < 0) b = 0;
        if (e > a.M) e = a.M;
        if (e <= b) continue;
        int t0 = b / bm, t1 = (e - 1) / bm + 1;
        if (t1 > full_tiles) t1 = full_tiles;
        if (t0 < prev_end) t0 = prev_end;                               // shares a tile with the previous range
        if (t1 <= t0) continue;
        if (cnt > 0 && t0 == prev_end) count[cnt - 1] += t1 - t0;
        else {
            if (cnt == 8) return;                                       // too many pieces: keep the plain order
            first[cnt] = t0; count[cnt] = t1 - t0; ++cnt;
        }
        prev_end = t1;
    }
    int total = 0;
    for (int i = 0; i < cnt; ++i) total += count[i];
    if (total == 0 || total >= tiles_m) return;                         // nothing / everything selected: the plain order
    a.sel_n = cnt; a.sel_total = total;
    for (int i = 0; i < cnt; ++i) { a.sel_tm_first[i] = first[i]; a.sel_tm_count[i] = count[i]; }
}

// ---- epilogue shared by every geometry -------------------------------------------------------------------------
// The accumulator layout gives a lane ONE output row and 4-column quads, so storing straight from it writes 8/16-byte
// pieces into 32 different rows per instruction (measured: 1.8-2.8 TB/s on the output stream, a quarter of the SigLIP
// GEMMs' time).  Instead every wave turns its tile through its own slice of the (now idle) LDS, 32 rows at a time:
//   write: lane (fr, fh) puts its 4*NI quads of row fr into a [32][WTN] fp32 image (row stride WTN*4 + 16 bytes: the 8
//          lanes of a ds_write_b128 group land in 8 distinct 16-byte bank groups);
//   read:  lane l takes 8 consecutive columns of row l / LPR, so that a wave instruction covers whole rows of the
//          wave tile: 16-byte T stores / 32-byte fp32 loads and stores per lane, 256-512 contiguous bytes per row.
// bias / position table / activation / residual add are applied on the read side, in the same order per element as the
// definition (acc + bias, + table row, act, + residual), so results are bit-identical to the direct form.
// DS operations of one wave execute in order, so the write -> read -> next write sequence needs no barrier; the caller
// guarantees that no other wave still reads k-tiles from `stage`.
// `put(mi, stage)` writes rows mi*32 .. mi*32+31 of the wave tile into the image (it depends on the MFMA shape's accumulator layout).
template <int WTN> struct GemmImage { static constexpr int RS = WTN * 4 + 16; };   // LDS row stride of the fp32 image

// Row scales of the folded RMSNorm for the BM rows of a tile: rstd[r] = rsqrt(sum_j rowsq_in[m0 + r, j] / norm_dim + eps), kept in a
// BM-float LDS array behind the k-tile ring; the epilogue reads one float per row.  Two phases so that the memory latency costs
// nothing (a tile's prologue and epilogue are exposed: one workgroup per CU): `load` issues this thread's share of the row's
// partials BEFORE the first LDS-DMA pieces (older VMEM operations complete first, so the counted vmcnt waits of the main loop are
// unaffected) and holds them in registers through the main loop; `finish`, after the last k-tile, sums them, exchanges with the
// 1 or 3 other threads of the row and writes the scale to LDS (the caller's next barrier publishes it).
// The summation order is fixed — four quarter sums of contiguous partials, ((q0 + q1) + (q2 + q3)) — whatever the tile geometry
// (NT / BM = 2 or 4 threads share a row), so a row's scale does not depend on the geometry chosen for the problem size or on
// where the row sits in a packed batch.
template <typename C>
struct GemmRowScale {
    // geometries whose thread count is 2x or 4x the tile rows share a row between 2 / 4 threads and prefetch; any other geometry (384 x 128:
    // 512 threads for 384 rows) gives thread r < BM the whole row and reads the partials after the main loop (LATE: one exposed L2 round
    // trip per tile, on shapes whose tiles run for tens of microseconds)
    static constexpr bool LATE = !(C::NT % C::BM == 0 && (C::NT / C::BM == 2 || C::NT / C::BM == 4));
    static constexpr int TPR = LATE ? 1 : C::NT / C::BM;                     // threads per row
    static constexpr int QPT = 4 / TPR;                                      // quarters per thread
    f32x4 v[LATE ? 1 : QPT][4];                                              // fast path: 16 partials per quarter (hidden size 4096)
    const float* part;
    int qlen;

    LMI_DEV void load(const GemmArgs& p, int m0, int tid) {
        const int r = tid / TPR, sub = tid % TPR;
        part = p.rowsq_in + (long)imin(m0 + imin(r, C::BM - 1), p.M - 1) * p.rowsq_parts + sub * QPT * (p.rowsq_parts >> 2);
        qlen = p.rowsq_parts >> 2;                                           // rowsq_parts % 4 == 0 is checked by the launcher
        if (!LATE && qlen == 16) {
#pragma unroll
            for (int i = 0; i < QPT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = *(const f32x4*)(part + i * 16 + j * 4);
        }
    }
    LMI_DEV void finish(const GemmArgs& p, int tid, float* rstd_lds) {
        float q[QPT];
#pragma unroll
        for (int i = 0; i < QPT; ++i) {
            float acc = 0.f;
            if (!LATE && qlen == 16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
            } else if ((qlen & 3) == 0) {
                for (int j = 0; j < qlen; j += 4) {
                    const f32x4 w = *(const f32x4*)(part + i * qlen + j);
                    acc += (w[0] + w[1]) + (w[2] + w[3]);
                }
            } else {
                for (int j = 0; j < qlen; ++j) acc += part[i * qlen + j];
            }
            q[i] = acc;
        }
        float s;
        if (TPR == 1) {
            s = (q[0] + q[1 % QPT]) + (q[2 % QPT] + q[3 % QPT]);             // the same order as the shared-row forms below
        } else if (TPR == 2) {
            s = q[0] + q[QPT - 1];                                           // (q0 + q1) resp. (q2 + q3)
            s += shfl_xor(s, 1);
        } else {
            s = q[0];
            s += shfl_xor(s, 1);                                             // (q0 + q1), (q2 + q3)
            s += shfl_xor(s, 2);
        }
        if (tid % TPR == 0 && tid / TPR < C::BM) rstd_lds[tid / TPR] = 1.0f / sqrtf(s / (float)p.norm_dim + p.norm_eps);
        lds_write_drain();
    }
};
// the epilogues that may finish a folded RMSNorm (the others never carry the row-scale registers)
template <int EPI> struct GemmCanScale { static constexpr bool value = (EPI == EPI_STORE_T || EPI == EPI_SWIGLU_T || EPI == EPI_QKV_ROPE_T); };

// `tile_sel` (workgroup-uniform; LO4 producers): does any row of this tile carry a residual image (gemm_tile_selected)?  A tile without one
// must not even LOAD row_sel: 16 dependent byte loads per wave and tile in the exposed epilogue cost gate/up + 3.9 % and o / down + 2.5 % on
// the C3 step, where 28 of 29 row tiles hold no selected row (round 6, profiles/r06_lo4_epilogue_rowsel_ab.txt).
template <typename T, int EPI, int ACT, typename C, bool OUT4 = false, typename Put>
LMI_DEV void gemm_epilogue(const GemmArgs& p, Put put, int m0, int n0, int wm, int wn, int lane, char* stage, const float* rstd_lds, bool tile_sel = true) {
    typedef typename vec_of<T>::x8 T8;
    constexpr int RS = GemmImage<C::WTN>::RS;
    static_assert(32 * RS <= C::SMEM / (C::NT / 64), "per-wave LDS slice too small for the epilogue image");
    const int nw0 = n0 + wn * C::WTN;                                   // first column of this wave
    if (nw0 >= p.N) return;                                             // N % 128 == 0 and WTN | 128: all or nothing
    constexpr bool PAIRED = (EPI == EPI_SWIGLU_T || EPI == EPI_QKV_ROPE_T);   // a lane reads two 8-column pieces 32 columns apart
    constexpr int OUT_COLS = PAIRED ? C::WTN / 2 : C::WTN;
    constexpr int LPR = OUT_COLS / 8, RPI = 64 / LPR, ITERS = 32 / RPI; // lanes per row, rows per instruction
    const int r_in = lane / LPR;
    // 8 lanes per 64-column row (every production geometry, plain epilogues): rotating the lanes' column groups by one on
    // rows 2, 3, 6, 7 of an instruction makes each of ds_read_b128's four lane groups hit 16 distinct 16-byte slots (with
    // lane -> column fixed, rows two apart share slots: 2-way conflicts, 4.5 % of the ring kernels' LDS cycles by PMC).
    // A row is still covered by 8 consecutive lanes, so stores stay whole 128/256-byte row segments.
    // (not in producer mode: there the 8 lanes of a row sum their squares with a fixed shuffle tree, and the pairing of column
    // groups in that tree must not depend on the row's position — bit-identical partials wherever a row sits in a packed batch)
    // (nor when the residual image of the output is wanted: a 32-column block must sit in one lane quad, below)
    const int oc = ((LPR == 8 && !PAIRED && !(EPI == EPI_RESID_F32 && p.norm_out) && !(OUT4 && p.out4)) ? ((lane + 7 * ((r_in >> 1) & 1)) & 7) : (lane % LPR)) * 8;
    // source columns of this lane in the image: plain = oc..oc+7; SwiGLU = gate block, up block 32 columns further;
    // RoPE = first-half block, rotate-half partner block 32 columns further
    const int sc = PAIRED ? (oc >> 5) * 64 + (oc & 31) : oc;
    f32x4 gam0 = {0.f, 0.f, 0.f, 0.f}, gam1 = gam0;          // producer mode: the next RMSNorm's gain for this lane's 8 columns
    if (EPI == EPI_RESID_F32 && p.norm_out) {
        gam0 = *(const f32x4*)(p.norm_gamma + nw0 + oc);
        gam1 = *(const f32x4*)(p.norm_gamma + nw0 + oc + 4);
    }
    f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
    if (!PAIRED && p.bias) {
        bias0 = *(const f32x4*)(p.bias + nw0 + oc);
        bias1 = *(const f32x4*)(p.bias + nw0 + oc + 4);
    }
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        const int mb = m0 + wm * C::WTM + mi * 32;
        if (mb >= p.M) break;
        put(mi, stage);
        wave_lds_fence();
        if (EPI == EPI_QKV_ROPE_T) {
            // wave-uniform: which of q | k | v this wave's 64 columns belong to, and which half-block of its head
            const bool is_v = nw0 >= p.rope_q + p.rope_k, is_k = !is_v && nw0 >= p.rope_q;
            const int hb = (nw0 >> 6) & 1;                              // 64-column block inside the 128-wide head
            const int hbase = nw0 & ~127;                               // first column of the head
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int row = it * RPI + r_in, m = mb + row;
                const int mc = imin(m, p.M - 1);
                const char* src = stage + row * RS + sc * 4;
                f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 16);
                f32x4 b0 = *(const f32x4*)(src + 128), b1 = *(const f32x4*)(src + 144);
                if (GemmCanScale<EPI>::value && p.rowsq_in) {
                    const float rstd = rstd_lds[wm * C::WTM + mi * 32 + row];
                    a0 *= rstd; a1 *= rstd; b0 *= rstd; b1 *= rstd;
                }
                T8 o1, o2;
                int c1, c2;                                             // destination columns of the two 8-wide pieces
                if (is_v) {                                             // v: natural order, plain copy
                    c1 = nw0 + oc; c2 = c1 + 32;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o1[e] = OutCvt<T>::cvt(a0[e]); o1[4 + e] = OutCvt<T>::cvt(a1[e]); o2[e] = OutCvt<T>::cvt(b0[e]); o2[4 + e] = OutCvt<T>::cvt(b1[e]); }
                } else {                                                // q / k: rotate-half RoPE on the fp32 accumulators
                    const int d1 = hb * 32 + oc;                        // 0..63: element of the first half; partner d1 + 64
                    c1 = hbase + d1; c2 = c1 + 64;
                    const float* cs = p.rope_cos + (long)mc * 64 + d1;
                    const float* sn = p.rope_sin + (long)mc * 64 + d1;
                    const f32x4 cs0 = *(const f32x4*)cs, cs1 = *(const f32x4*)(cs + 4);
                    const f32x4 sn0 = *(const f32x4*)sn, sn1 = *(const f32x4*)(sn + 4);
                    // explicit fma shape (one individually rounded product, one fused multiply-add): left to the compiler, each
                    // kernel instantiation contracts `a*c - b*s` its own way and a row's bits would depend on the tile geometry
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o1[e] = OutCvt<T>::cvt(sep_rn(__builtin_fmaf(a0[e], cs0[e], -mul_rn(b0[e], sn0[e]))));
                        o1[4 + e] = OutCvt<T>::cvt(sep_rn(__builtin_fmaf(a1[e], cs1[e], -mul_rn(b1[e], sn1[e]))));
                        o2[e] = OutCvt<T>::cvt(sep_rn(__builtin_fmaf(b0[e], cs0[e], mul_rn(a0[e], sn0[e]))));
                        o2[4 + e] = OutCvt<T>::cvt(sep_rn(__builtin_fmaf(b1[e], cs1[e], mul_rn(a1[e], sn1[e]))));
                    }
                }
                if (m < p.M) {
                    T* orow_p = (T*)p.out + (long)m * p.ldo;
                    *(T8*)(orow_p + c1) = o1;
                    *(T8*)(orow_p + c2) = o2;
                    if ((is_k || is_v) && p.k_cache) {
                        T* crow = (T*)(is_k ? p.k_cache : p.v_cache) + (long)(p.cache_pos0 + m) * p.ld_cache - (is_k ? p.rope_q : p.rope_q + p.rope_k);
                        *(T8*)(crow + c1) = o1;
                        *(T8*)(crow + c2) = o2;
                    }
                }
            }
            wave_lds_fence();
            continue;
        }
        if (EPI == EPI_SWIGLU_T) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int row = it * RPI + r_in, m = mb + row;
                const char* src = stage + row * RS + sc * 4;
                f32x4 g0 = *(const f32x4*)src, g1 = *(const f32x4*)(src + 16);
                f32x4 u0 = *(const f32x4*)(src + 128), u1 = *(const f32x4*)(src + 144);
                if (GemmCanScale<EPI>::value && p.rowsq_in) {
                    const float rstd = rstd_lds[wm * C::WTM + mi * 32 + row];
                    g0 *= rstd; g1 *= rstd; u0 *= rstd; u1 *= rstd;
                }
                T8 o;
                const float os = (sizeof(T) == 1) ? p.out_scale : 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = OutCvt<T>::cvt(sep_rn(fast_silu(g0[e]) * u0[e] * os));
                    o[4 + e] = OutCvt<T>::cvt(sep_rn(fast_silu(g1[e]) * u1[e] * os));
                }
                if constexpr (OUT4 && sizeof(T) == 2) {
                    const bool sel = tile_sel && (!p.row_sel || p.row_sel[m < p.M ? m : p.M - 1]);
                    if (p.out4 && wave_any(sel)) {                   // (wave-uniform) residual image of the products: a row's 4 lanes = one 32-column block
                        float y[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { y[e] = fast_silu(g0[e]) * u0[e]; y[4 + e] = fast_silu(g1[e]) * u1[e]; }
                        unsigned sb;
                        const unsigned codes = lo4_encode8<T>(y, o, sb);
                        if (m < p.M && sel) {
                            const long orow4 = p.row_map ? (long)p.row_map[m] : (long)m;
                            const int col = (nw0 >> 1) + oc;
                            *(unsigned*)((char*)p.out4 + orow4 * p.ld_out4 + (col >> 1)) = codes;
                            if ((lane & 3) == 0) p.out4_scale[orow4 * p.ld_out4s + (col >> 5)] = (uint8_t)sb;
                        }
                    }
                }
                if (m < p.M) {
                    const long orow = p.row_map ? (long)p.row_map[m] : (long)m;
                    if (p.swiglu_f32) {                              // wave-uniform: the unrounded products
                        float* d32 = (float*)p.out + orow * p.ldo + (nw0 >> 1) + oc;
                        *(f32x4*)d32 = f32x4{fast_silu(g0[0]) * u0[0], fast_silu(g0[1]) * u0[1], fast_silu(g0[2]) * u0[2], fast_silu(g0[3]) * u0[3]};
                        *(f32x4*)(d32 + 4) = f32x4{fast_silu(g1[0]) * u1[0], fast_silu(g1[1]) * u1[1], fast_silu(g1[2]) * u1[2], fast_silu(g1[3]) * u1[3]};
                    } else {
                        st_epi<2>((T8*)((T*)p.out + orow * p.ldo + (nw0 >> 1) + oc), o);
                    }
                }
            }
            wave_lds_fence();
            continue;
        }
        f32x4 v0[ITERS], v1[ITERS], old0[ITERS], old1[ITERS];
        long orow[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {                              // all loads of the 32 rows first ...
            const int row = it * RPI + r_in;
            const int mc = imin(mb + row, p.M - 1);                       // clamped: loads are unconditional, stores masked
            orow[it] = p.row_map ? (long)p.row_map[mc] : (long)mc;
            const char* src = stage + row * RS + sc * 4;
            v0[it] = *(const f32x4*)src;
            v1[it] = *(const f32x4*)(src + 16);
            if (GemmCanScale<EPI>::value && p.rowsq_in) {
                const float rstd = rstd_lds[wm * C::WTM + mi * 32 + row];
                v0[it] *= rstd; v1[it] *= rstd;
            }
            v0[it] += bias0;
            v1[it] += bias1;
            if (p.addmat) {                                               // SigLIP position table (patch-embed GEMM only)
                const float* arow = p.addmat + (long)(p.add_rows ? p.add_rows[mc] : mc % p.add_period) * p.N + nw0 + oc;
                v0[it] += *(const f32x4*)arow;
                v1[it] += *(const f32x4*)(arow + 4);
            }
            if (EPI == EPI_RESID_F32) {
                const float* drow = (const float*)p.out + orow[it] * p.ldo + nw0 + oc;
                old0[it] = ld_epi<1>((const f32x4*)drow);
                old1[it] = ld_epi<1>((const f32x4*)(drow + 4));
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {                              // ... then the arithmetic and the stores
            if (ACT != ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[it][e] = act_apply(v0[it][e], ACT); v1[it][e] = act_apply(v1[it][e], ACT); }
            }
            if (EPI == EPI_RESID_F32) { v0[it] += old0[it]; v1[it] += old1[it]; }
            float sq = 0.f;
            if (EPI == EPI_RESID_F32 && p.norm_out) {
                // sum of squares of this wave's 64 columns of the updated row: the 8 lanes of a row are consecutive; every lane
                // takes part (rows past M carry clamped duplicates and are not stored)
#pragma unroll
                for (int e = 0; e < 4; ++e) sq = __builtin_fmaf(v0[it][e], v0[it][e], sq);       // fixed fma chain: same bits in every instantiation
#pragma unroll
                for (int e = 0; e < 4; ++e) sq = __builtin_fmaf(v1[it][e], v1[it][e], sq);
                sq += shfl_xor(sq, 1); sq += shfl_xor(sq, 2); sq += shfl_xor(sq, 4);
            }
            // residual image of the 16-bit values handed to the next GEMM (every lane takes part in the quad exchange, rows past M included)
            unsigned codes4 = 0, sb4 = 0;
            T8 o4;
            bool sel4 = false;
            if constexpr (OUT4 && sizeof(T) == 2 && (EPI == EPI_STORE_T || EPI == EPI_RESID_F32)) {
                if (tile_sel && p.out4 && (EPI == EPI_STORE_T || p.norm_out)) sel4 = !p.row_sel || p.row_sel[imin(mb + it * RPI + r_in, p.M - 1)];
                if (wave_any(sel4)) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[e] = EPI == EPI_STORE_T ? v0[it][e] : v0[it][e] * gam0[e];
                        y[4 + e] = EPI == EPI_STORE_T ? v1[it][e] : v1[it][e] * gam1[e];
                    }
                    codes4 = lo4_encode8<T>(y, o4, sb4);
                }
            }
            if (mb + it * RPI + r_in >= p.M) continue;
            if constexpr (OUT4 && sizeof(T) == 2 && (EPI == EPI_STORE_T || EPI == EPI_RESID_F32)) {
                if (sel4) {
                    *(unsigned*)((char*)p.out4 + orow[it] * p.ld_out4 + ((nw0 + oc) >> 1)) = codes4;
                    if ((lane & 3) == 0) p.out4_scale[orow[it] * p.ld_out4s + ((nw0 + oc) >> 5)] = (uint8_t)sb4;
                }
            }
            if (EPI == EPI_STORE_T) {
                T8 o;
                const float os = (sizeof(T) == 1) ? p.out_scale : 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = OutCvt<T>::cvt(sep_rn(v0[it][e] * os)); o[4 + e] = OutCvt<T>::cvt(sep_rn(v1[it][e] * os)); }
                st_epi<2>((T8*)((T*)p.out + orow[it] * p.ldo + nw0 + oc), o);
            } else {
                float* drow = (float*)p.out + orow[it] * p.ldo + nw0 + oc;
                st_epi<1>((f32x4*)drow, v0[it]);
                st_epi<1>((f32x4*)(drow + 4), v1[it]);
                if (EPI == EPI_RESID_F32 && p.norm_out) {
                    // the next RMSNorm, started here: gain applied and rounded; its row scale is finished by the consumer GEMM
                    T8 hn;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hn[e] = OutCvt<T>::cvt(sep_rn(v0[it][e] * gam0[e])); hn[4 + e] = OutCvt<T>::cvt(sep_rn(v1[it][e] * gam1[e])); }
                    *(T8*)((T*)p.norm_out + orow[it] * p.ld_norm + nw0 + oc) = hn;
                    if ((lane & 7) == 0) p.rowsq_out[orow[it] * (long)(p.N >> 6) + (nw0 >> 6)] = sq;
                }
            }
        }
    }
}

// ---- operand staging shared by both schedules -----------------------------------------------------------------
// Per-thread LDS-DMA sources: one 128-byte row per pass (8 lanes per row), the 16-byte chunk position fixed per thread
// and swizzled on the source side; loop-invariant 32-bit lane offsets, the k-tile advances a scalar offset.  Tail rows
// are clamped to the last valid row (their products are masked on store).  In pixel-shuffle mode logical A row m is the
// 2x2 neighbourhood of ViT tokens of shuffled token m, its K axis the four (dh, dw) segments of C channels.
// W in the PACKED order (p.w_packed; 16-bit operands): the layout lmi_gemm_skinny streams — 16-row group rg, 128-k step s, 32-k chunk c
// form one 1-KiB block at ((rg * K/128 + s) * 4 + c) * 1024 with the 16-byte piece (row i, 8-k piece g) at (16 g + i) * 16 — is a
// permutation of the 16-byte pieces of the row-major matrix that keeps every row group and every 64-k tile together: tile kt of a
// row group is the two consecutive blocks at kt * 2048.  An LDS-DMA lane picks its source piece freely, so the same fragments (the same
// MFMA order, the same bits) are staged from either layout: the loop-invariant lane offset, the scalar step per k-tile and the order
// of the pieces inside the W image (gemm_lds_off_wp) differ.  One copy of the LLM weights then serves the prefill GEMM and the decode kernels.
template <int AMODE, typename C, int ES = 2>          // ES = bytes per operand element (2: f16 / bf16, 1: fp8)
struct GemmStager {
    BufRsrc a_buf, w_buf;
    unsigned a_src[C::A_PASSES], w_src[C::W_PASSES];
    int ps_c, ps_grid, lda;
    unsigned w_kstep;                                    // bytes a k-tile advances every W lane offset by
    char* wave_base;

    LMI_DEV void init(const GemmArgs& p, int m0, int n0, int tid, char* smem, int wave) {
        a_buf = make_buf(p.A, p.a_bytes);
        w_buf = make_buf(p.W, p.w_bytes);
        const int srow = tid >> 3;                       // physical row inside a pass
        const int pc = tid & 7;                          // physical 16-byte chunk inside the 128-byte row
#pragma unroll
        for (int ps = 0; ps < C::A_PASSES; ++ps) {
            const int r = ps * C::ROWS_PER_PASS + srow;
            const int lc = pc ^ ((r >> 1) & 7);
            const int am = imin(m0 + r, p.M - 1);
            long arow;
            if (AMODE == AMODE_PIXSHUF) {
                const int g = p.ps_grid, h = g >> 1, per = h * h;
                const int tile = am / per, pp = am - tile * per;
                const int ph = pp / h, pw = pp - ph * h;
                arow = (long)tile * g * g + (long)(2 * ph) * g + 2 * pw;
            } else {
                arow = am;
            }
            a_src[ps] = (unsigned)(arow * p.lda * ES + lc * 16);
        }
#pragma unroll
        for (int ps = 0; ps < C::W_PASSES; ++ps) {
            const int r = ps * C::ROWS_PER_PASS + srow;
            const int lc = pc ^ ((r >> 1) & 7);
            if (ES == 2 && p.w_packed) {
                // chunk-major piece (gemm_lds_off_wp): lane l of the wave's piece = row l & 7, slot l >> 3
                const int l = tid & 63, rp = ps * C::ROWS_PER_PASS + wave * 8 + (l & 7);
                const int c = (l >> 3) ^ ((rp >> 3) & 1);
                const int n = imin(n0 + rp, p.N - 1);
                w_src[ps] = (unsigned)((long)(n >> 4) * 32 * p.K + (c >> 2) * 1024 + ((c & 3) * 16 + (n & 15)) * 16);
            } else {
                w_src[ps] = (unsigned)((long)imin(n0 + r, p.N - 1) * p.ldw * ES + lc * 16);
            }
        }
        w_kstep = (ES == 2 && p.w_packed) ? 2048u : 128u;
        ps_c = (AMODE == AMODE_PIXSHUF) ? (p.K >> 2) : 1;    // channels per shuffle segment
        ps_grid = p.ps_grid;
        lda = p.lda;
        wave_base = smem + wave * 1024;
    }
    // Low-bit correction phase: from here on the pieces come from the fp4 images (row-major, 128 bytes = 256 k per k-tile and row; plain
    // A mode).  Called between the last 16-bit piece and the first fp4 piece; pieces already in flight do not depend on these registers.
    LMI_DEV void switch_lo4(const GemmArgs& p, int m0, int n0, int tid) {
        a_buf = make_buf(p.A4, p.a4_bytes);
        w_buf = make_buf(p.W4, p.w4_bytes);
        const int srow = tid >> 3, pc = tid & 7;
#pragma unroll
        for (int ps = 0; ps < C::A_PASSES; ++ps) {
            const int r = ps * C::ROWS_PER_PASS + srow;
            a_src[ps] = (unsigned)((long)imin(m0 + r, p.M - 1) * p.lda4 + ((pc ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int ps = 0; ps < C::W_PASSES; ++ps) {
            const int r = ps * C::ROWS_PER_PASS + srow;
            w_src[ps] = (unsigned)((long)imin(n0 + r, p.N - 1) * p.ldw4 + ((pc ^ ((r >> 1) & 7)) << 4));
        }
        w_kstep = 128u;
    }
    // one LDS-DMA instruction: piece g (0..G-1) of k-tile kt into ring slot `slot`
    LMI_DEV void issue(int g, int kt, int slot) const {
        char* base = wave_base + slot * C::STAGE_BYTES;
        if (g < C::A_PASSES) {
            unsigned a_off = (unsigned)kt * 128u;                    // one k-tile = 128 bytes of every row
            if (AMODE == AMODE_PIXSHUF) {
                const int k0 = kt * (128 / ES);
                const int seg = k0 / ps_c;                           // 0..3 = (dh, dw)
                a_off = (unsigned)((((seg >> 1) * ps_grid + (seg & 1)) * lda + (k0 - seg * ps_c)) * ES);
            }
            glds16_buf(a_buf, a_src[g], a_off, base + g * (C::ROWS_PER_PASS * 128));
        } else {
            const int gw = g - C::A_PASSES;
            glds16_buf(w_buf, w_src[gw], (unsigned)kt * w_kstep, base + C::A_BYTES + gw * (C::ROWS_PER_PASS * 128));
        }
    }
};


// write phase for v_mfma_f32_32x32x16 accumulators acc[NI][MI]: lane (fr, fh) owns row fr and the quads n = ni*32 + q*8 + fh*4
template <typename C>
LMI_DEV void gemm_put32(const f32x16 (&acc)[C::NI][C::MI], int mi, int lane, char* stage) {
    constexpr int RS = GemmImage<C::WTN>::RS;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][q * 4 + e];
            *(f32x4*)(stage + fr * RS + (ni * 32 + q * 8 + fh * 4) * 4) = v;
        }
}
template <typename T, int EPI, int ACT, int AMODE, typename C, typename TA = T, bool LO4 = false>
__global__ void __launch_bounds__(C::NT) gemm_kernel(GemmArgs p) {
    typedef typename GemmFrag<TA>::type Frag;
    constexpr int KS = GemmFrag<TA>::KS, ES = (int)sizeof(TA);
    static_assert(!LO4 || (sizeof(TA) == 2 && AMODE == AMODE_PLAIN), "the low-bit correction phase follows a 16-bit pass over a plain A");
    LMI_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_id();
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int tiles_m = (p.M + C::BM - 1) / C::BM, tiles_n = (p.N + C::BN - 1) / C::BN;
    int tm, tn;
    if (!(LO4 ? gemm_tile_coords_sel(p, (int)blockIdx.x, tiles_m, tiles_n, tm, tn)
              : gemm_tile_coords((int)blockIdx.x, tiles_m, tiles_n, p.group_m, p.order, tm, tn))) return;
    const int m0 = tm * C::BM, n0 = tn * C::BN;

    GemmStager<AMODE, C, ES> stager;
    stager.init(p, m0, n0, tid, smem, wave);
    float* rstd_lds = (float*)(smem + C::SMEM);
    constexpr bool CAN_SCALE = GemmCanScale<EPI>::value;
    GemmRowScale<C> row_scale;
    if (CAN_SCALE && p.rowsq_in) row_scale.load(p, m0, tid);            // folded RMSNorm: partial sums requested ahead of the first DMA pieces

    f32x16 acc[C::NI][C::MI];
#pragma unroll
    for (int i = 0; i < C::NI; ++i)
#pragma unroll
        for (int j = 0; j < C::MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    GemmWOff<TA> w_off;
    w_off.init(p.w_packed, fr, fh);
    const int nt = p.K / (128 / ES);                                // k-tiles of 128 bytes per row
    // LO4: the ring simply continues over the fp4 k-tiles (global tile index tg = nt + u for fp4 tile u); the stager switches sources
    // when the first fp4 tile is issued, the fragment offsets / MFMA when it is multiplied
    const bool sel_tile = LO4 && gemm_tile_selected(p, m0, C::BM);   // (workgroup-uniform) no selected row: the 16-bit pass alone
    const int nt_all = sel_tile ? nt + p.K4 / 256 : nt;
    GemmLo4<C> lo4;
    if constexpr (LO4) lo4.init(p, m0, n0, tid, wave, wn, fr, smem);
    // VMEM operations per k-tile and thread: G operand pieces (+ the scale pieces of an fp4 tile).  The counted wait of tile t uses the
    // count of ITS phase: at the phase boundary the 16-bit count is the smaller one, i.e. the wait is at worst early-complete.
    constexpr int GL = C::G + (LO4 ? GemmLo4<C>::SP : 0);
    auto issue_tile_piece = [&](int g, int tg, int slot) {
        if (LO4 && tg >= nt) {
            if (g == 0) { if (tg == nt) stager.switch_lo4(p, m0, n0, tid); lo4.issue(tg - nt, slot); }
            stager.issue(g, tg - nt, slot);
        } else {
            stager.issue(g, tg, slot);
        }
    };

    // ---- prologue: D tiles in flight -----------------------------------------------------------------------------
#pragma unroll
    for (int d = 0; d < C::D; ++d)
        if (d < nt_all) {
#pragma unroll
            for (int g = 0; g < C::G; ++g) issue_tile_piece(g, d, d);
        }

    auto k_tile = [&](auto lo_tag, int t) {
        constexpr bool LO = decltype(lo_tag)::value;
        // tile t has landed once at most the D-1 younger tiles are outstanding (ring tail: everything)
        if (C::D >= 2 && t + C::D - 1 < nt_all) wait_vmcnt_barrier<(C::D >= 2 ? (LO ? GL : C::G) * (C::D - 1) : 0)>();
        else wait_vmcnt_barrier<0>();
        const int slot = t % C::STAGES;
        const char* a_t = smem + slot * C::STAGE_BYTES;
        const char* w_t = a_t + C::A_BYTES;
        const int t_issue = t + C::D;                                // k-tile whose loads are issued under this tile
        const bool do_issue = t_issue < nt_all;
        const int slot_issue = t_issue % C::STAGES;                  // == slot of tile t-1: free since the barrier
        int sc_lo[C::MI], sc_hi[C::MI];
        if constexpr (LO) {
#pragma unroll
            for (int i = 0; i < C::MI; ++i) lo4.read(slot, wm * C::WTM + i * 32 + fr, fh, sc_lo[i], sc_hi[i]);
        }
        static_for<KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            Frag af[C::MI], wf[C::NI];
#pragma unroll
            for (int i = 0; i < C::MI; ++i) af[i] = gemm_frag_load<TA>(a_t, wm * C::WTM + i * 32 + fr, ks, fh);
#pragma unroll
            for (int i = 0; i < C::NI; ++i) wf[i] = gemm_wfrag_load<TA>(w_t, wn * C::WTN + i * 32, fr, ks, fh, w_off);
            if (do_issue) {
#pragma unroll
                for (int g = ks * C::G / KS; g < (ks + 1) * C::G / KS; ++g) issue_tile_piece(g, t_issue, slot_issue);
            }
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < C::MI; ++mi) {
                    if constexpr (LO)
                        acc[ni][mi] = mfma32_fp4<0, (ks & 1) * 2>(frag_bits(wf[ni]), frag_bits(af[mi]), acc[ni][mi],
                                                                 lo4.w_sc[ni], ks < 2 ? sc_lo[mi] : sc_hi[mi]);
                    else
                        acc[ni][mi] = gemm_mma(wf[ni], af[mi], acc[ni][mi], p.scale_e8m0);
                }
        });
    };
    for (int t = 0; t < nt; ++t) k_tile(std::false_type{}, t);
    if constexpr (LO4) {
        if (sel_tile) {
            w_off.init(0, fr, fh);                                   // the fp4 weight image is row-major
            for (int t = nt; t < nt_all; ++t) k_tile(std::true_type{}, t);
        }
    }

    if (CAN_SCALE && p.rowsq_in) row_scale.finish(p, tid, rstd_lds);   // published by the barrier below
    raw_barrier();                                                 // every wave is done reading k-tiles: LDS is free
    gemm_epilogue<T, EPI, ACT, C, LO4>(p, [&](int mi, char* st) { gemm_put32<C>(acc, mi, lane, st); }, m0, n0, wm, wn, lane,
                                  smem + wave * (C::SMEM / (C::NT / 64)), rstd_lds, !LO4 || sel_tile);
}

// ------------------------------------------------------------------------------------------------------------------
// Staggered two-group schedule (8 waves, 2-slot ring).  Each k-step (one of the four 16-deep slices of a k-tile)
// is a LOAD segment {6 ds_read_b128 of this wave's fragments + this wave's share of the NEXT tile's LDS-DMA} and
// an MFMA segment {MI*NI MFMAs at raised priority}, each closed by a workgroup barrier.  Waves 4-7 run exactly
// one barrier behind waves 0-3 (one extra barrier at entry, balanced by one for waves 0-3 at exit), so on every
// SIMD — which hosts one wave of each group — one wave is always in its MFMA segment while its partner is in its
// LOAD segment: matrix pipe beside LDS/VMEM issue instead of both waves competing for the same pipe in lock-step.
// Hazards: fragments are read only in LOAD segments; the pieces of tile t+1 are issued in the LOAD segments of
// k-steps 0 and 1 of tile t (the slot's previous tenant, tile t-1, was last read one full segment earlier by the
// lagging group) and every wave drains its own pieces (vmcnt(0)) before the barrier that closes its k-step-3 LOAD
// segment, which precedes the first read of tile t+1 by either group.
// Prologue: k-tiles 0 AND 1 are requested at entry (both ring slots are free: the second tile's latency hides behind the wait
// for the first; -0.3 % on the C3 step).  Waves whose 64 columns lie entirely past N (N % 256 == 128: SigLIP q|k|v) skip their
// MFMAs and fragment reads, as the padding row blocks of the tail row-tile do.
// VAR 0 = as described (production); VAR 1 = the single-tile prologue of rounds 1-2 (kept as the A/B reference, gemm.config 6).
// Measured and dropped (profiles/README.md): no s_setprio; LDS-DMA pieces between the MFMAs (1-2 % faster in isolated bursts,
// 2.4 % slower over the power-capped step); L2 touches — of the own k-tile t + 2 ahead of its LDS-DMA pieces (+2.8 % step time
// with one touch per patch line, +17.5 % with every workgroup touching its own 512 lines: VMEM issue is what the LOAD segments
// are short of, not L2 hit rate), of the successor workgroup's first k-tiles (+0.4 %), of the residual epilogue's rows (+0.3 %);
// a strip-major tile order that lets all XCDs share one W strip (+0.4 %).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int EPI, int ACT, int AMODE, typename C, int VAR, typename TA = T, bool LO4 = false>
__global__ void __launch_bounds__(C::NT) gemm_stagger_kernel(GemmArgs p) {
    static_assert(C::STAGES == 2 && C::NT == 512, "staggered schedule: 8 waves, 2-slot ring");
    static_assert(!LO4 || (sizeof(TA) == 2 && AMODE == AMODE_PLAIN && VAR == 0), "the low-bit correction phase follows a 16-bit pass over a plain A");
    typedef typename GemmFrag<TA>::type Frag;
    // fp8: 2 k-steps of 64 per k-tile (8 MFMAs of 64 cycles each per MFMA segment); all DMA pieces of the next tile go out in
    // k-step 0's LOAD segment and are drained in k-step 1's: one MFMA segment of flight, as the 16-bit schedule has two of half the length
    constexpr int KS = GemmFrag<TA>::KS, ES = (int)sizeof(TA), KS_ISSUE = KS / 2;
    LMI_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_id();
    const int grp = wave >> 2;                  // waves w and w+4 share a SIMD (measured: other pairings lose 20-25 %)
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int tiles_m = (p.M + C::BM - 1) / C::BM, tiles_n = (p.N + C::BN - 1) / C::BN;
    int tm, tn;
    if (!(LO4 ? gemm_tile_coords_sel(p, (int)blockIdx.x, tiles_m, tiles_n, tm, tn)
              : gemm_tile_coords((int)blockIdx.x, tiles_m, tiles_n, p.group_m, p.order, tm, tn))) return;
    const int m0 = tm * C::BM, n0 = tn * C::BN;

    GemmStager<AMODE, C, ES> stager;
    stager.init(p, m0, n0, tid, smem, wave);
    auto issue_piece = [&](int g, int kt, int slot) { stager.issue(g, kt, slot); };
    float* rstd_lds = (float*)(smem + C::SMEM);
    constexpr bool CAN_SCALE = GemmCanScale<EPI>::value;
    GemmRowScale<C> row_scale;
    if (CAN_SCALE && p.rowsq_in) row_scale.load(p, m0, tid);            // folded RMSNorm: partial sums requested ahead of the first DMA pieces

    f32x16 acc[C::NI][C::MI];
#pragma unroll
    for (int i = 0; i < C::NI; ++i)
#pragma unroll
        for (int j = 0; j < C::MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    GemmWOff<TA> w_off;
    w_off.init(p.w_packed, fr, fh);
    const int nt = p.K / (128 / ES);                                // k-tiles of 128 bytes per row
    GemmLo4<C> lo4;
    if constexpr (LO4) lo4.init(p, m0, n0, tid, wave, wn, fr, smem);
#pragma unroll
    for (int g = 0; g < C::G; ++g) issue_piece(g, 0, 0);
    constexpr bool TWO = (VAR == 0);                                // two-k-tile prologue
    if (TWO && nt > 1) {
#pragma unroll
        for (int g = 0; g < C::G; ++g) issue_piece(g, 1, 1);
        wait_vmcnt_barrier<C::G>();                                // k-tile 0 landed; k-tile 1 in flight
    } else {
        wait_vmcnt_barrier<0>();
    }
    if (grp == 1) raw_barrier();                                   // waves 4-7 run one barrier behind

    // 32-row blocks of this wave's tile that hold rows below M: the tail row-tile (M = 7187 leaves 19 rows of 256) skips the
    // MFMAs and fragment reads of the blocks that are entirely padding — they would cost as much energy as useful ones, and
    // the step runs at the package power cap.  Barriers and LDS-DMA pieces are unaffected.
    const int rows_left = p.M - (m0 + wm * C::WTM);
    int nmi = rows_left >= C::WTM ? C::MI : (rows_left <= 0 ? 0 : (rows_left + 31) >> 5);
    if (n0 + wn * C::WTN >= p.N) nmi = 0;              // this wave's columns are all past N (masked on store anyway)
    // ISSUE (0 = nothing, 1 = the operand pieces of the next k-tile, 2 = those + the block-scale pieces of an fp4 k-tile), FULL (no
    // padding blocks) and LO (an fp4 k-tile of the correction phase) are compile-time: a runtime test per LDS-DMA piece or per MFMA
    // would cut the MFMA segment into scheduling regions with a branch each.  `tg` = global k-tile index (slot tg & 1), `kt_next` =
    // index of the next k-tile inside ITS phase.
    auto tile = [&](auto issue_tag, auto full_tag, auto lo_tag, int tg, int kt_next) {
        constexpr int ISSUE = decltype(issue_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value, LO = decltype(lo_tag)::value;
        const char* a_t = smem + (tg & 1) * C::STAGE_BYTES;
        const char* w_t = a_t + C::A_BYTES;
        int sc_lo[C::MI], sc_hi[C::MI];
        static_for<KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            // ---- LOAD segment ---------------------------------------------------------------------------------
            Frag af[C::MI], wf[C::NI];
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
                if (FULL || i < nmi) af[i] = gemm_frag_load<TA>(a_t, wm * C::WTM + i * 32 + fr, ks, fh);
#pragma unroll
            for (int i = 0; i < C::NI; ++i)
                if (FULL || nmi > 0) wf[i] = gemm_wfrag_load<TA>(w_t, wn * C::WTN + i * 32, fr, ks, fh, w_off);
            if constexpr (LO && ks == 0) {
#pragma unroll
                for (int i = 0; i < C::MI; ++i)
                    if (FULL || i < nmi) lo4.read(tg & 1, wm * C::WTM + i * 32 + fr, fh, sc_lo[i], sc_hi[i]);
            }
            if (ks < KS_ISSUE && ISSUE) {
#pragma unroll
                for (int g = ks * C::G / KS_ISSUE; g < (ks + 1) * C::G / KS_ISSUE; ++g) issue_piece(g, kt_next, (tg + 1) & 1);
                if constexpr (LO4 && ISSUE == 2 && ks == 0) lo4.issue(kt_next, (tg + 1) & 1);
            }
            if (ks == KS - 1) wait_vmcnt_barrier<0>(); else raw_barrier();
            // ---- MFMA segment ---------------------------------------------------------------------------------
            sched_fence();
            setprio_hi();
#pragma unroll
            for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < C::MI; ++mi)
                    if (FULL || mi < nmi) {
                        if constexpr (LO)
                            acc[ni][mi] = mfma32_fp4<0, (ks & 1) * 2>(frag_bits(wf[ni]), frag_bits(af[mi]), acc[ni][mi],
                                                                     lo4.w_sc[ni], ks < 2 ? sc_lo[mi] : sc_hi[mi]);
                        else
                            acc[ni][mi] = gemm_mma(wf[ni], af[mi], acc[ni][mi], p.scale_e8m0);
                    }
            setprio_lo();
            sched_fence();
            raw_barrier();
        });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    const bool sel_tile = LO4 && gemm_tile_selected(p, m0, C::BM);   // (workgroup-uniform) no selected row: the 16-bit pass alone
    auto run = [&](auto full_tag) {
        const std::false_type hi{};
        auto fast_seq = [&]() {                                    // the 16-bit k-tiles alone
            if (TWO && nt > 1) {
                tile(I0{}, full_tag, hi, 0, 0);                    // k-tile 1 is already in flight; its pieces drain in k-step 3 as usual
                for (int t = 1; t + 1 < nt; ++t) tile(I1{}, full_tag, hi, t, t + 1);
                tile(I0{}, full_tag, hi, nt - 1, 0);
            } else {
                for (int t = 0; t + 1 < nt; ++t) tile(I1{}, full_tag, hi, t, t + 1);
                tile(I0{}, full_tag, hi, nt - 1, 0);
            }
        };
        if constexpr (LO4) {
            if (!sel_tile) { fast_seq(); return; }                 // (workgroup-uniform) no row of this tile carries a residual image
            // 16-bit tiles 0 .. nt-1 (nt >= 2, checked by the launcher), then fp4 tiles 0 .. n4-1 in the same ring: the last 16-bit tile
            // issues fp4 tile 0 (its operand pieces were issued under tile nt-2 and drained there, so the stager may switch sources)
            const int n4 = p.K4 / 256;
            tile(I0{}, full_tag, hi, 0, 0);
            for (int t = 1; t + 1 < nt; ++t) tile(I1{}, full_tag, hi, t, t + 1);
            stager.switch_lo4(p, m0, n0, tid);
            tile(I2{}, full_tag, hi, nt - 1, 0);
            w_off.init(0, fr, fh);                                 // the fp4 weight image is row-major
            const std::true_type lo{};
            for (int u = 0; u + 1 < n4; ++u) tile(I2{}, full_tag, lo, nt + u, u + 1);
            tile(I0{}, full_tag, lo, nt + n4 - 1, 0);
        } else {
            fast_seq();
        }
    };
    if (nmi == C::MI) run(std::true_type{}); else run(std::false_type{});
    if (grp == 0) raw_barrier();                                   // balance the barrier count
    if (CAN_SCALE && p.rowsq_in) {                                 // (workgroup-uniform) row scales of the folded RMSNorm -> LDS, one more
        row_scale.finish(p, tid, rstd_lds);                        // barrier to publish them before any wave's epilogue reads them
        raw_barrier();
    }
    // past its last barrier a wave knows that every other wave has finished its last LOAD segment: LDS is free
    gemm_epilogue<T, EPI, ACT, C, LO4>(p, [&](int mi, char* st) { gemm_put32<C>(acc, mi, lane, st); }, m0, n0, wm, wn, lane,
                                  smem + wave * (C::SMEM / (C::NT / 64)), rstd_lds, !LO4 || sel_tile);
}

}  // namespace lmi
