// attention.h — variable-length FlashAttention-2 forward for gfx950 (wave64, v_mfma_f32_32x32x16).
//
// Serves both attention shapes on the Leopard prefill path:
//   * SigLIP: non-causal, 16 heads x 72, one 676-token sequence per tile (cu_seqlens = 0,676,1352,..)
//     third-party SiglipAttention; in-tree analogue megatron_patch/model/llava/transformer.py:456-512
//     (flash_attn_varlen_func with cu_seqlens)
//   * Llama-3.1: causal GQA 32 q / 8 kv heads x 128 over the merged [img0..imgN | text] sequence
//     (megatron_patch/model/llava/transformer.py:678-885; GQA repeat :829-836)
//
// Work split: one workgroup = 128 query rows of one (sequence, head) = 4 waves x 32 rows; K/V walk in
// 64-key tiles staged through registers into padded LDS images (the next tile's global loads are in
// flight while the current tile is computed).
//
// Register-level design (everything stays in the lane that owns query row q = lane&31):
//   S^T = K . Q^T   mfma32(A = K rows, B = Q rows)  -> lane (q, hi) holds 16 keys of its own row per 32-key block;
//                   row max / row sum are in-lane + ONE exchange with lane^32.
//   O^T = V^T . P^T mfma32(A = V^T, B = P^T)        -> the B operand is exactly the lane's own P registers (the
//                   contraction index is a free permutation of keys, so no cross-lane shuffle of P is needed);
//                   the A operand (8 keys for one d) comes from ds_read_b64_tr_b16 transpose reads of the
//                   row-major V image; O^T's column is q, so the online-softmax rescale is a per-lane scalar.
// Softmax statistics and accumulators are fp32; P is rounded to the 16-bit compute type for the PV MFMA.
#pragma once
#include <type_traits>
#include "lmi_device.h"

namespace lmi {

struct AttnArgs {
    const void* q;            // [total_q, ...] head h at q + row*ldq + h*D      (elements of T)
    const void* k;            // head kvh at k + row*ldk + kvh*D
    const void* v;
    void* out;                // [total_q, ...] head h at out + row*ldo + h*D
    const int* cu_q;          // [nseq+1]
    const int* cu_k;          // [nseq+1]
    const int* k_len;         // optional [nseq] (LDS-DMA kernel only): sequence s's keys are rows [cu_k[s], cu_k[s] + k_len[s]) — a pooled,
                              // strided KV cache of a decode batch, where cu_k holds the slots' first rows and only k_len changes per step
    int ldq, ldk, ldv, ldo;
    int n_heads, n_kv_heads;
    float scale;              // softmax scale (head_dim^-0.5)
    int window;               // sliding window (Mistral): query i sees keys j with i - j < window; 0 = unlimited
    int n_qblocks;            // ceil(max_seqlen_q / ATT_BQ) (1-D grid decode of the LDS-DMA kernel)
    // split-KV (decode: few query rows against a long cache): the key range is cut into n_splits chunks of split_tiles
    // 64-key tiles, each workgroup writes an unnormalised partial (O fp32, reference max, row sum) and
    // attn_combine_kernel merges them.  n_splits <= 1: single pass, normalised output straight to `out`.
    int n_splits, split_tiles, part_rows;
    float* part_o;            // [n_splits, part_rows, n_heads, D]
    float* part_ml;           // [n_splits, part_rows, n_heads, 2]  (m, l)
    float* out_f32;           // LDS-DMA kernel, optional: instead of `out`, the normalised output in fp32 [total_q, ...] (row stride ldo32) — the
    int ldo32;                // split-operand precision mode hands it to lmi_split_hi_lo instead of rounding it to 16 bits here
    void* out_fp8;            // LDS-DMA kernel, optional: instead of `out`, write e4m3(O * out_fp8_scale) bytes [total_q, ...] (row stride ldo8) —
    float out_fp8_scale;      // the o_proj operand of the fp8 schedule straight from the attention epilogue (no conversion launch)
    int ldo8;
    // LDS-DMA kernel, optional (beside `out`): the MX fp4 image of the rounding residuals O - T(O) and its block scales, for the low-bit
    // correction phase of the projection that consumes `out` (gemm.h LO4).  The image has its OWN k order: head h occupies the 32-element
    // blocks [h * NDB, (h + 1) * NDB) — head_dim 128: the natural order; 72 / 96: every head padded to 96 (zero codes), so that no block
    // straddles two heads (= two workgroups); the projection's weight image is laid out the same way (leopard_amd.engine).
    uint8_t* out4;            // [total_q, ld_out4 bytes]: head h, block db at byte (h * NDB + db) * 16
    uint8_t* out4_scale;      // [total_q, ld_out4s]: byte h * NDB + db
    int ld_out4, ld_out4s;
    const uint8_t* row_sel;   // [total_q] or null: the rows whose image is wanted (gemm.h GemmArgs::row_sel); the others only get their 16-bit row
    int gqa_pack;             // LDS-DMA kernel, decode: a workgroup's 4 waves take the 4 query heads of ONE kv head (32 query rows per block)
    int check_k_extent;       // 1 = the launcher could not bound a sequence's K / V extent (< 4 GiB): the kernel checks (and traps)
};

constexpr int ATT_BQ = 128, ATT_BKV = 64, ATT_THREADS = 256;
constexpr float ATT_DEFER_LOG2 = 8.0f;        // LDS-DMA kernels: the softmax reference may lag the row maximum by 2^8

template <int D> struct AttnGeom {
    static constexpr int DK = (D + 15) / 16 * 16;            // QK^T contraction length (zero padded)
    static constexpr int NKS = DK / 16;
    static constexpr int NDB = (D + 31) / 32;                // 32-row blocks of O^T
    static constexpr int CH = D / 8;                         // 16-byte chunks per row in HBM
    // K image row stride: an odd number of 16-byte chunks -> the 16 rows of a ds_read_b128 lane group land on
    // 16 distinct slots of the 256-byte bank row.
    static constexpr int KSTRIDE = ((DK / 8) | 1) * 16;
    // V image row stride: odd multiple of 64 bytes -> the 4 key rows of one transpose read use disjoint banks.
    static constexpr int VROW = NDB * 64;
    static constexpr int VSTRIDE = ((VROW / 64) & 1) ? VROW : VROW + 64;
    static constexpr int K_BYTES = ATT_BKV * KSTRIDE;
    static constexpr int V_BYTES = ATT_BKV * VSTRIDE;
    static constexpr int SMEM = K_BYTES + V_BYTES;
    static constexpr int LD_ITERS = (ATT_BKV * CH + ATT_THREADS - 1) / ATT_THREADS;
};

// USE_TR = false replaces the hardware transpose reads of V by plain 16-bit LDS gathers (slow; kept as an
// on-device cross-check of the ds_read_b64_tr_b16 addressing — tests run both).
template <typename T, int D, bool CAUSAL, bool USE_TR>
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd_kernel(AttnArgs p) {
    typedef AttnGeom<D> G;
    typedef typename vec_of<T>::x8 T8;
    typedef typename vec_of<T>::x4 T4;
    LMI_DYN_SMEM(smem);
    char* k_lds = smem;
    char* v_lds = smem + G::K_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int seq = blockIdx.z, head = blockIdx.y;
    const int kvh = head / (p.n_heads / p.n_kv_heads);
    const int q_beg = p.cu_q[seq], len_q = p.cu_q[seq + 1] - q_beg;
    const int k_beg = p.cu_k[seq], len_k = p.cu_k[seq + 1] - k_beg;
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x;          // heavy (late) causal blocks first
    const int q0 = qb * ATT_BQ;
    if (q0 >= len_q) return;                                      // whole block exits together
    const int shift = len_k - len_q;                              // causal: key j visible to query i iff j <= i + shift
    int kv_end = len_k;
    if (CAUSAL) kv_end = imin(len_k, q0 + ATT_BQ + shift);
    const int n_tiles = (kv_end + ATT_BKV - 1) / ATT_BKV;

    // zero the K pad columns once (QK^T contracts over DK >= D; stale LDS bits could be NaN)
    if (G::DK > D) {
        for (int i = tid; i < ATT_BKV * (G::DK - D) / 8; i += ATT_THREADS) {
            const int r = i / ((G::DK - D) / 8), c = i % ((G::DK - D) / 8);
            *(u32x4*)(k_lds + r * G::KSTRIDE + (G::CH + c) * 16) = u32x4{0, 0, 0, 0};
        }
    }

    // ---- Q fragments straight from HBM: lane (q = fr, hi = fh) holds Q[q][16ks + 8hi .. +7] -------------------
    const int my_q = q0 + wave * 32 + fr;
    const int my_q_ld = imin(my_q, len_q - 1);
    const T* q_row = (const T*)p.q + (long)(q_beg + my_q_ld) * p.ldq + head * D;
    T8 qf[G::NKS];
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
        const int c = 2 * ks + fh;
        if (c < G::CH) qf[ks] = *(const T8*)(q_row + c * 8);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (T)0.0f;
        }
    }

    // ---- K/V staging through registers ------------------------------------------------------------------
    const T* k_base = (const T*)p.k + (long)k_beg * p.ldk + kvh * D;
    const T* v_base = (const T*)p.v + (long)k_beg * p.ldv + kvh * D;
    u32x4 kreg[G::LD_ITERS], vreg[G::LD_ITERS];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int it = 0; it < G::LD_ITERS; ++it) {
            const int i = tid + it * ATT_THREADS;
            if (i < ATT_BKV * G::CH) {
                const int r = i / G::CH, c = i - r * G::CH;
                const int key = imin(t * ATT_BKV + r, len_k - 1);
                kreg[it] = *(const u32x4*)(k_base + (long)key * p.ldk + c * 8);
                vreg[it] = *(const u32x4*)(v_base + (long)key * p.ldv + c * 8);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G::LD_ITERS; ++it) {
            const int i = tid + it * ATT_THREADS;
            if (i < ATT_BKV * G::CH) {
                const int r = i / G::CH, c = i - r * G::CH;
                *(u32x4*)(k_lds + r * G::KSTRIDE + c * 16) = kreg[it];
                *(u32x4*)(v_lds + r * G::VSTRIDE + c * 16) = vreg[it];
            }
        }
    };

    f32x16 o_acc[G::NDB];
#pragma unroll
    for (int i = 0; i < G::NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;              // scale * log2(e)
    const int wave_q_lo = q0 + wave * 32, wave_q_hi = wave_q_lo + 31;

    // transpose-read lane geometry (see lmi_device.h ds_read_tr16_b64)
    const int tr_j = (lane & 15) >> 2, tr_g = lane & 3, tr_half = (lane >> 4) & 1;

    if (n_tiles > 0) load_tile(0);
    for (int t = 0; t < n_tiles; ++t) {
        __syncthreads();                                          // everyone is done with tile t-1
        store_tile();
        __syncthreads();
        if (t + 1 < n_tiles) load_tile(t + 1);                    // flies under the MFMAs below
        const int kv0 = t * ATT_BKV;
        if (CAUSAL && kv0 > wave_q_hi + shift) continue;          // tile fully masked for this wave (uniform)

        // ---- S^T = K . Q^T ----------------------------------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < G::NKS; ++ks) {
                const T8 kf = *(const T8*)(k_lds + (b * 32 + fr) * G::KSTRIDE + (2 * ks + fh) * 16);
                s[b] = mfma32(kf, qf[ks], s[b]);
            }
        }
        // ---- mask -----------------------------------------------------------------------------------
        const bool need_mask = (kv0 + ATT_BKV > len_k) || (CAUSAL && (kv0 + ATT_BKV - 1 > wave_q_lo + shift)) ||
                               (CAUSAL && p.window > 0 && kv0 <= wave_q_hi + shift - p.window);
        if (need_mask) {
            const int lim = CAUSAL ? imin(len_k - 1, my_q + shift) : len_k - 1;   // last visible key
            const int lo = (CAUSAL && p.window > 0) ? my_q + shift - p.window + 1 : 0;   // first visible key
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key > lim || key < lo) s[b][r] = -INFINITY;
                }
        }
        // ---- online softmax ---------------------------------------------------------------------------
        float mx = s[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][r]);
        mx = fmaxf(mx, shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        // rows that see no key at all so far (only query rows past len_q+shift<0 corner cases) keep m = -inf
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2((m_run - m_use) * c2);
        m_run = m_new;
        const float mc = m_use * c2;
        float psum = 0.f;
        T8 pf[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(s[b][r] * c2 - mc);
                psum += pv;
                pf[b][r >> 3][r & 7] = (T)pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < G::NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;

        // ---- O^T += V^T . P^T -------------------------------------------------------------------------
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kbase = b * 32 + 16 * u + 4 * fh;
                const char* a0 = v_lds + (kbase + tr_j) * G::VSTRIDE + (16 * tr_half + 4 * tr_g) * 2;
                if (USE_TR) {
                    u32x4 vraw[G::NDB];
                    ds_read_tr16_batch<G::NDB, 8 * G::VSTRIDE>(a0, vraw);
#pragma unroll
                    for (int db = 0; db < G::NDB; ++db)
                        o_acc[db] = mfma32(__builtin_bit_cast(T8, vraw[db]), pf[b][u], o_acc[db]);
                } else {
#pragma unroll
                    for (int db = 0; db < G::NDB; ++db) {
                        T8 vf;
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            vf[j] = *(const T*)(v_lds + (kbase + (j & 3) + 8 * (j >> 2)) * G::VSTRIDE + (db * 32 + fr) * 2);
                        o_acc[db] = mfma32(vf, pf[b][u], o_acc[db]);
                    }
                }
            }
    }

    // ---- finish: O / l, store rows of this lane's query ---------------------------------------------------
    const float l_tot = l_run + shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (my_q < len_q) {
        T* o_row = (T*)p.out + (long)(q_beg + my_q) * p.ldo + head * D;
#pragma unroll
        for (int db = 0; db < G::NDB; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = db * 32 + 8 * qd + 4 * fh;
                if (d < D) {
                    T4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (T)(o_acc[db][qd * 4 + e] * inv);
                    *(T4*)(o_row + d) = o;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_fwd_dma_kernel — production variant (head_dim 128 and 72).  Same register-level design as attn_fwd_kernel, but K and V
// tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) into a 2-slot ring: no staging VGPRs, no ds_write
// pass, one counted wait + one raw barrier per 64-key tile, and the register budget fits 2 waves per SIMD so a
// second workgroup's MFMAs run under this one's softmax.  LDS-DMA writes lane-linearly, so the LDS images are
// unpadded [64][256 B] and de-conflicted by XOR swizzles applied to the per-lane SOURCE chunk and again on read:
//   K: 16-byte chunk c of row r at c ^ (r & 15)          -> conflict-free ds_read_b128 (16 distinct rows / group)
//   V: 16-byte chunk c of row r at c ^ ((r & 3) << 2)    -> the 4 key rows of one transpose read use 4 distinct
//                                                            64-byte bank groups
// ------------------------------------------------------------------------------------------------------------------
template <int D> struct AttnDmaGeom {
    static constexpr int ROWB = D * 2;                        // bytes per K/V row in LDS (unpadded: LDS-DMA is lane-linear)
    static constexpr int CH = D / 8;                          // 16-byte chunks per row
    static constexpr int NKS = (D + 15) / 16;                 // 16-deep QK^T steps (Q zero-padded past D)
    static constexpr int NDB = (D + 31) / 32;                 // 32-row blocks of O^T
    static constexpr int TILE_BYTES = ATT_BKV * ROWB;         // one K or V tile image
    static constexpr int PIECES = TILE_BYTES / 1024;          // 1-KiB LDS-DMA wave pieces per image
    static constexpr int PPW = (PIECES + 3) / 4;              // pieces per wave (4 waves)
    static constexpr int SMEM = 4 * TILE_BYTES + 64;          // 2 slots x (K + V) (+ slack read by padded columns)
    static constexpr bool SWZ = (D == 128);                   // XOR swizzles need power-of-two rows; 144-byte rows
                                                              // (9 chunks, odd) are conflict-free for b128 as they are
    static constexpr bool VPERM = (D == 72);                  // ... but not for the transposed V reads: see att_vpos72
};
// d = 72: a ds_read_b64_tr_b16 lane group takes the four 16-byte chunks of one 32-wide d-block from four consecutive key
// rows; with 144-byte rows those 16 chunks fall on only 12 of the 16 chunk slots of the 256-byte bank row (PMC: 35 % of the
// kernel's LDS cycles were conflict cycles).  The V image therefore stores chunk c of key row r at position
// att_vpos72(r & 3, c) of the row (a free permutation: LDS-DMA lanes pick their source chunk): for each d-block the four
// rows' chunks then cover all 16 slots exactly once, and the rows' chunk 8 (d 64..71) land on four different slots.
LMI_DEV int att_vpos72(int r4, int c) {
    constexpr unsigned long long POS[4] = {0x087654321ull, 0x876543210ull, 0x087216543ull, 0x876105432ull};   // nibble c = position
    return (int)((POS[r4] >> (4 * c)) & 15);
}
LMI_DEV int att_vchunk72(int r4, int pos) {                    // inverse: which chunk sits at `pos`
    constexpr unsigned long long INV[4] = {0x765432108ull, 0x876543210ull, 0x763210548ull, 0x876321054ull};
    return (int)((INV[r4] >> (4 * pos)) & 15);
}

template <typename T, int D, bool CAUSAL, bool STREAM = false>      // STREAM: non-temporal K / V loads (decode: each tile is read once)
__global__ void __launch_bounds__(ATT_THREADS, 2) attn_fwd_dma_kernel(AttnArgs p) {
    constexpr int NW = ATT_THREADS / 64, BQ = ATT_BQ, PPW = AttnDmaGeom<D>::PPW;
    typedef AttnDmaGeom<D> G;
    constexpr int NKS = G::NKS, NDB = G::NDB;
    typedef typename vec_of<T>::x8 T8;
    LMI_DYN_SMEM(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int fr = lane & 31, fh = lane >> 5;
    // ---- 1-D grid -> (sequence, query block, head).  Heads vary fastest so that (a) the heavy late causal blocks of ALL
    // heads are dispatched first and the light ones fill the tail, and (b) workgroup b runs on XCD b % 8 and takes kv head
    // b % n_kv_heads: with 8 kv heads each XCD's L2 serves one kv head's K/V stream to every query head that shares it.
    // GQA-packed decode (p.gqa_pack; decode launchers set it when n_heads == NW * n_kv_heads and there are <= 32 query rows per
    // sequence): the workgroup belongs to one KV head and its NW waves take that head's NW query heads — every wave has real rows
    // (a plain decode block has one real row in wave 0 and three idle waves) and the K / V tiles are DMA'd once for the group
    // instead of once per query head.
    const int bid = (int)blockIdx.x;
    const bool pack = p.gqa_pack != 0;
    const int grid_heads = pack ? p.n_kv_heads : p.n_heads;
    const int h_idx = bid % grid_heads;
    int rest = bid / grid_heads, split = 0;
    if (p.n_splits > 1) { split = rest % p.n_splits; rest /= p.n_splits; }
    const int qb = p.n_qblocks - 1 - rest % p.n_qblocks, seq = rest / p.n_qblocks;
    const int kvh = h_idx % p.n_kv_heads;
    const int head = pack ? kvh * NW + wave : kvh * (p.n_heads / p.n_kv_heads) + h_idx / p.n_kv_heads;
    const int q_beg = p.cu_q[seq], len_q = p.cu_q[seq + 1] - q_beg;
    const int k_beg = p.cu_k[seq], len_k = p.k_len ? p.k_len[seq] : p.cu_k[seq + 1] - k_beg;
    const int bq = pack ? 32 : BQ;                                 // query rows per workgroup
    const int q0 = qb * bq;
    if (q0 >= len_q) return;
    const int shift = len_k - len_q;
    int kv_end = len_k;
    if (CAUSAL) kv_end = imin(len_k, q0 + bq + shift);
    const int n_tiles = (kv_end + ATT_BKV - 1) / ATT_BKV;
    const int wave_q_lo = pack ? q0 : q0 + wave * 32, wave_q_hi = wave_q_lo + 31;
    // tiles this wave computes: the later ones are fully masked for its 32 rows (it still issues its DMA pieces and
    // joins the barriers for the other waves in the drain loop below)
    int my_tiles = n_tiles;
    if (CAUSAL) my_tiles = imax(0, imin(n_tiles, (wave_q_hi + shift) / ATT_BKV + 1));
    // split-KV: this workgroup only walks tiles [t_begin, t_end)
    const int t_begin = p.n_splits > 1 ? imin(n_tiles, split * p.split_tiles) : 0;
    const int t_end = p.n_splits > 1 ? imin(n_tiles, t_begin + p.split_tiles) : n_tiles;
    my_tiles = imin(my_tiles, t_end);
    if (wave_q_lo >= len_q) my_tiles = t_begin;                    // all 32 rows of this wave are past the sequence: feed and sync only

    const int my_q = wave_q_lo + fr;
    const T* q_row = (const T*)p.q + (long)(q_beg + imin(my_q, len_q - 1)) * p.ldq + head * D;
    T8 qf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = 2 * ks + fh;
        if (c < G::CH) qf[ks] = *(const T8*)(q_row + c * 8);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = (T)0.0f;
        }
    }

    // ---- LDS-DMA sources: wave w owns pieces w, w+4, ...; lane l of piece q is chunk q*64+l of the tile image.  K and V of
    // this (sequence, kv head) are addressed through buffer resources: per-lane byte offsets are loop invariant, the tile
    // advances a scalar offset, and rows past len_k read as zeros (no address clamp in the loop).
    const T* k_base = (const T*)p.k + (long)k_beg * p.ldk + kvh * D;
    const T* v_base = (const T*)p.v + (long)k_beg * p.ldv + kvh * D;
    // 32-bit buffer offsets: one sequence's K (or V) rows must span < 4 GiB (ldk = 6144 halves: 349 k keys).  Where the host knows
    // the longest key sequence (self-attention: cu_k == cu_q; decode: max_seqlen_k) the launcher has already returned LMI_EINVAL;
    // only cross-attention launches, whose key lengths exist on the device alone, carry the check into the kernel.
    if (p.check_k_extent && (((long)len_k * p.ldk + D) * 2 >= (1L << 32) || ((long)len_k * p.ldv + D) * 2 >= (1L << 32))) lmi_trap();
    const BufRsrc k_buf = make_buf(k_base, len_k > 0 ? (unsigned)(((long)(len_k - 1) * p.ldk + D) * 2) : 0u);
    const BufRsrc v_buf = make_buf(v_base, len_k > 0 ? (unsigned)(((long)(len_k - 1) * p.ldv + D) * 2) : 0u);
    unsigned p_ko[PPW], p_vo[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int ci = (wave + NW * i) * 64 + lane;
        const int r = ci / G::CH, c = ci - r * G::CH;
        p_ko[i] = (unsigned)(r * p.ldk + ((G::SWZ ? (c ^ (r & 15)) : c) << 3)) * 2u;
        const int cv = G::VPERM ? att_vchunk72(r & 3, c) : (G::SWZ ? (c ^ ((r & 3) << 2)) : c);
        p_vo[i] = (unsigned)(r * p.ldv + (cv << 3)) * 2u;
    }
    // piece j of this wave for tile t: j < PPW are its K pieces, the rest its V pieces
    auto issue_piece = [&](int j, int t, int slot) {
        const int i = j < PPW ? j : j - PPW;
        if ((G::PIECES % NW) != 0 && wave + NW * i >= G::PIECES) return;      // wave-uniform; only ragged piece counts (d = 72) branch
        char* dst = smem + slot * 2 * G::TILE_BYTES + (j < PPW ? 0 : G::TILE_BYTES) + (wave + NW * i) * 1024;
        if (j < PPW) glds16_buf<STREAM ? 2 : 0>(k_buf, p_ko[i], (unsigned)t * (unsigned)(ATT_BKV * 2) * (unsigned)p.ldk, dst);
        else glds16_buf<STREAM ? 2 : 0>(v_buf, p_vo[i], (unsigned)t * (unsigned)(ATT_BKV * 2) * (unsigned)p.ldv, dst);
    };
    auto issue_tile = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j < 2 * PPW; ++j) issue_piece(j, t, slot);
    };

    f32x16 o_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = p.scale * 1.4426950408889634f;

    // K fragment reads: lane (row fr of a 32-key block, k-chunk 2ks+fh); the swizzle term depends on the lane only
    int k_off[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = 2 * ks + fh;
        k_off[ks] = fr * G::ROWB + ((G::SWZ ? (c ^ (fr & 15)) : c) << 4);
    }
    // transpose-read lane geometry: lane i = 4j+g of a 16-lane group addresses key row j, 8-byte column group g;
    // with the V swizzle the 64-byte granule of d-block db is db ^ j (constant per lane)
    const int tr_j = (lane & 15) >> 2, tr_g = lane & 3, tr_half = (lane >> 4) & 1;
    int v_off[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        if (G::VPERM) {
            // lanes past the last real chunk (d >= 72) repeat chunk 8's address: identical addresses broadcast
            const int c = imin(db * 4 + tr_half * 2 + (tr_g >> 1), G::CH - 1);
            v_off[db] = (4 * fh + tr_j) * G::ROWB + att_vpos72(tr_j, c) * 16 + (tr_g & 1) * 8;
        } else {
            v_off[db] = (4 * fh + tr_j) * G::ROWB + ((G::SWZ ? (db ^ tr_j) : db) << 6) + tr_half * 32 + tr_g * 8;
        }
    }

    if (t_begin < t_end) issue_tile(t_begin, 0);
    int t = t_begin;
    LMI_PROF_DECL();
    for (; t < my_tiles; ++t) {
        LMI_PROF_MARK(0);
        wait_vmcnt_barrier<0>();                                   // tile t landed; slot of tile t-1 is free
        LMI_PROF_MARK(1);
        LMI_PROF_MARK(2);
        const int kv0 = t * ATT_BKV;
        const char* k_lds = smem + ((t - t_begin) & 1) * 2 * G::TILE_BYTES;
        const char* v_lds = k_lds + G::TILE_BYTES;

        f32x16 s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
        // S^T = K . Q^T.  K fragment reads run three MFMA pairs ahead of their use, and the LDS-DMA pieces of tile t+1 (an
        // LDS-DMA instruction costs ~60 issue cycles) ride one per k-step between the MFMA pairs instead of in a burst.
        auto qk = [&](auto with_dma) {
            constexpr bool DMA = decltype(with_dma)::value;
            constexpr int AHEAD = 3;                                 // K fragment pairs in flight ahead of their MFMAs
            T8 kf[NKS][2];
            sched_fence();
#pragma unroll
            for (int ks = 0; ks < AHEAD && ks < NKS; ++ks)
#pragma unroll
                for (int b = 0; b < 2; ++b) kf[ks][b] = *(const T8*)(k_lds + b * 32 * G::ROWB + k_off[ks]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sched_fence();                                       // one k-step per scheduling region: reads, 2 MFMAs, 1 DMA piece
                if (ks + AHEAD < NKS) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) kf[ks + AHEAD][b] = *(const T8*)(k_lds + b * 32 * G::ROWB + k_off[ks + AHEAD]);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    s[b] = mfma32(kf[ks][b], qf[ks], s[b]);
                }
                if (DMA && ks < 2 * PPW) issue_piece(ks, t + 1, (t + 1 - t_begin) & 1);
            }
            sched_fence();
            if (DMA) {
#pragma unroll
                for (int j = NKS; j < 2 * PPW; ++j) issue_piece(j, t + 1, (t + 1 - t_begin) & 1);
            }
        };
        if (t + 1 < t_end) qk(std::true_type{}); else qk(std::false_type{});
        LMI_PROF_TOUCH(s[0][15]); LMI_PROF_TOUCH(s[1][15]);
        LMI_PROF_MARK(3);
        const bool need_mask = (kv0 + ATT_BKV > len_k) || (CAUSAL && (kv0 + ATT_BKV - 1 > wave_q_lo + shift)) ||
                               (CAUSAL && p.window > 0 && kv0 <= wave_q_hi + shift - p.window);
        if (need_mask) {
            const int lim = CAUSAL ? imin(len_k - 1, my_q + shift) : len_k - 1;   // last visible key
            const int lo = (CAUSAL && p.window > 0) ? my_q + shift - p.window + 1 : 0;   // first visible key
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key > lim || key < lo) s[b][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][r]);
        mx = xhalf_max(mx);
        // Deferred rescale: m_run is the reference the exponentials are taken against, not necessarily the true running
        // maximum.  It moves (and O, l are rescaled, wave-uniformly) only when some row of the wave outgrew it by more
        // than 2^ATT_DEFER_LOG2, so P stays <= 2^ATT_DEFER_LOG2 (exact in the 16-bit P and the fp32 sums) and the 64
        // multiplies per tile disappear from all but the first few tiles of a row block.
        const float m_cand = fmaxf(m_run, mx);
        if (wave_any((m_cand - m_run) * c2 > ATT_DEFER_LOG2)) {     // -inf - -inf = NaN compares false: nothing seen yet
            const float alpha = fast_exp2((m_run - ((m_cand == -INFINITY) ? 0.f : m_cand)) * c2);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
            m_run = m_cand;
        }
        const float mc = ((m_run == -INFINITY) ? 0.f : m_run) * c2;
        LMI_PROF_TOUCH(o_acc[0][0]);
        LMI_PROF_MARK(4);
        // V^T fragments, one 16-key group at a time, double buffered; group 0 is requested before the exponentials
        u32x2 vlo[2][NDB], vhi[2][NDB];
        tr16_issue<NDB, 8 * G::ROWB>(v_lds, v_off, 0, vlo[0], vhi[0]);
        float psum = 0.f;
        T8 pf[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(s[b][r] * c2 - mc);
                psum += pv;
                pf[b][r >> 3][r & 7] = (T)pv;
            }
        l_run += psum;
        LMI_PROF_TOUCH(psum);
        LMI_PROF_MARK(5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
                tr16_issue<NDB, 8 * G::ROWB>(v_lds, v_off, (g + 1) * 16 * G::ROWB, vlo[(g + 1) & 1], vhi[(g + 1) & 1]);
                lgkm_fence<2 * NDB>(vlo[g & 1], vhi[g & 1]);
            } else {
                lgkm_fence<0>(vlo[g & 1], vhi[g & 1]);
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const u32x4 a = u32x4{vlo[g & 1][db][0], vlo[g & 1][db][1], vhi[g & 1][db][0], vhi[g & 1][db][1]};
                o_acc[db] = mfma32(__builtin_bit_cast(T8, a), pf[g >> 1][g & 1], o_acc[db]);
            }
        }
#ifdef LMI_ATTN_PROF
#pragma unroll
        for (int db = 0; db < NDB; ++db) LMI_PROF_TOUCH(o_acc[db][15]);
#endif
        LMI_PROF_MARK(6);
    }
    LMI_PROF_DUMP();
    for (; t < t_end; ++t) {                                       // drain: keep feeding / syncing for the waves below the diagonal
        wait_vmcnt_barrier<0>();
        if (t + 1 < t_end) issue_tile(t + 1, (t + 1 - t_begin) & 1);
    }

    // ---- finish: O / l; the two half-wave lanes of a row trade 4-element groups so each stores 16 contiguous bytes ------
    const float l_tot = xhalf_sum(l_run);
    if (p.n_splits > 1) {                                          // unnormalised partial for attn_combine_kernel
        if (my_q < len_q) {
            const long base = ((long)split * p.part_rows + (q_beg + my_q)) * p.n_heads + head;
            float* po = p.part_o + base * D;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = db * 32 + 8 * qd + 4 * fh;
                    if (d < D) *(f32x4*)(po + d) = f32x4{o_acc[db][4 * qd], o_acc[db][4 * qd + 1], o_acc[db][4 * qd + 2], o_acc[db][4 * qd + 3]};
                }
            if (fh == 0) { p.part_ml[base * 2] = m_run; p.part_ml[base * 2 + 1] = l_tot; }
        }
        return;
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (p.out_f32) {
        float* o32 = p.out_f32 + (long)(q_beg + imin(my_q, len_q - 1)) * p.ldo32 + head * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int d = db * 32 + 8 * qd + 4 * fh;
                if (my_q < len_q && d < D)
                    *(f32x4*)(o32 + d) = f32x4{o_acc[db][4 * qd] * inv, o_acc[db][4 * qd + 1] * inv, o_acc[db][4 * qd + 2] * inv, o_acc[db][4 * qd + 3] * inv};
            }
        return;
    }
    if (p.out_fp8) {
        // fp8 operand of the next GEMM: 4 e4m3 bytes per dword; after the half-wave exchange the low lane owns d = db*32 + 16qp + 0..7
        // and the high one + 8..15 — 8 contiguous bytes per lane
        const float sc = inv * p.out_fp8_scale;
        uint8_t* o8 = (uint8_t*)p.out_fp8 + (long)(q_beg + imin(my_q, len_q - 1)) * p.ldo8 + head * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned a = 0, b = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a |= (unsigned)to_fp8(o_acc[db][8 * qp + e] * sc) << (8 * e);
                    b |= (unsigned)to_fp8(o_acc[db][8 * qp + 4 + e] * sc) << (8 * e);
                }
                swap_hi_lo(a, b);
                const int d = db * 32 + 16 * qp + 8 * fh;
                if (my_q < len_q && d < D) *(u32x2*)(o8 + d) = u32x2{a, b};
            }
        return;
    }
    const long row4 = q_beg + imin(my_q, len_q - 1);
    const bool sel4 = p.out4 && (!p.row_sel || p.row_sel[row4]);
    if (p.out4 && wave_any(sel4)) {
        // (wave-uniform) residual image of the rows about to be rounded.  Block db of this head = the 16 values of this lane + the 16 of
        // its half-wave partner: block maximum by one lane exchange; this lane's codes are two dwords (quads qd = 0, 1 and 2, 3: 16 bits per
        // quad); after trading one dword the low lane owns bytes 0..7 of the block (d 0..15) and the high lane bytes 8..15 — the same trade
        // as the 16-bit rows below.
        uint8_t* o4 = p.out4 + row4 * p.ld_out4 + head * (NDB * 16);
        uint8_t* s4 = p.out4_scale + row4 * p.ld_out4s + head * NDB;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            float lo[16], amax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = db * 32 + 8 * (r >> 2) + 4 * fh + (r & 3);
                const float y = o_acc[db][r] * inv;
                lo[r] = d < D ? y - (float)(T)y : 0.f;
                amax = fmaxf(amax, __builtin_fabsf(lo[r]));
            }
            amax = xhalf_max(amax);
            float scale, sinv;
            const unsigned sb = lo4_scale_byte(amax, scale, sinv);
            const float g0[8] = {lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
            const float g1[8] = {lo[8], lo[9], lo[10], lo[11], lo[12], lo[13], lo[14], lo[15]};
            unsigned a = fp4_pack8(g0, scale, sinv), b = fp4_pack8(g1, scale, sinv);
            swap_hi_lo(a, b);                                      // low lane: b = partner's first dword; high lane: a = partner's second
            const unsigned own = fh ? b : a, other = fh ? a : b;
            const unsigned w0 = fh ? ((other & 0xffffu) | (own << 16)) : ((own & 0xffffu) | (other << 16));
            const unsigned w1 = fh ? ((other >> 16) | (own & 0xffff0000u)) : ((own >> 16) | (other & 0xffff0000u));
            if (my_q < len_q && sel4) {
                *(u32x2*)(o4 + db * 16 + fh * 8) = u32x2{w0, w1};
                if (fh == 0) s4[db] = (uint8_t)sb;
            }
        }
    }
    T* o_row = (T*)p.out + (long)(q_beg + imin(my_q, len_q - 1)) * p.ldo + head * D;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            // this lane holds d = db*32 + 8*qd + 4*fh + e for qd = 2qp (a) and 2qp+1 (b); after the exchange the low
            // half-wave lane owns d = db*32 + 16qp + 0..7 and the high one d = db*32 + 16qp + 8..15
            unsigned a[2], b[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                a[w] = pack2<T>(o_acc[db][8 * qp + 2 * w] * inv, o_acc[db][8 * qp + 2 * w + 1] * inv);
                b[w] = pack2<T>(o_acc[db][8 * qp + 4 + 2 * w] * inv, o_acc[db][8 * qp + 4 + 2 * w + 1] * inv);
                swap_hi_lo(a[w], b[w]);
            }
            const int d = db * 32 + 16 * qp + 8 * fh;
            if (my_q < len_q && d < D) *(u32x4*)(o_row + d) = u32x4{a[0], a[1], b[0], b[1]};
        }
}

// Merge of the split-KV partials: out[row, head, :] = sum_s w_s O_s / sum_s w_s l_s with w_s = 2^((m_s - max m) c2); a split that saw
// no visible key has m = -inf, weight 0 (its partial is all zeros).  One workgroup per (row, head), n_splits <= 64: wave 0 turns the
// (m, l) pairs into weights (lane s = split s), then the four waves each take the splits s = wave (mod 4), four partial rows in
// flight per wave (the loads, not the arithmetic, are this kernel's time), and the four wave sums meet in LDS in wave order — the
// result does not depend on timing.
template <typename T, int D>
__global__ void __launch_bounds__(256) attn_combine_kernel(const float* part_o, const float* part_ml, T* out, const int* cu_q,
                                                           int n_seq, int n_heads, int n_splits, int part_rows, int ldo, float scale, int lo_rows) {
    static_assert(D == 128, "attn_combine_kernel: one float2 per lane");
    __shared__ float wl[64], red[4][D + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = (int)blockIdx.x;
    const int row = item / n_heads, head = item - row * n_heads;
    if (row >= cu_q[n_seq]) return;                                // whole workgroup
    const float c2 = scale * 1.4426950408889634f;
    const long stride = (long)part_rows * n_heads;                 // (row, head) items per split
    const long base0 = (long)row * n_heads + head;
    float lw = 0.f;
    if (wave == 0) {
        float m = -INFINITY, l = 0.f;
        if (lane < n_splits) {
            m = part_ml[(base0 + lane * stride) * 2];
            l = part_ml[(base0 + lane * stride) * 2 + 1];
        }
        const float M = wave_max(m);
        const float w = (m == -INFINITY) ? 0.f : fast_exp2((m - M) * c2);
        wl[lane] = w;
        lw = wave_sum(w * l);
        if (lane == 0) red[0][D] = lw;
    }
    __syncthreads();
    f32x2 acc = {0.f, 0.f};
    for (int s0 = wave; s0 < n_splits; s0 += 16) {
        f32x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sp = imin(s0 + 4 * u, n_splits - 1);
            v[u] = *(const f32x2*)(part_o + (base0 + sp * stride) * D + lane * 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + 4 * u < n_splits) {
                const float w = wl[s0 + 4 * u];
                acc[0] += w * v[u][0];
                acc[1] += w * v[u][1];
            }
    }
    red[wave][lane * 2] = acc[0];
    red[wave][lane * 2 + 1] = acc[1];
    const float l_tot = red[0][D];
    __syncthreads();
    if (threadIdx.x < D) {
        const int d = threadIdx.x;
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const float y = sep_rn(((red[0][d] + red[1][d]) + (red[2][d] + red[3][d])) * inv);
        const T hi = (T)y;
        out[(long)row * ldo + head * D + d] = hi;
        // decode precision mode (skinny.h "hl"): the residual of the rounding as a 16-bit row of its own, lo_rows rows further down
        if (lo_rows > 0) out[(long)(row + lo_rows) * ldo + head * D + d] = (T)(y - (float)hi);
    }
}

}  // namespace lmi
