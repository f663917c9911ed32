// attention64.h — causal / full FlashAttention-2 forward, head_dim 128, with 64 query rows per wave (prefill of long sequences).
//
// Reference op: the Llama-3.1 / Mistral self-attention of the prefill (third-party LlamaAttention; in-tree analogue
// megatron_patch/model/llava/transformer.py:506-512,678-885), same arguments and results as attn_fwd_dma_kernel (attention.h).
//
// Why a second kernel.  attn_fwd_dma_kernel gives a wave 32 query rows and relies on a second workgroup per CU to fill the matrix
// pipe while the first is in its softmax; measured (profiles/r01_attn_cycle_accounting.txt) the two overlap little: 52 % matrix-pipe
// utilisation, and every wave reads the whole K and V tile from LDS for its 32 rows (75 B/clk/CU of the 128 B/clk the LDS moves).
// Here a wave owns TWO 32-row query blocks (A, B) and the overlap is built into ONE instruction stream:
//   * the work unit is a 32-key STEP (half a 64-key tile).  In steady state one region of straight-line code holds
//         MFMA:  S(i+1) = K(i+1) . Q^T  for both blocks (16)      +   O += V(i-1)^T . P(i-1)  for both blocks (16)
//         VALU:  P(i) = exp2(S(i) c - m c), row sums, 16-bit rounding               +   row maxima of S(i+1)
//     hand-interleaved one VALU group per MFMA (a gfx950 SIMD issues ~4 VALU operations under a 32x32x16 MFMA for free,
//     profiles/r01_ubench_mfma_valu_overlap.txt), so the exponentials of step i run under the matrix work of steps i+1 and i-1;
//   * K fragments (QK) and V^T fragments (PV) are read from LDS once for BOTH blocks: half the LDS bytes per FLOP;
//   * O (128 registers) and Q (64) are pinned in AGPRs — MFMA reads and writes them there directly — which leaves the
//     256 architectural VGPRs to two S buffers, two P buffers and the K / V fragments in flight;
//   * the softmax reference is the DEFERRED one of attention.h (it moves only when a row outgrew it by 2^8), and the check is
//     software-pipelined too: the maxima of S(i+1) are known at the end of region i, so the (rare) rescale happens between
//     regions: the pending P(i) — taken against the old reference — is first flushed into O, then O and l are rescaled.
//   * K / V tiles arrive by LDS-DMA into a 4-slot ring (128 KiB: one workgroup per CU), two tiles ahead, one counted wait +
//     one raw barrier per 64-key tile; every wave issues exactly 8 pieces per tile, for tiles 0 .. n_tiles + 1 — pieces of tiles
//     past the end are range-checked to zeros by the buffer resource — so the counted wait is always vmcnt(8).
// Causal structure: a workgroup is 256 query rows (4 waves x 64); a wave stops computing after its last visible tile and then
// only feeds the ring and joins the barriers.  Masks are applied to the one or two diagonal tiles, outside the pipelined region
// (a wave-uniform test per step).
#pragma once
#include <type_traits>
#include "attention.h"

namespace lmi {

// NBLK = 32-row query blocks per wave.  2: 4 waves x 64 rows, one wave per SIMD (512 registers: O and Q in AGPRs, the scores through
// inline-asm MFMAs with VGPR destinations); 1: 8 waves x 32 rows, two waves per SIMD (256 registers, every MFMA a builtin in its VGPR form:
// the second wave of a SIMD fills the issue slots the first one's MFMAs block).  Either way a workgroup is 256 query rows of one head
// and owns the CU (128 KiB ring).
template <int NBLK_> struct Attn64Geom {
    static constexpr int NBLK = NBLK_;
    static constexpr int D = 128, NW = 8 / NBLK, RPW = 32 * NBLK, BQ = NW * RPW, NKS = 8, NDB = 4, ROWB = 256, CH = 16;
    static constexpr int NT = NW * 64;
    static constexpr int TILE_BYTES = ATT_BKV * ROWB;        // one K or V tile image (16 KiB)
    static constexpr int SLOT_BYTES = 2 * TILE_BYTES;
    static constexpr int NSLOT = 4;                           // ring slots; tiles k+1 and k+2 are in flight while tile k is computed
    static constexpr int PPW = 16 / NW;                       // K pieces per wave per tile (and as many V pieces): 16 x 1 KiB per image
    static constexpr int SMEM = NSLOT * SLOT_BYTES;
    static_assert(BQ == 256 && (PPW == 2 || PPW == 4), "workgroup = 256 query rows");
};

// S^T += K . Q^T with the destination in ARCHITECTURAL VGPRs and the Q fragment read from an AGPR.  With one wave per SIMD (512
// registers) hipcc selects the AGPR-destination form for every MFMA builtin of a function (the choice is per function), which would put
// the scores in AGPRs and cost a v_accvgpr_read per score before the exponentials; the scores therefore go through inline asm.
// Hazard contract of the callers (the compiler cannot see inside the asm): a VALU instruction reads these destinations no earlier than
// 11 wait states after the MFMA (8-pass XDL write -> VALU read) — in the steady-state region the maxima follow 16 PV MFMAs later; the
// two places that read them right away insert mfma_result_nops() first.  The zero-accumulator form marks its destination early-clobber:
// a 16-register MFMA destination must not overlap the A / B operands.
// pin_*: empty volatile asm that "uses and redefines" a value — zero instructions, but hipcc can neither sink the computation of the value past
// this point (out of the pipelined region, towards its next use) nor reassociate across it; the interleaving written below is the one issued.
#ifndef LMI_EMU
template <typename V> LMI_DEV void pin_agpr(V& v) { asm volatile("" : "+a"(v)); }
template <typename V> LMI_DEV void pin_vgpr(V& v) { asm volatile("" : "+v"(v)); }
LMI_DEV void mfma_result_nops() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }
LMI_DEV void mfma32_sv(f16x8 k, f16x8 q, f32x16& s) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
LMI_DEV void mfma32_sv(bf16x8 k, bf16x8 q, f32x16& s) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
LMI_DEV void mfma32_sv0(f16x8 k, f16x8 q, f32x16& s) { asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
LMI_DEV void mfma32_sv0(bf16x8 k, bf16x8 q, f32x16& s) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
#else
template <typename V> inline void pin_agpr(V&) {}
template <typename V> inline void pin_vgpr(V&) {}
inline void mfma_result_nops() {}
template <typename V8> inline void mfma32_sv(V8 k, V8 q, f32x16& s) { s = mfma32(k, q, s); }
template <typename V8> inline void mfma32_sv0(V8 k, V8 q, f32x16& s) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    s = mfma32(k, q, z);
}
#endif

template <typename T, bool CAUSAL, int NBLK>
__global__ void __launch_bounds__(Attn64Geom<NBLK>::NT, 3 - NBLK) attn_fwd_r64_kernel(AttnArgs p) {
    typedef Attn64Geom<NBLK> G;
    typedef typename vec_of<T>::x8 T8;
    constexpr int NKS = G::NKS, NDB = G::NDB, NW = G::NW, PPW = G::PPW, ROWB = G::ROWB;
    constexpr bool ASM_QK = (NBLK == 2);                           // scores through the inline-asm MFMA (see mfma32_sv); O and Q pinned in AGPRs
    // where the diagonal / tail masks go.  NBLK == 1: after the region that produced the scores, followed by the masked maxima (exact: a
    // row's reference never exceeds its visible maximum).  NBLK == 2: before the region that exponentiates them, the in-region maxima being
    // taken over unmasked scores (a reference somewhat above the visible maximum is valid, only less tight): with 250 live registers any
    // extra code between the asm MFMAs and the next region makes hipcc spill scores right behind an MFMA it does not know to be one.
    constexpr bool MASK_AFTER = (NBLK == 1);
    LMI_DYN_SMEM(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int fr = lane & 31, fh = lane >> 5;
    // 1-D grid, heads fastest (heavy late causal blocks of all heads first; workgroup b runs on XCD b % 8 = one kv head per XCD)
    const int bid = (int)blockIdx.x;
    const int h_idx = bid % p.n_heads;
    const int rest = bid / p.n_heads;
    const int qb = p.n_qblocks - 1 - rest % p.n_qblocks, seq = rest / p.n_qblocks;
    const int kvh = h_idx % p.n_kv_heads;
    const int head = kvh * (p.n_heads / p.n_kv_heads) + h_idx / p.n_kv_heads;
    const int q_beg = p.cu_q[seq], len_q = p.cu_q[seq + 1] - q_beg;
    const int k_beg = p.cu_k[seq], len_k = p.cu_k[seq + 1] - k_beg;
    const int q0 = qb * G::BQ;
    if (q0 >= len_q) return;
    const int shift = len_k - len_q;                               // causal: key j visible to query i iff j <= i + shift
    const int kv_end = CAUSAL ? imin(len_k, q0 + G::BQ + shift) : len_k;
    const int n_tiles = kv_end > 0 ? (kv_end + ATT_BKV - 1) / ATT_BKV : 0;
    const int wq0 = q0 + wave * G::RPW;
    int my_tiles = n_tiles;                                        // tiles this wave computes; later ones are fully masked for its 64 rows
    if (CAUSAL) my_tiles = (wq0 + G::RPW - 1 + shift < 0) ? 0 : imin(n_tiles, (wq0 + G::RPW - 1 + shift) / ATT_BKV + 1);
    if (wq0 >= len_q) my_tiles = 0;

    // ---- Q fragments (both blocks), pinned in AGPRs ----------------------------------------------------------------------
    int my_q[NBLK];
    T8 qf[NBLK][NKS];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        my_q[blk] = wq0 + blk * 32 + fr;
        const T* q_row = (const T*)p.q + (long)(q_beg + imin(my_q[blk], len_q - 1)) * p.ldq + head * G::D;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[blk][ks] = *(const T8*)(q_row + (2 * ks + fh) * 8);
    }
    if constexpr (ASM_QK) {
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) pin_agpr(qf[blk][ks]);
    }

    // ---- LDS-DMA sources (as attention.h: K chunk c of row r at c ^ (r & 15), V chunk c at c ^ ((r & 3) << 2)) --------------
    const T* k_base = (const T*)p.k + (long)k_beg * p.ldk + kvh * G::D;
    const T* v_base = (const T*)p.v + (long)k_beg * p.ldv + kvh * G::D;
    const BufRsrc k_buf = make_buf(k_base, len_k > 0 ? (unsigned)(((long)(len_k - 1) * p.ldk + G::D) * 2) : 0u);
    const BufRsrc v_buf = make_buf(v_base, len_k > 0 ? (unsigned)(((long)(len_k - 1) * p.ldv + G::D) * 2) : 0u);
    unsigned p_ko[PPW], p_vo[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int ci = (wave + NW * i) * 64 + lane;
        const int r = ci / G::CH, c = ci - r * G::CH;
        p_ko[i] = (unsigned)(r * p.ldk + ((c ^ (r & 15)) << 3)) * 2u;
        p_vo[i] = (unsigned)(r * p.ldv + ((c ^ ((r & 3) << 2)) << 3)) * 2u;
    }
    // piece j of this wave for tile t: j < PPW are its K pieces, the rest its V pieces (tiles past the end read zeros)
    auto issue_piece = [&](int j, int t) {
        const int i = j < PPW ? j : j - PPW;
        char* dst = smem + (t & (G::NSLOT - 1)) * G::SLOT_BYTES + (j < PPW ? 0 : G::TILE_BYTES) + (wave + NW * i) * 1024;
        if (j < PPW) glds16_buf(k_buf, p_ko[i], (unsigned)t * (unsigned)(ATT_BKV * 2) * (unsigned)p.ldk, dst);
        else glds16_buf(v_buf, p_vo[i], (unsigned)t * (unsigned)(ATT_BKV * 2) * (unsigned)p.ldv, dst);
    };

    // fragment read offsets (lane dependent, loop invariant)
    int k_off[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int c = 2 * ks + fh;
        k_off[ks] = fr * ROWB + ((c ^ (fr & 15)) << 4);
    }
    const int tr_j = (lane & 15) >> 2, tr_g = lane & 3, tr_half = (lane >> 4) & 1;
    int v_off[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) v_off[db] = (4 * fh + tr_j) * ROWB + ((db ^ tr_j) << 6) + tr_half * 32 + tr_g * 8;

    f32x16 o_acc[NBLK][NDB];
    float m_run[NBLK], l_run[NBLK];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        m_run[blk] = -INFINITY;
        l_run[blk] = 0.f;
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[blk][i][r] = 0.f;
            if constexpr (ASM_QK) pin_agpr(o_acc[blk][i]);
        }
    }
    const float c2 = p.scale * 1.4426950408889634f;

    // ---- building blocks -----------------------------------------------------------------------------------------------------
    auto need_mask = [&](int t, int b) {                           // wave-uniform
        const int kv0 = t * ATT_BKV + b * 32;
        return (kv0 + 32 > len_k) || (CAUSAL && (kv0 + 31 > wq0 + shift)) || (CAUSAL && p.window > 0 && kv0 <= wq0 + G::RPW - 1 + shift - p.window);
    };
    auto apply_mask = [&](f32x16 (&s)[NBLK], int t, int b) {
        const int kv0 = t * ATT_BKV + b * 32;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const int lim = CAUSAL ? imin(len_k - 1, my_q[blk] + shift) : len_k - 1;
            const int lo = (CAUSAL && p.window > 0) ? my_q[blk] + shift - p.window + 1 : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (key > lim || key < lo) s[blk][r] = -INFINITY;
            }
        }
    };
    // row maxima of a step's scores against the running reference: returns true (wave-uniform) when some row outgrew it by 2^8
    float m_cand[NBLK];
    auto max_check = [&](const f32x16 (&s)[NBLK]) {
        bool need = false;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            float mx = fmaxf(fmaxf(s[blk][0], s[blk][1]), s[blk][2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) mx = fmaxf(fmaxf(mx, s[blk][r]), s[blk][r + 1]);
            mx = fmaxf(mx, s[blk][15]);
            mx = xhalf_max(mx);
            m_cand[blk] = fmaxf(m_run[blk], mx);
            need = need || ((m_cand[blk] - m_run[blk]) * c2 > ATT_DEFER_LOG2);     // -inf - -inf = NaN compares false
        }
        return wave_any(need);
    };
    // O^T += V^T(step) . P^T for both blocks, not overlapped with anything (the flush of a rescale, and the tail)
    auto pv_plain = [&](const char* v_lds, int b, const T8 (&pf)[NBLK][2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x2 vlo[NDB], vhi[NDB];
            tr16_issue<NDB, 8 * ROWB>(v_lds, v_off, (2 * b + u) * 16 * ROWB, vlo, vhi);
            lgkm_fence<0>(vlo, vhi);
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const u32x4 a = u32x4{vlo[db][0], vlo[db][1], vhi[db][0], vhi[db][1]};
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) o_acc[blk][db] = mfma32(__builtin_bit_cast(T8, a), pf[blk][u], o_acc[blk][db]);
            }
        }
    };
    // (rare) move the softmax reference of both blocks to m_cand: one 16-register O block at a time — the accumulators live in AGPRs and
    // pass through VGPRs for the multiply; fenced so that the transient stays 16 registers, not 128
    auto rescale = [&]() {
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const float alpha = fast_exp2((m_run[blk] - ((m_cand[blk] == -INFINITY) ? 0.f : m_cand[blk])) * c2);
            l_run[blk] *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                sched_fence();
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[blk][i][r] *= alpha;
                if constexpr (ASM_QK) pin_agpr(o_acc[blk][i]);
            }
            m_run[blk] = m_cand[blk];
        }
        sched_fence();
    };
    auto zero_p = [&](T8 (&pf)[NBLK][2]) {
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[blk][u][e] = (T)0.0f;
    };

    LMI_PROF_DECL();
    // ---- the region: QK of the NEXT step, PV of the PREVIOUS step, exponentials of the CURRENT step ----------------------------------
    //   s_cur / p_cur : scores of the current step (in) -> probabilities (out)
    //   HAS_QK: s_nxt = K(k_next, b_next) . Q^T and its row maxima -> returns the wave-uniform "reference must move" (see MASK_AFTER for
    //           the diagonal / tail steps); this wave's LDS-DMA pieces [DMA_J0, DMA_J0 + PPW) of tile dma_tile ride between the MFMAs
    //   HAS_PV: O += V(v_prev, b_prev)^T . p_prv
    // Written as 16 NBLK slots of {one MFMA, one slice of VALU work}, each closed by a scheduling fence so that the order below is the order
    // issued.  QK half: MFMA (ks, block), every second slot the exponentials of one score pair; PV half: MFMA (16-key group, d-block,
    // block), alternating an exponential pair and four scores of s_nxt into the row maxima.
    auto region = [&](auto has_qk, auto has_pv, auto dma_j0, f32x16 (&s_cur)[NBLK], T8 (&p_cur)[NBLK][2], f32x16 (&s_nxt)[NBLK],
                      const T8 (&p_prv)[NBLK][2], const char* k_next, int b_next, const char* v_prev, int b_prev, int dma_tile) -> bool {
        constexpr bool HAS_QK = decltype(has_qk)::value, HAS_PV = decltype(has_pv)::value;
        constexpr int DMA_J0 = decltype(dma_j0)::value;
        f32x2 mc2[NBLK], psum2[NBLK];
        float mx[NBLK];
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const float mc = ((m_run[blk] == -INFINITY) ? 0.f : m_run[blk]) * c2;
            mc2[blk] = f32x2{mc, mc};
            psum2[blk] = f32x2{0.f, 0.f};
            mx[blk] = -INFINITY;
        }
        const f32x2 c22 = f32x2{c2, c2};
        auto exp_slice = [&](int k) {                                // k = 0 .. 8 NBLK - 1: score pair k / NBLK of block k % NBLK
            const int blk = k % NBLK, r = (k / NBLK) * 2;
            const f32x2 tv = f32x2{s_cur[blk][r], s_cur[blk][r + 1]} * c22 - mc2[blk];
            f32x2 ev = f32x2{fast_exp2(tv[0]), fast_exp2(tv[1])};
            pin_vgpr(ev);
            psum2[blk] += ev;
            pin_vgpr(psum2[blk]);
            p_cur[blk][r >> 3][r & 7] = (T)ev[0];
            p_cur[blk][r >> 3][(r & 7) + 1] = (T)ev[1];
            if ((r & 7) == 6) pin_vgpr(p_cur[blk][r >> 3]);       // a complete 8-key operand
        };
        auto max_slice = [&](int k) {                                // k = 0 .. 4 NBLK - 1: scores 4 (k / NBLK) .. + 3 of block k % NBLK of s_nxt
            const int blk = k % NBLK, r = (k / NBLK) * 4;
            mx[blk] = fmaxf(fmaxf(mx[blk], s_nxt[blk][r]), s_nxt[blk][r + 1]);
            mx[blk] = fmaxf(fmaxf(mx[blk], s_nxt[blk][r + 2]), s_nxt[blk][r + 3]);
            pin_vgpr(mx[blk]);
        };
        sched_fence();
        if (HAS_QK) {
            constexpr int AHEAD = 3;
            T8 kf[NKS];
            const char* kb = k_next + b_next * 32 * ROWB;
#pragma unroll
            for (int ks = 0; ks < AHEAD; ++ks) kf[ks] = *(const T8*)(kb + k_off[ks]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sched_fence();
                if (ks + AHEAD < NKS) kf[ks + AHEAD] = *(const T8*)(kb + k_off[ks + AHEAD]);
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    if constexpr (ASM_QK) {
                        if (ks == 0) mfma32_sv0(kf[ks], qf[blk][ks], s_nxt[blk]); else mfma32_sv(kf[ks], qf[blk][ks], s_nxt[blk]);
                    } else {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        s_nxt[blk] = mfma32(kf[ks], qf[blk][ks], ks == 0 ? z : s_nxt[blk]);
                    }
                    const int j = ks * NBLK + blk;                  // slot index in the half
                    if ((j & 1) == 0) exp_slice(j >> 1);
                    if (blk + 1 < NBLK) sched_fence();
                }
                // PPW pieces per region: after every second k-step (4 pieces) or every fourth (2 pieces)
                if (PPW == 4 ? (ks & 1) : ((ks & 3) == 3)) issue_piece(DMA_J0 + (PPW == 4 ? (ks >> 1) : (ks >> 2)), dma_tile);
            }
            sched_fence();
            LMI_PROF_MARK(2);
            if constexpr (ASM_QK && !HAS_PV) mfma_result_nops();              // the scores are read right away (maxima without PV MFMAs in between)
        } else {
#pragma unroll
            for (int k = 0; k < 4 * NBLK; ++k) exp_slice(k);
        }
        if (HAS_PV) {
            u32x2 vlo[2][NDB], vhi[2][NDB];
            tr16_issue<NDB, 8 * ROWB>(v_prev, v_off, (2 * b_prev) * 16 * ROWB, vlo[0], vhi[0]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 0) {
                    tr16_issue<NDB, 8 * ROWB>(v_prev, v_off, (2 * b_prev + 1) * 16 * ROWB, vlo[1], vhi[1]);
                    lgkm_fence<2 * NDB>(vlo[0], vhi[0]);
                } else {
                    lgkm_fence<0>(vlo[1], vhi[1]);
                }
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const u32x4 a = u32x4{vlo[u][db][0], vlo[u][db][1], vhi[u][db][0], vhi[u][db][1]};
#pragma unroll
                    for (int blk = 0; blk < NBLK; ++blk) {
                        sched_fence();
                        o_acc[blk][db] = mfma32(__builtin_bit_cast(T8, a), p_prv[blk][u], o_acc[blk][db]);
                        const int j = (u * NDB + db) * NBLK + blk;
                        if ((j & 1) == 0) exp_slice(4 * NBLK + (j >> 1));
                        else if (HAS_QK) max_slice(j >> 1);
                    }
                }
            }
            sched_fence();
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    if constexpr (ASM_QK) pin_agpr(o_acc[blk][db]); else pin_vgpr(o_acc[blk][db]);
                }
        } else {
#pragma unroll
            for (int k = 4 * NBLK; k < 8 * NBLK; ++k) exp_slice(k);
            if (HAS_QK) {
#pragma unroll
                for (int k = 0; k < 4 * NBLK; ++k) max_slice(k);
            }
        }
        LMI_PROF_MARK(3);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) l_run[blk] += psum2[blk][0] + psum2[blk][1];
        bool need = false;
        if (HAS_QK) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
                m_cand[blk] = fmaxf(m_run[blk], xhalf_max(mx[blk]));
                need = need || ((m_cand[blk] - m_run[blk]) * c2 > ATT_DEFER_LOG2);      // -inf - -inf = NaN compares false
            }
            need = wave_any(need);
        }
        LMI_PROF_MARK(4);
        return need;
    };
    typedef std::true_type Y;
    typedef std::false_type N;
    typedef std::integral_constant<int, 0> J0;
    typedef std::integral_constant<int, PPW> JV;

    // ---- ring prologue: tiles 0 and 1 requested, B(0) ---------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2 * PPW; ++j) issue_piece(j, 0);
#pragma unroll
    for (int j = 0; j < 2 * PPW; ++j) issue_piece(j, 1);
    wait_vmcnt_barrier<2 * PPW>();                                 // B(0): tile 0 landed (tile 1 in flight)
    auto slot_k = [&](int t) { return (const char*)smem + (t & (G::NSLOT - 1)) * G::SLOT_BYTES; };
    auto slot_v = [&](int t) { return (const char*)smem + (t & (G::NSLOT - 1)) * G::SLOT_BYTES + G::TILE_BYTES; };

    if (my_tiles > 0) {
        f32x16 s0[NBLK], s1[NBLK];
        T8 p0[NBLK][2], p1[NBLK][2];
        zero_p(p0); zero_p(p1);
        // after a region whose maxima say the reference must move: the probabilities just produced were taken against the OLD reference —
        // flush them into O, then rescale O and l, and leave zeros for the PV slot of the next region
        auto settle = [&](bool need, T8 (&p_new)[NBLK][2], int t, int b) {
            if (need) {
                pv_plain(slot_v(t), b, p_new);
                zero_p(p_new);
                rescale();
            }
        };
        // step (0, 0): plain QK, no overlap (K pieces of tile 2 ride along)
        {
            constexpr int AHEAD = 3;
            T8 kf[NKS];
            const char* kb = slot_k(0);
#pragma unroll
            for (int ks = 0; ks < AHEAD; ++ks) kf[ks] = *(const T8*)(kb + k_off[ks]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                sched_fence();
                if (ks + AHEAD < NKS) kf[ks + AHEAD] = *(const T8*)(kb + k_off[ks + AHEAD]);
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    if constexpr (ASM_QK) {
                        if (ks == 0) mfma32_sv0(kf[ks], qf[blk][ks], s0[blk]); else mfma32_sv(kf[ks], qf[blk][ks], s0[blk]);
                    } else {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        s0[blk] = mfma32(kf[ks], qf[blk][ks], ks == 0 ? z : s0[blk]);
                    }
                }
                if (PPW == 4 ? (ks & 1) : ((ks & 3) == 3)) issue_piece(PPW == 4 ? (ks >> 1) : (ks >> 2), 2);
            }
            sched_fence();
            if constexpr (ASM_QK) mfma_result_nops();
            if (MASK_AFTER) {
                if (need_mask(0, 0)) apply_mask(s0, 0, 0);
                if (max_check(s0)) rescale();                      // from -inf: alpha = 0 on zero accumulators
            } else {
                if (max_check(s0)) rescale();                      // (maxima of the unmasked scores)
                if (need_mask(0, 0)) apply_mask(s0, 0, 0);
            }
        }
        // first region: QK(0, 1), no PV yet (V pieces of tile 2 ride along)
        {
            bool need = region(Y{}, N{}, JV{}, s0, p0, s1, p1, slot_k(0), 1, nullptr, 0, 2);
            if (MASK_AFTER && need_mask(0, 1)) { apply_mask(s1, 0, 1); need = max_check(s1); }
            settle(need, p0, 0, 0);
        }
        for (int t = 0; t + 1 < my_tiles; ++t) {
            // step (t, 1) current; next = (t + 1, 0) behind B(t + 1); previous = (t, 0)
            LMI_PROF_MARK(0);
            wait_vmcnt_barrier<2 * PPW>();                         // B(t + 1): tile t + 1 landed, slot of tile t - 1 free for tile t + 3
            {
                LMI_PROF_MARK(1);
                if (!MASK_AFTER && need_mask(t, 1)) apply_mask(s1, t, 1);
                bool need = region(Y{}, Y{}, J0{}, s1, p1, s0, p0, slot_k(t + 1), 0, slot_v(t), 0, t + 3);
                if (MASK_AFTER && need_mask(t + 1, 0)) { apply_mask(s0, t + 1, 0); need = max_check(s0); }
                LMI_PROF_MARK(5);
                settle(need, p1, t, 1);
                LMI_PROF_MARK(6);
            }
            // step (t + 1, 0) current; next = (t + 1, 1); previous = (t, 1)
            {
                if (!MASK_AFTER && need_mask(t + 1, 0)) apply_mask(s0, t + 1, 0);
                bool need = region(Y{}, Y{}, JV{}, s0, p0, s1, p1, slot_k(t + 1), 1, slot_v(t), 1, t + 3);
                if (MASK_AFTER && need_mask(t + 1, 1)) { apply_mask(s1, t + 1, 1); need = max_check(s1); }
                LMI_PROF_MARK(5);
                settle(need, p0, t + 1, 0);
                LMI_PROF_MARK(6);
            }
        }
        // last step (my_tiles - 1, 1): no next step; PV of (my_tiles - 1, 0) under its exponentials, then its own PV bare
        if (!MASK_AFTER && need_mask(my_tiles - 1, 1)) apply_mask(s1, my_tiles - 1, 1);
        region(N{}, Y{}, J0{}, s1, p1, s0, p0, nullptr, 0, slot_v(my_tiles - 1), 0, 0);
        pv_plain(slot_v(my_tiles - 1), 1, p1);
    } else {
#pragma unroll
        for (int j = 0; j < 2 * PPW; ++j) issue_piece(j, 2);       // B(0) passed: this wave's share of tile 2
    }
    // drain: keep feeding the ring and joining the barriers for the waves still below the diagonal
    for (int k = imax(my_tiles, 1); k < n_tiles; ++k) {
        wait_vmcnt_barrier<2 * PPW>();                             // B(k)
#pragma unroll
        for (int j = 0; j < 2 * PPW; ++j) issue_piece(j, k + 2);
    }
    LMI_PROF_DUMP();
    wait_vmcnt<0>();                                               // no LDS-DMA may outlive the workgroup's LDS allocation

    // ---- finish: O / l; the two half-wave lanes of a row trade 4-element groups so each stores 16 contiguous bytes ------
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const float l_tot = xhalf_sum(l_run[blk]);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        T* o_row = (T*)p.out + (long)(q_beg + imin(my_q[blk], len_q - 1)) * p.ldo + head * G::D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned a[2], b[2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    a[w] = pack2<T>(o_acc[blk][db][8 * qp + 2 * w] * inv, o_acc[blk][db][8 * qp + 2 * w + 1] * inv);
                    b[w] = pack2<T>(o_acc[blk][db][8 * qp + 4 + 2 * w] * inv, o_acc[blk][db][8 * qp + 4 + 2 * w + 1] * inv);
                    swap_hi_lo(a[w], b[w]);
                }
                const int d = db * 32 + 16 * qp + 8 * fh;
                if (my_q[blk] < len_q) *(u32x4*)(o_row + d) = u32x4{a[0], a[1], b[0], b[1]};
            }
    }
}

}  // namespace lmi
