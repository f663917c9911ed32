// skinny.h — out[M, N] = epilogue( X[M, K] . W[N, K]^T ) for M <= 16: the projections of a BATCHED decode step
// (several samples per GPU; reference loop EVAL:448-454 is batch 1).
//
// A decode step is bound by the weight stream (16 GB of 16-bit weights per token for Llama-3.1-8B); with B sequences in flight the
// same stream serves B tokens.  One workgroup = 16 weight rows (32 for SwiGLU: 16 gate + their 16 up rows), 8 waves that split K in
// 128-element steps (wave w takes steps w, w + 8, ...).  Per step a lane loads 64 contiguous bytes of "its" weight row and 64 bytes
// of "its" batch row — lane (i, g) = (l & 15, l >> 4): row i, k-range [32 c + 8 g, + 8) of 32-k block c = 0..3 — which is exactly the
// v_mfma_f32_16x16x32 operand layout, so the fragments go from global memory / L2 straight into the matrix pipe: no LDS staging, no
// shuffles, 4 MFMAs per step, D[n][m] in 4 accumulator registers.  Lanes of batch rows >= M load nothing (zeros).  The 8 partial
// tiles meet in LDS, are summed in wave order (deterministic) and the epilogue runs on 256 threads.
// Why the batch rows are not staged in LDS: X is M x K x 2 bytes (224 KiB at K = 14336, M = 8) per workgroup either way; from L2 it
// costs (M / 16) of the weight stream's VMEM issue slots and nothing else.
// Weight layout.  In the nn.Linear layout a wave instruction gathers 64 separate 16-byte pieces (adjacent lanes = adjacent ROWS):
// the address path takes one lane per clock, 64 cycles per KiB, and the kernel is bound by VMEM issue (~4 TB/s) instead of HBM.
// LAYOUT 1 (packed) reads weights pre-arranged in the MFMA operand order (leopard_amd.weights.skinny_pack, done once at load):
// block (16-row group r, k-step s, 32-k chunk c) is 1 KiB with lane l's 16 bytes at l * 16 — one fully coalesced 1-KiB request
// per instruction.  LAYOUT 2 reads the nn.Linear layout with a COALESCING lane order — lane l takes piece l & 3 of row l >> 2, so four
// adjacent lanes cover 64 contiguous bytes of a row and a wave instruction is 16 segments of 64 bytes instead of 64 separate 16-byte
// pieces — and then moves every dword to the lane the MFMA wants it in with one ds_bpermute_b32 (lane 16 g + i <- lane 4 i + g): no second
// copy of the weights.  LAYOUT 0 = the nn.Linear layout read in the MFMA lane order directly (the slow reference form).  All three give
// the same bits.
// Requirements: K % 128 == 0, N % 16 == 0 (SwiGLU: N % 64 == 0, rows interleaved [32 gate | 32 up] as weights.py lays gate/up out),
// 16-byte aligned rows.
#pragma once
#include "lmi_device.h"

namespace lmi {

enum { SK_STORE_T = 0, SK_RESID_F32 = 1, SK_SWIGLU_T = 2, SK_STORE_F32 = 3, SK_QKV_ROPE_T = 4 };

// SK_QKV_ROPE_T: the q | k | v projection of a batched decode step with RoPE and the KV append in the epilogue.  W rows in
// weights.rope_permute_rows order (q / k heads stored as d = [0..31, 64..95, 32..63, 96..127]), so that the two 16-row blocks a workgroup
// owns — rows r and r + 32 of a 64-row group, exactly the SwiGLU pairing — hold first-half elements and their rotate-half partners.
// Row m of the batch rotates at its own position pos[m] (device memory) and appends K / V to row m * cache_stride + pos[m] of the pooled
// caches; v rows (natural order) are two independent blocks.

// The RMSNorms of the batched decode step folded into its projections, as lmi_gemm_ex does for the prefill (gemm.h GemmRowScale):
//   producer (SK_RESID_F32, norm_out != null): after x += acc it also writes norm_out[m, n] = T(x[m, n] * gamma[n]) — the next norm's
//       gain applied, its row scale still missing — and rowsq_out[m, unit] = the sum of x[m, n]^2 over the unit's 16 columns (one
//       partial per workgroup and row, written once: no atomics, bit-reproducible);
//   consumer (any epilogue, rowsq_in != null): accumulator row m is multiplied by rstd[m] = rsqrt(sum_j rowsq_in[m, j] / norm_dim + eps)
//       before RoPE / SwiGLU / store; the partials are summed in a fixed order (32 lanes per row: strided partial sums, then a butterfly).
// hl (round 6: the decode step's precision mode, engine.precision = "lo4" / "split"): X holds 2 M rows — rows [0, M) = T(x), rows [M, 2 M) =
// T(x - T(x)), the 16-bit image of the hand-over rounding's residual (2 M <= 16) — and both products land in the same sums, so the
// projection sees the operand to ~22 bits at NO extra weight traffic (the kernel is bound by the weight stream; the lo rows are two of
// its <= 16 batch rows).  Every 16-bit output that is itself a projection operand (SwiGLU / STORE results, the producer's T(x gamma)) is
// written as such a pair again: row m and row m + M.
struct SkinnyNorm {
    int hl;                   // 0 = plain; 1 = X is [hi rows | lo rows] (2 M rows) and 16-bit operand outputs are written as pairs
    const float* rowsq_in;    // [M, parts_in] or null
    int parts_in;
    float inv_dim, eps;
    void* norm_out;           // T [M, ld_norm] or null (SK_RESID_F32 only)
    int ld_norm;
    const float* gamma;       // [N]
    float* rowsq_out;         // [M, N / 16]
};

template <typename T, int EPI, int LAYOUT>
__global__ void __launch_bounds__(512) skinny_gemm_kernel(const T* W, const T* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, RopeEpi rp,
                                                          SkinnyNorm nm) {
    typedef typename vec_of<T>::x8 T8;
    constexpr bool PACKED = (LAYOUT == 1), COAL = (LAYOUT == 2);
    constexpr bool PAIR = (EPI == SK_SWIGLU_T || EPI == SK_QKV_ROPE_T);
    constexpr int NW = PAIR ? 2 : 1;                               // weight row blocks per workgroup (gate, up / first half, rotate-half partner)
    constexpr int DEPTH = PAIR ? 2 : 3;                            // k-steps of loads in flight per wave (<= 128 VGPRs: two workgroups per CU)
    __shared__ float part[8][NW][64][4];
    __shared__ float rstd_s[16], sq_s[16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int i = lane & 15, g = lane >> 4;
    const int unit = blockIdx.x;
    // consumer: this thread's share of the row-sum-of-squares partials (row tid >> 5, partials (tid & 31), + 32, ...): requested first,
    // reduced after the main loop
    float ssq = 0.f;
    if (nm.rowsq_in && (tid >> 5) < M)
        for (int q = tid & 31; q < nm.parts_in; q += 32) ssq += nm.rowsq_in[(long)(tid >> 5) * nm.parts_in + q];
    // first weight row of block b: plain = 16 rows per unit; SwiGLU = unit u -> 64-row group u >> 1, half u & 1: gate rows at +16 * half,
    // their up partners 32 rows further
    const int row0 = PAIR ? (unit >> 1) * 64 + (unit & 1) * 16 : unit * 16;
    const int nsteps_all = K >> 7;
    const T* wrow[NW];
#pragma unroll
    for (int b = 0; b < NW; ++b)
        wrow[b] = PACKED ? W + (long)((row0 + 32 * b) >> 4) * nsteps_all * 2048 + lane * 8      // 16-row group x k-steps x 4 KiB
                  : COAL ? W + (long)(row0 + 32 * b + (lane >> 2)) * ldw + 8 * (lane & 3)        // row l >> 2, 16-byte piece l & 3
                         : W + (long)(row0 + 32 * b + i) * ldw + 8 * g;
    const int src_lane = 4 * i + g;                                // COAL: MFMA lane (i, g) = 16 g + i takes what lane 4 i + g loaded
    const T* xrow = X + (long)i * ldx + 8 * g;
    const int MX = nm.hl ? 2 * M : M;                              // batch rows in X
    const bool has_x = i < MX;
    const int nsteps = K >> 7;
    const int my_steps = (nsteps - wave + 7) >> 3;                 // steps wave, wave + 8, ...
    T8 wv[DEPTH][NW][4], xv[DEPTH][4];
    auto load = [&](int slot, int s) {
        const int k0 = (wave + 8 * s) * 128;
#pragma unroll
        for (int b = 0; b < NW; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                wv[slot][b][c] = ld_stream(PACKED ? (const T8*)(wrow[b] + (long)(wave + 8 * s) * 2048 + c * 512) : (const T8*)(wrow[b] + k0 + 32 * c));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (has_x) xv[slot][c] = *(const T8*)(xrow + k0 + 32 * c);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[slot][c][e] = (T)0.0f;
            }
        }
    };
    f32x4 acc[NW];
#pragma unroll
    for (int b = 0; b < NW; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < my_steps) load(d, d);
    for (int base = 0; base < my_steps; base += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = base + d;
            if (s < my_steps) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int b = 0; b < NW; ++b) {
                        T8 wf = wv[d][b][c];
                        if (COAL) {                                 // lane transpose of the 16 x 4 piece grid, one dword at a time
                            u32x4 raw = __builtin_bit_cast(u32x4, wf);
#pragma unroll
                            for (int e = 0; e < 4; ++e) raw[e] = (uint32_t)shfl_idx((int)raw[e], src_lane);
                            wf = __builtin_bit_cast(T8, raw);
                        }
                        acc[b] = mfma16(wf, xv[d][c], acc[b]);
                    }
                if (s + DEPTH < my_steps) load(d, s + DEPTH);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NW; ++b) *(f32x4*)part[wave][b][lane] = acc[b];
    if (nm.rowsq_in) {
#pragma unroll
        for (int msk = 16; msk >= 1; msk >>= 1) ssq += shfl_xor(ssq, msk);      // the 32 lanes of one row (a half wave)
        if ((tid & 31) == 0) rstd_s[tid >> 5] = 1.0f / sqrtf(ssq * nm.inv_dim + nm.eps);
    }
    __syncthreads();
    // D[n][m]: lane l, register r -> n = 4 (l >> 4) + r, m = l & 15.  Thread t < 256 finishes element (l = t & 63, r = t >> 6).
    if (tid < 256) {
        const int l = tid & 63, r = tid >> 6;
        const int n = 4 * (l >> 4) + r, m = l & 15;
        float v[NW];
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            float s = part[0][b][l][r];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += part[w][b][l][r];     // fixed order: results do not depend on timing
            if (nm.hl && m < M) {                                   // + the lo row's product (lane l + M holds (n, m + M)): hi sum, lo sum, then their sum
                float s2 = part[0][b][l + M][r];
#pragma unroll
                for (int w = 1; w < 8; ++w) s2 += part[w][b][l + M][r];
                s += s2;
            }
            v[b] = nm.rowsq_in ? s * rstd_s[m] : s;
        }
        if (m < M) {
            if (EPI == SK_QKV_ROPE_T) {
                const int c1 = row0 + n, c2 = c1 + 32;              // the two columns (in W's row order) this thread finishes
                T* orow = (T*)out + (long)m * ldo;
                const int p = rp.pos[m];
                const long crow = (long)m * rp.cache_stride + p;
                if (c1 >= rp.rope_q + rp.rope_k) {                   // v: natural order, two plain values
                    const int vc = rp.rope_q + rp.rope_k;
                    orow[c1] = (T)v[0];
                    orow[c2] = (T)v[1];
                    T* vrow = (T*)rp.v_cache + crow * rp.ld_cache - vc;
                    vrow[c1] = (T)v[0];
                    vrow[c2] = (T)v[1];
                } else {                                             // q / k: rotate-half on the fp32 sums, natural order on store
                    const int hbase = c1 & ~127, j = c1 & 127;       // j in [0, 32) or [64, 96): position in the permuted head
                    const int d1 = (j >> 6) * 32 + (j & 31);         // first-half element 0..63; partner d1 + 64
                    const float cs = rp.cos_all[(long)p * 64 + d1], sn = rp.sin_all[(long)p * 64 + d1];
                    const T o1 = (T)__builtin_fmaf(v[0], cs, -mul_rn(v[1], sn));
                    const T o2 = (T)__builtin_fmaf(v[1], cs, mul_rn(v[0], sn));
                    orow[hbase + d1] = o1;
                    orow[hbase + d1 + 64] = o2;
                    if (c1 >= rp.rope_q) {
                        T* krow = (T*)rp.k_cache + crow * rp.ld_cache - rp.rope_q;
                        krow[hbase + d1] = o1;
                        krow[hbase + d1 + 64] = o2;
                    }
                }
            } else if (EPI == SK_SWIGLU_T) {
                const float gt = v[0], up = v[NW - 1];
                const float y = sep_rn(gt / (1.0f + fexp(-gt)) * up);
                const T hi = (T)y;
                ((T*)out)[(long)m * ldo + (unit >> 1) * 32 + (unit & 1) * 16 + n] = hi;
                if (nm.hl) ((T*)out)[(long)(m + M) * ldo + (unit >> 1) * 32 + (unit & 1) * 16 + n] = (T)(y - (float)hi);
            } else if (EPI == SK_STORE_T) {
                const T hi = (T)v[0];
                ((T*)out)[(long)m * ldo + row0 + n] = hi;
                if (nm.hl) ((T*)out)[(long)(m + M) * ldo + row0 + n] = (T)(v[0] - (float)hi);
            } else if (EPI == SK_STORE_F32) {
                ((float*)out)[(long)m * ldo + row0 + n] = v[0];
            } else {
                float* xo = (float*)out + (long)m * ldo + row0 + n;
                const float xn = *xo + v[0];
                *xo = xn;
                if (nm.norm_out) {
                    const float y = sep_rn(xn * nm.gamma[row0 + n]);
                    const T hi = (T)y;
                    ((T*)nm.norm_out)[(long)m * nm.ld_norm + row0 + n] = hi;
                    if (nm.hl) ((T*)nm.norm_out)[(long)(m + M) * nm.ld_norm + row0 + n] = (T)(y - (float)hi);
                    sq_s[n][m] = xn * xn;
                }
            }
        }
    }
    if (EPI == SK_RESID_F32 && nm.norm_out) {
        __syncthreads();
        if (tid < M) {
            float q = 0.f;
#pragma unroll
            for (int n = 0; n < 16; ++n) q += sq_s[n][tid];
            nm.rowsq_out[(long)tid * (N >> 4) + unit] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE + KV append for the B rows of a batched decode step: row s is sequence s's next token at position pos[s] (device memory:
// the step is graph-captured); its rotated K and its V go to row s * cache_stride + pos[s] of the pooled caches.
// Same arithmetic per element as rope_kernel.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_rows_kernel(T* qkv, int S, int ld, int n_q, int n_kv, int D, const float* cosT, const float* sinT,
                                 T* k_cache, T* v_cache, int ld_cache, long cache_stride, const int* pos) {
    typedef typename vec_of<T>::x8 T8;
    const int half = D >> 1, cpr = half >> 3;
    const int rot_heads = n_q + n_kv;
    const int per_tok = rot_heads * cpr + n_kv * (D >> 3);
    const long total = (long)S * per_tok;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int s = (int)(idx / per_tok);
        int w = (int)(idx - (long)s * per_tok);
        const int p = pos[s];
        T* row = qkv + (long)s * ld;
        const long crow = (long)s * cache_stride + p;
        if (w < rot_heads * cpr) {
            const int h = w / cpr, c = w - h * cpr;
            T* p1 = row + h * D + c * 8;
            T* p2 = p1 + half;
            const T8 a = *(const T8*)p1, b = *(const T8*)p2;
            const float* cs = cosT + (long)p * half + c * 8;
            const float* sn = sinT + (long)p * half + c * 8;
            T8 oa, ob;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = (float)a[e], x2 = (float)b[e];
                oa[e] = (T)(x1 * cs[e] - x2 * sn[e]);
                ob[e] = (T)(x2 * cs[e] + x1 * sn[e]);
            }
            *(T8*)p1 = oa;
            *(T8*)p2 = ob;
            if (h >= n_q) {
                T* kc = k_cache + crow * ld_cache + (h - n_q) * D + c * 8;
                *(T8*)kc = oa;
                *(T8*)(kc + half) = ob;
            }
        } else {
            w -= rot_heads * cpr;
            const int h = w / (D >> 3), c = w - h * (D >> 3);
            *(T8*)(v_cache + crow * ld_cache + h * D + c * 8) = *(const T8*)(row + (n_q + n_kv + h) * D + c * 8);
        }
    }
}

}  // namespace lmi
