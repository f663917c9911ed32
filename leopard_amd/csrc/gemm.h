// gemm.h — the MFMA GEMM family:  out[M,N] = epilogue( A[M,K] . W[N,K]^T )     (nn.Linear layout)
//
// Reference ops served: SigLIP patch-embed (im2col rows), q/k/v/out, fc1/fc2; the
// projector's linear_1 (with the 2x2 pixel-shuffle folded into the A-row gather, EVAL:165-192) and
// linear_2; Llama q/k/v/o, gate/up (SwiGLU fused in the epilogue), down; all-position lm_head.
//
// Kernel shape (gfx950): 128x128 output tile, BK=64, 256 threads = 4 waves in a 2x2 grid, each wave
// owns a 64x64 sub-tile as 2x2 v_mfma_f32_32x32x16 accumulators (64 fp32 / lane).
//   * Both operands are K-contiguous, staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     double buffered (2 x 32 KiB): the next k-tile streams in while the current one feeds the MFMAs.
//   * LDS image is lane-linear per wave instruction (hardware rule), so the bank-conflict swizzle is applied
//     on the SOURCE address and again on the ds_read_b128 address: 16-byte chunk c of row r is stored at
//     chunk c ^ ((r>>1)&7).  With 128-byte rows this makes every 16-lane ds_read_b128 group hit 16
//     distinct 16-byte slots of the 256-byte bank row (checked in tests/test_emu_kernels.py).
//   * The MFMA is issued "swapped": A-operand = W rows (n), B-operand = A rows (m), so that a lane owns ONE
//     output row m and 4 consecutive n per accumulator quad -> 8/16-byte vector epilogue stores, row-wise
//     fused epilogues (bias, GELU, residual add into the fp32 stream, SwiGLU on interleaved gate/up blocks).
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs each get a contiguous slab of the tile space, walked in
//     groups of GROUP_M row-tiles so that a slab's A/W panels stay in that XCD's private 4 MiB L2.
// Requirements (met by weight preparation, leopard_amd/engine.py): N % 128 == 0, K % 64 == 0, 16-byte
// aligned rows.  M is arbitrary (tail rows are clamped on load and masked on store).
#pragma once
#include "lmi_device.h"

namespace lmi {

enum { EPI_STORE_T = 0, EPI_RESID_F32 = 1, EPI_STORE_F32 = 2, EPI_SWIGLU_T = 3 };
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2 };
enum { AMODE_PLAIN = 0, AMODE_PIXSHUF = 1 };

struct GemmArgs {
    const void* A;        // [M, K] (plain) or ViT output [tiles*G*G, C] (pixel-shuffle mode)
    const void* W;        // [N, K]
    void* out;            // T or fp32, leading dimension ldo
    const float* bias;    // [N] or null (for SwiGLU: unused)
    const float* addmat;  // optional [add_period, N] fp32 matrix added row-periodically (SigLIP pos-emb)
    const int* row_map;   // optional: output row of logical row m (scatter into the merged sequence)
    int M, N, K;
    int lda, ldw, ldo;
    int add_period;
    int ps_grid;          // pixel-shuffle: G (26); output tokens per tile = (G/2)^2; C = K/4
};

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64, GEMM_THREADS = 256;
constexpr int GEMM_TILE_BYTES = GEMM_BM * GEMM_BK * 2;            // 16 KiB per operand per stage
constexpr int GEMM_SMEM_BYTES = 4 * GEMM_TILE_BYTES;              // 2 stages x (A + W) = 64 KiB
constexpr int GEMM_GROUP_M = 8;

LMI_DEV float act_apply(float x, int act) {
    if (act == ACT_GELU_TANH) {
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        return 0.5f * x * (1.0f + tanhf(u));
    }
    if (act == ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    return x;
}

// byte offset of logical 16-byte chunk `lc` of row `r` inside a [128][64] 16-bit tile
LMI_DEV int gemm_lds_off(int r, int lc) { return r * 128 + ((lc ^ ((r >> 1) & 7)) << 4); }

// XCD-aware, grouped tile order (bijective for any tile count)
LMI_DEV void gemm_tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = GEMM_GROUP_M * tiles_n;
    const int g = swz / per_group;
    const int first_m = g * GEMM_GROUP_M;
    const int gsize = imin(GEMM_GROUP_M, tiles_m - first_m);
    const int in_g = swz - g * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
}

template <typename T, int EPI, int ACT, int AMODE>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_kernel(GemmArgs p) {
    typedef typename vec_of<T>::x8 T8;
    typedef typename vec_of<T>::x4 T4;
    LMI_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM, tiles_n = p.N / GEMM_BN;
    int tm, tn;
    gemm_tile_coords((int)blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    // ---- per-thread staging sources: 4 passes x (one A row, one W row); chunk position is fixed --------
    const int srow = tid >> 3;                       // physical row inside a 32-row pass
    const int pc = tid & 7;                          // physical 16-byte chunk inside the 128-byte row
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 32 + srow;
        const int lc = pc ^ ((r >> 1) & 7);
        int am = imin(m0 + r, p.M - 1);
        long arow;
        if (AMODE == AMODE_PIXSHUF) {
            const int g = p.ps_grid, h = g >> 1, per = h * h;
            const int tile = am / per, pp = am - tile * per;
            const int ph = pp / h, pw = pp - ph * h;
            arow = (long)tile * g * g + (long)(2 * ph) * g + 2 * pw;
        } else {
            arow = am;
        }
        a_src[ps] = (const char*)p.A + (arow * p.lda + lc * 8) * 2;
        w_src[ps] = (const char*)p.W + ((long)(n0 + r) * p.ldw + lc * 8) * 2;
    }
    const int ps_c = (AMODE == AMODE_PIXSHUF) ? (p.K >> 2) : 0;     // channels per shuffle segment

    auto stage = [&](int kt, int buf) {
        char* a_dst = smem + buf * 2 * GEMM_TILE_BYTES + wave * 1024;
        char* w_dst = a_dst + GEMM_TILE_BYTES;
        long a_off = (long)kt * GEMM_BK * 2;
        if (AMODE == AMODE_PIXSHUF) {
            const int k0 = kt * GEMM_BK;
            const int seg = k0 / ps_c;                               // 0..3 = (dh, dw)
            a_off = ((long)((seg >> 1) * p.ps_grid + (seg & 1)) * p.lda + (k0 - seg * ps_c)) * 2;
        }
        const long w_off = (long)kt * GEMM_BK * 2;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            glds16(a_src[ps] + a_off, a_dst + ps * 4096);
            glds16(w_src[ps] + w_off, w_dst + ps * 4096);
        }
    };

    f32x16 acc[2][2];                                 // [ni][mi]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int nt = p.K / GEMM_BK;
    stage(0, 0);
#ifndef LMI_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) stage(t + 1, buf ^ 1);
        const char* a_t = smem + buf * 2 * GEMM_TILE_BYTES;
        const char* w_t = a_t + GEMM_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            T8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const T8*)(a_t + gemm_lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                wf[i] = *(const T8*)(w_t + gemm_lds_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = mfma32(wf[ni], af[mi], acc[ni][mi]);
        }
#ifndef LMI_EMU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
    }

    // ---- epilogue: lane owns row m = .. + fr and 4 consecutive n per accumulator quad ------------------
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + fr;
        if (m >= p.M) continue;
        const long orow = p.row_map ? (long)p.row_map[m] : (long)m;
        if (EPI == EPI_SWIGLU_T) {
            // W rows are interleaved in 32-row blocks [gate | up]; this wave holds gate in ni=0, up in ni=1
            const int ocol0 = ((n0 + wn * 64) >> 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = acc[0][mi][q * 4 + e], u = acc[1][mi][q * 4 + e];
                    v[e] = (T)(g / (1.0f + lmi::fexp(-g)) * u);
                }
                *(T4*)((T*)p.out + orow * p.ldo + ocol0 + q * 8 + fh * 4) = v;
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + ni * 32 + q * 8 + fh * 4;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][q * 4 + e];
                    if (p.bias) {
                        const f32x4 b = *(const f32x4*)(p.bias + n);
                        v += b;
                    }
                    if (p.addmat) {
                        const f32x4 a = *(const f32x4*)(p.addmat + (long)(m % p.add_period) * p.N + n);
                        v += a;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ACT);
                    if (EPI == EPI_STORE_T) {
                        T4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (T)v[e];
                        *(T4*)((T*)p.out + orow * p.ldo + n) = o;
                    } else if (EPI == EPI_RESID_F32) {
                        f32x4* dst = (f32x4*)((float*)p.out + orow * p.ldo + n);
                        *dst = *dst + v;
                    } else {
                        *(f32x4*)((float*)p.out + orow * p.ldo + n) = v;
                    }
                }
        }
    }
}

}  // namespace lmi
