// patch_embed.h — a7 front end: SigLIP's patch convolution (3 -> D, kernel = stride = P, bias) + position embedding as ONE
// im2col + MFMA GEMM (north_star: "ViT patch-conv as an im2col+MFMA GEMM with LDS-staged 14x14 tiles"; reference:
// SiglipVisionEmbeddings via EVAL:268, preceded by SiglipImageProcessor's rescale / normalise, EVAL:403-405).
//
// The P x P x 3 pixel block of a patch never exists in HBM as an im2col row: each k-tile of the GEMM stages the next slice of
// the 128 patches' pixel rows straight from the image (u8 HWC tiles from the GPU tiler, or the processor's fp32 CHW
// pixel_values) into LDS — normalised with the processor's exact arithmetic and rounded to the MFMA operand type on the way —
// next to the matching slice of the weight.  K order is the image's own: k = ky * RP + kx * 3 + c with every pixel row (3P
// values) padded to RP = roundup(3P, 8), so an 8-element chunk is 8 consecutive bytes of one image row (u8 input); the weight
// is laid out to match at load time (weights.py: patch_w_fused, zero in the pad positions).
//
// 128 x 128 output tile, 4 waves (2 x 2, wave tile 64 x 64 = 2 x 2 MFMA 32x32x16 tiles), 64-deep k-tiles in a 2-slot LDS ring
// fed through registers (global loads of tile t+1 in flight under the MFMAs of tile t); 128-byte LDS rows with the 16-byte
// chunk index XOR-ed by the row so that fragment reads and staging writes are bank-conflict free.  The weight rows are the
// MFMA A operand (as in gemm.h), so a lane owns one patch row and 4 consecutive output columns per accumulator quad; the
// epilogue turns each wave's tile through LDS and stores 256-byte row segments of  acc + bias + pos_emb[patch index].
#pragma once
#include "lmi_device.h"

namespace lmi {

struct PatchEmbedArgs {
    const void* pix;         // u8 [n, S, S, 3] or fp32 [n, 3, S, S]
    const void* W;           // T [N, ldw], K order (ky, kx, c) with rows of RP, zero padded to KP
    const float* bias;       // [N]
    const float* pos;        // fp32 [G*G, N]
    float* out;              // fp32 [M, ldo], M = n * G * G
    int M, S, P, G, N, RP, KP, ldw, ldo;
};

constexpr int PE_BM = 128, PE_BN = 128, PE_BK = 64;
constexpr int PE_STAGE = (PE_BM + PE_BN) * PE_BK * 2;            // bytes per ring slot (A then B)
constexpr int PE_EPI_LD = 68;                                    // floats per row of a wave's 64 x 64 epilogue image (+4: banks)
constexpr int PE_SMEM = 4 * 64 * PE_EPI_LD * 4;                  // 69632 >= 2 * PE_STAGE (65536)

template <typename T, bool FROM_U8>
__global__ void __launch_bounds__(256) patch_embed_kernel(PatchEmbedArgs p) {
    typedef typename vec_of<T>::x8 T8;
    LMI_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.N / PE_BN;
    const int mt = blockIdx.x / NT, nt = blockIdx.x - mt * NT;   // the N-tiles of one patch block run together: pixels shared in L2
    const int m0 = mt * PE_BM, n0 = nt * PE_BN;
    const int KT = p.KP / PE_BK, GG = p.G * p.G, RL = 3 * p.P;

    // ---- staging roles: 4 chunks of 8 k per thread for A (pixels) and for B (weights) --------------------------------------
    int a_row[4], a_c8[4];
    long a_base[4];                                              // element offset of the patch's first pixel (ky = 0, kx = 0, c = 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        a_row[i] = id >> 3;
        a_c8[i] = id & 7;
        const int m = imin(m0 + a_row[i], p.M - 1);              // tail rows: clamped loads, masked stores
        const int n = m / GG, rem = m - n * GG, py = rem / p.G, px = rem - py * p.G;
        a_base[i] = FROM_U8 ? (((long)n * p.S + py * p.P) * p.S + px * p.P) * 3
                            : ((long)n * 3 * p.S + py * p.P) * p.S + px * p.P;
    }
    uint64_t a_u8[4];
    f32x8 a_f32[4];
    u32x4 b_reg[4];
    int a_valid[4];

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kt * PE_BK + a_c8[i] * 8;
            const int ky = k / p.RP, j0 = k - ky * p.RP;
            const int valid = ky < p.P ? imin(8, RL - j0) : 0;   // elements of this chunk that are pixels (the rest: K padding)
            a_valid[i] = valid;
            if (FROM_U8) {
                const uint8_t* src = (const uint8_t*)p.pix + a_base[i] + (long)ky * p.S * 3 + j0;
                uint64_t v = 0;
                if (valid == 8) __builtin_memcpy(&v, src, 8);    // 8 consecutive bytes of one image row (2-byte aligned)
                else
                    for (int e = 0; e < valid; ++e) v |= (uint64_t)src[e] << (8 * e);
                a_u8[i] = v;
            } else {
                const float* src = (const float*)p.pix + a_base[i] + (long)ky * p.S;
                const long plane = (long)p.S * p.S;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = j0 + e, kx = j / 3, c = j - 3 * kx;
                    a_f32[i][e] = e < valid ? src[c * plane + kx] : 0.f;
                }
            }
            const int id = tid + 256 * i, row = id >> 3, c8 = id & 7;
            b_reg[i] = *(const u32x4*)((const T*)p.W + (long)(n0 + row) * p.ldw + kt * PE_BK + c8 * 8);
        }
    };
    auto store_tile = [&](int slot) {
        char* As = smem + slot * PE_STAGE;
        char* Bs = As + PE_BM * PE_BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v;
                if (FROM_U8) {
                    const float u = (float)((a_u8[i] >> (8 * e)) & 0xff);
                    v = mul_rn(sub_rn(mul_rn(u, 1.0f / 255.0f), 0.5f), 2.0f);      // the processor's arithmetic, no contraction
                } else {
                    v = a_f32[i][e];
                }
                t[e] = (T)(e < a_valid[i] ? v : 0.f);
            }
            *(T8*)(As + a_row[i] * 128 + ((a_c8[i] ^ (a_row[i] & 7)) << 4)) = t;
            const int id = tid + 256 * i, row = id >> 3, c8 = id & 7;
            *(u32x4*)(Bs + row * 128 + ((c8 ^ (row & 7)) << 4)) = b_reg[i];
        }
    };

    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_tile(kt + 1);
        const char* As = smem + (kt & 1) * PE_STAGE;
        const char* Bs = As + PE_BM * PE_BK * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            T8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                af[i] = *(const T8*)(As + ra * 128 + (((ks * 2 + lh) ^ (ra & 7)) << 4));
                wf[i] = *(const T8*)(Bs + rb * 128 + (((ks * 2 + lh) ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma32(wf[ni], af[mi], acc[mi][ni]);
        }
        if (kt + 1 < KT) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: wave tile -> LDS image -> 256-byte row segments of acc + bias + pos_emb ---------------------------------
    float* img = (float*)smem + wave * 64 * PE_EPI_LD;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
                *(f32x4*)(img + (mi * 32 + lr) * PE_EPI_LD + ni * 32 + 8 * q + 4 * lh) = v;
            }
    wave_lds_fence();
    const int c = (lane & 15) * 4, ng = n0 + wn * 64 + c;
    const f32x4 bias = *(const f32x4*)(p.bias + ng);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + (lane >> 4), mg = m0 + wm * 64 + r;
        if (mg < p.M) {
            const f32x4 v = *(const f32x4*)(img + r * PE_EPI_LD + c);
            const f32x4 pe = *(const f32x4*)(p.pos + (long)(mg % GG) * p.N + ng);
            *(f32x4*)(p.out + (long)mg * p.ldo + ng) = (v + bias) + pe;
        }
    }
}

}  // namespace lmi
