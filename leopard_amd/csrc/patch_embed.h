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
#include <type_traits>

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

struct PeTab { int x, y; };
constexpr int PE_MAX_KT = 16;                                    // k-tiles the per-k-tile gather table is sized for (P <= 21)

template <typename T, bool FROM_U8>
__global__ void __launch_bounds__(256, 2) patch_embed_kernel(PatchEmbedArgs p) {
    typedef typename vec_of<T>::x8 T8;
    LMI_DYN_SMEM(smem);
    // where the 8-element chunk c8 of k-tile kt comes from, relative to the patch's first pixel — the same for every patch, so it
    // is worked out once per workgroup (the only integer divisions of the kernel) instead of per thread and k-tile:
    //   u8 input:   tab[kt*8 + c8] = {byte offset of the chunk's first pixel value, number of real values (0: K padding)}
    //   fp32 input: etab[kt*8 + c8][e] = element offset of value e (channel plane + row + column), -1: padding
    __shared__ PeTab tab[PE_MAX_KT * 8];
    __shared__ int etab[FROM_U8 ? 1 : PE_MAX_KT * 8][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.N / PE_BN;
    const int mt = blockIdx.x / NT, nt = blockIdx.x - mt * NT;   // the N-tiles of one patch block run together: pixels shared in L2
    const int m0 = mt * PE_BM, n0 = nt * PE_BN;
    const int KT = p.KP / PE_BK, GG = p.G * p.G, RL = 3 * p.P;
    if (tid < KT * 8) {
        const int k = tid * 8, ky = k / p.RP, j0 = k - ky * p.RP;
        const int valid = ky < p.P ? imax(0, imin(8, RL - j0)) : 0;
        if (FROM_U8) {
            // a partial chunk (the tail of a pixel row) is read as the 8 bytes that END at its last value and shifted down, so no
            // load ever leaves the image
            tab[tid] = PeTab{ky * p.S * 3 + j0 - (valid ? 8 - valid : 0), valid};
        } else {
            tab[tid] = PeTab{0, valid};
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e, kx = j / 3, c = j - 3 * kx;
                etab[FROM_U8 ? 0 : tid][e] = e < valid ? c * p.S * p.S + ky * p.S + kx : -1;
            }
        }
    }
    __syncthreads();

    // ---- staging roles: thread -> chunk c8 = tid & 7 of rows (tid >> 3) + 32 i, for A (pixels) and for B (weights) ---------
    const int c8 = tid & 7, row0 = tid >> 3;
    const char* a_src[4];                                        // the patch's first pixel (ky = 0, kx = 0, c = 0)
    const T* b_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 32 * i;
        const int m = imin(m0 + row, p.M - 1);                   // tail rows: clamped loads, masked stores
        const int n = m / GG, rem = m - n * GG, py = rem / p.G, px = rem - py * p.G;
        a_src[i] = FROM_U8 ? (const char*)p.pix + (((long)n * p.S + py * p.P) * p.S + px * p.P) * 3
                           : (const char*)((const float*)p.pix + ((long)n * 3 * p.S + py * p.P) * p.S + px * p.P);
        b_src[i] = (const T*)p.W + (long)(n0 + row) * p.ldw + c8 * 8;
        // LDS rows are 128 bytes (64 k); the 16-byte chunk index is XOR-ed with (row >> 1) & 7: rows r and r + 1 sit in different
        // halves of the 64 banks, so the 16 rows one ds_read_b128 group touches cover every bank once
        lds_off[i] = row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4);
    }
    // registers of one k-tile on its way from global memory to LDS; two sets, so that the loads of tile t+2 are in flight while
    // tile t is multiplied and tile t+1 is converted and written
    struct Stage {
        uint32_t lo[4], hi[4];
        f32x8 f[FROM_U8 ? 1 : 4];
        u32x4 b[4];
    };
    Stage st0, st1;

    auto load_tile = [&](Stage& st, int kt) {
        const PeTab t = tab[kt * 8 + c8];
        if (FROM_U8) {
            const int sh = t.y ? 8 * (8 - t.y) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint64_t v = 0;
                if (t.y) __builtin_memcpy(&v, a_src[i] + t.x, 8);              // 8 consecutive bytes of one image row (2-byte aligned)
                v >>= sh;
                st.lo[i] = (uint32_t)v;
                st.hi[i] = (uint32_t)(v >> 32);
            }
        } else {
            int eo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) eo[e] = etab[FROM_U8 ? 0 : kt * 8 + c8][e];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) st.f[FROM_U8 ? 0 : i][e] = eo[e] >= 0 ? ((const float*)a_src[i])[eo[e]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) st.b[i] = *(const u32x4*)(b_src[i] + kt * PE_BK);
    };
    // u8 -> operand: the processor's arithmetic, (u * (1/255) - 0.5) * 2 without contraction.  For f16 operands one FMA gives the
    // same 16-bit value for every one of the 256 inputs (tests/test_emu_patch_embed.py checks all of them through this kernel);
    // for bf16 one input differs, so bf16 keeps the three-step form.  Padding positions may hold any finite value: their weights
    // are zero.
    auto norm = [&](float u) -> float {
        if (std::is_same<T, f16_t>::value) return __builtin_fmaf(u, 2.0f / 255.0f, -1.0f);
        return mul_rn(sub_rn(mul_rn(u, 1.0f / 255.0f), 0.5f), 2.0f);
    };
    auto store_tile = [&](const Stage& st, int slot) {
        char* As = smem + slot * PE_STAGE;
        char* Bs = As + PE_BM * PE_BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T8 t;
            if (FROM_U8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t[e] = (T)norm((float)((st.lo[i] >> (8 * e)) & 0xff));
                    t[4 + e] = (T)norm((float)((st.hi[i] >> (8 * e)) & 0xff));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (T)st.f[FROM_U8 ? 0 : i][e];
            }
            *(T8*)(As + lds_off[i]) = t;
            *(u32x4*)(Bs + lds_off[i]) = st.b[i];
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

    // this wave's LDS writes are complete, then workgroup barrier — NOT __syncthreads(), which would also wait for the global loads
    // of the tile after next that are meant to stay in flight across it
    auto publish = [&]() {
        lds_write_drain();
        raw_barrier();
    };
    auto multiply = [&](int slot) {
        const char* As = smem + slot * PE_STAGE;
        const char* Bs = As + PE_BM * PE_BK * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            T8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + lr, rb = wn * 64 + i * 32 + lr;
                af[i] = *(const T8*)(As + ra * 128 + (((ks * 2 + lh) ^ ((ra >> 1) & 7)) << 4));
                wf[i] = *(const T8*)(Bs + rb * 128 + (((ks * 2 + lh) ^ ((rb >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma32(wf[ni], af[mi], acc[mi][ni]);
        }
    };

    load_tile(st0, 0);
    store_tile(st0, 0);
    if (FROM_U8) {
        // u8 pixels: 2 registers per chunk, so two tiles' loads fit in flight (distance 2); unrolled by two: static stage registers
        if (KT > 1) load_tile(st1, 1);
        publish();
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 2 < KT) load_tile(st0, kt + 2);            // tile kt in slot 0, tile kt + 1 in flight in st1
            multiply(0);
            if (kt + 1 < KT) store_tile(st1, 1);
            publish();
            if (kt + 1 >= KT) break;
            if (kt + 3 < KT) load_tile(st1, kt + 3);             // tile kt + 1 in slot 1, tile kt + 2 in flight in st0
            multiply(1);
            if (kt + 2 < KT) store_tile(st0, 0);
            publish();
        }
    } else {
        // fp32 pixel_values (the processor's output; the slower, compatibility input): 8 registers per chunk, distance 1
        publish();
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) load_tile(st0, kt + 1);
            multiply(kt & 1);
            if (kt + 1 < KT) store_tile(st0, (kt + 1) & 1);
            publish();
        }
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
    int pos_row = (m0 + wm * 64 + (lane >> 4)) % GG;            // patch index within its tile; advanced by 4 rows per iteration
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + (lane >> 4), mg = m0 + wm * 64 + r;
        if (mg < p.M) {
            const f32x4 v = *(const f32x4*)(img + r * PE_EPI_LD + c);
            const f32x4 pe = *(const f32x4*)(p.pos + (long)pos_row * p.N + ng);
            *(f32x4*)(p.out + (long)mg * p.ldo + ng) = (v + bias) + pe;
        }
        pos_row += 4;
        while (pos_row >= GG) pos_row -= GG;
    }
}

}  // namespace lmi
