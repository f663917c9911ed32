// lmi_device.h — the few hardware primitives every Leopard-MI kernel is written against.
//
// Device build (hipcc --offload-arch=gfx950): thin inline wrappers over the CDNA4 builtins
// (v_mfma_f32_32x32x16_{bf16,f16}, buffer_load_dwordx4 ... lds, ds_read_b64_tr_b16, wave64 shuffles).
//
// LMI_EMU build (host clang, tools/hipemu): the same names implemented by a lock-step fibre emulator
// so that kernel *logic* (tile indexing, swizzles, masks, epilogues) can be unit-tested on a machine
// without a GPU.  The emulator is a development tool only; the product library never contains it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef LMI_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#endif

namespace lmi {

typedef _Float16 f16_t;
typedef __bf16 bf16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// OCP fp8 e4m3fn storage (gfx950's fp8 MFMA format; NOT the MI300 fnuz encoding): operand type of the fp8 GEMM path
struct fp8_t { uint8_t v; };
typedef uint8_t u8x8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <typename T> struct vec_of;
template <> struct vec_of<f16_t> { typedef f16x8 x8; typedef f16x4 x4; typedef f16x2 x2; };
template <> struct vec_of<bf16_t> { typedef bf16x8 x8; typedef bf16x4 x4; typedef bf16x2 x2; };
template <> struct vec_of<float> { typedef f32x8 x8; typedef f32x4 x4; typedef f32x2 x2; };   // fp32-output norms
template <> struct vec_of<fp8_t> { typedef u8x8 x8; };                                          // fp8 outputs (8 bytes per lane)

#define LMI_WAVE 64

#ifndef LMI_EMU
// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
#define LMI_DEV __device__ __forceinline__
#define LMI_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]

LMI_DEV int lane_id() { return threadIdx.x & 63; }
LMI_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }   // provably wave-uniform

// Wait until at most N of this wave's VMEM operations (LDS-DMA pieces) are outstanding, then workgroup barrier.
// Raw s_barrier on purpose: __syncthreads() would drain vmcnt to 0 while LDS-DMA is in flight.
template <int N> LMI_DEV void wait_vmcnt_barrier() { asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory"); }
template <int N> LMI_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// D[32x32] += A[32x16] * B[16x32].  Lane l supplies A[l&31][8*(l>>5)+j] and B[8*(l>>5)+j][l&31], j=0..7;
// receives D[(r&3)+8*(r>>2)+4*(l>>5)][l&31], r=0..15.
LMI_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
LMI_DEV f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// D[16x16] += A[16x32] * B[32x16].  Lane l supplies A[l&15][8*(l>>4)+j] and B[8*(l>>4)+j][l&15], j=0..7; receives
// D[4*(l>>4)+r][l&15], r=0..3.  (Skinny-M decode GEMMs: 16 weight rows x up to 16 batch rows per instruction.)
LMI_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
LMI_DEV f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// async 16-byte global -> LDS copy through a buffer resource: LDS destination = lds_wave_base + lane*16 (wave-uniform base);
// source byte = base + voffset (per lane) + soffset (wave-uniform), both 32-bit; bytes at or beyond num_records read as zero
// (raw-buffer range check), so ragged tile tails need no address clamp.
struct BufRsrc {
    __amdgpu_buffer_rsrc_t r;
};
LMI_DEV BufRsrc make_buf(const void* base, unsigned num_records) {
    return BufRsrc{__builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)num_records, 0x00020000)};
}
// AUX = 2: non-temporal (a stream every byte of which is read once — the K / V cache of a decode step).
template <int AUX = 0>
LMI_DEV void glds16_buf(const BufRsrc& b, unsigned voffset, unsigned soffset, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voffset,
                                             (int)soffset, 0, AUX);
}
// the 4-byte form (LDS destination = lds_wave_base + lane*4): the block scales of the low-bit correction phase
LMI_DEV void glds4_buf(const BufRsrc& b, unsigned voffset, unsigned soffset, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, (int)voffset, (int)soffset, 0, 0);
}

// D[32x32] += A[32x64] * B[64x32] on fp8 e4m3 operands at twice the 16-bit MFMA rate (v_mfma_scale_f32_32x32x64_f8f6f4; the
// unscaled fp8 MFMA runs at the 16-bit rate).  Lane l supplies A[l&31][32*(l>>5) + j] and B[32*(l>>5) + j][l&31], j = 0..31
// (32 bytes; layout measured with tools/ubench/mfma_fp8_layout.hip), receives D as mfma32.  The E8M0 block scales are used as
// ONE power-of-two factor for the whole product: A's scale is 127 (2^0), B's is `scale_b` in every byte (127 + e -> x 2^e).
LMI_DEV f32x16 mfma32_fp8(v8i a, v8i b, f32x16 c, int scale_b) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, scale_b);
}
// fp32 -> fp8 e4m3fn, round to nearest even, saturating at +-448
LMI_DEV uint8_t to_fp8(float x) {
    x = fminf(fmaxf(x, -448.0f), 448.0f);
    return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xff);
}
// D[32x32] += A[32x64] * B[64x32] on fp4 e2m1 operands with MX block scales, at FOUR times the 16-bit MFMA rate (the low-bit correction
// phase of gemm.h).  Lane l supplies A[l&31][32*(l>>5) + j] and B[32*(l>>5) + j][l&31], j = 0..31, as 16 bytes (element j in nibble j & 1 of
// byte j >> 1, even element low; layout and scale semantics pinned by tools/ubench/mfma_fp4_layout.hip), and ONE E8M0 scale per operand for
// its 32 elements — i.e. per (row, 32-k block): byte SEL_A of `scale_a` / byte SEL_B of `scale_b`.
template <int SEL_A, int SEL_B>
LMI_DEV f32x16 mfma32_fp4(u32x4 a, u32x4 b, f32x16 c, int scale_a, int scale_b) {
    const v8i av = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0}, bv = {(int)b[0], (int)b[1], (int)b[2], (int)b[3], 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, SEL_A, scale_a, SEL_B, scale_b);
}
// 8 fp32 values / scale -> 8 e2m1 codes in one dword (element e in nibble e): v_cvt_scalef32_pk_fp4_f32, two elements per instruction.
// Round to nearest even, saturating at 6 — bit for bit the software rule to_fp4() below (every tie and the saturation range checked on
// the device: tools/ubench/mfma_fp4_layout.hip, profiles/r05_mfma_fp4_layout.txt), which the host emulator build uses.
LMI_DEV unsigned fp4_pack8(const float (&v)[8], float scale, float) {
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[0], v[1], scale, 0);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[2], v[3], scale, 1);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[4], v[5], scale, 2);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[6], v[7], scale, 3);
    return r;
}
// max over the 4 lanes of a lane quad (lanes 4q .. 4q+3), DPP quad permutes: no LDS traffic
LMI_DEV float quad_max(float v) {
    int x = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));      // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
}
// lane 4q + i of a quad receives `v` of lane 4q + SRC
template <int SRC> LMI_DEV unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, SRC * 0x55, 0xF, 0xF, false);
}


LMI_DEV void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }
// this wave's LDS writes have completed (before a raw s_barrier that publishes them to the other waves)
LMI_DEV void lds_write_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Between a wave's LDS writes and its reads of what OTHER lanes of the same wave wrote.  The hardware needs nothing (the DS
// operations of a wave execute in order); the empty asm only stops the compiler from reordering across it.
LMI_DEV void wave_lds_fence() { asm volatile("" ::: "memory"); }

LMI_DEV u32x2 ds_read_tr16_b64(const void* lds_ptr) {
    u32x2 r;
    unsigned addr = (unsigned)(size_t)lds_ptr;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    return r;
}


// Batched form used by attention: for d-block i = 0..N-1 read the 4-key groups at byte offsets i*64 and
// i*64 + ROW8 from one base address, one wait for all 2N reads.  out[i] = {lo.x, lo.y, hi.x, hi.y}.
template <int N, int ROW8>
LMI_DEV void ds_read_tr16_batch(const void* lds_ptr, u32x4* out) {
    static_assert(N == 3 || N == 4, "d-block count");
    unsigned addr = (unsigned)(size_t)lds_ptr;
    u32x2 a0, a1, a2, a3, b0, b1, b2, b3;
    if constexpr (N == 4) {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8 offset:%c9\n\t"
            "ds_read_b64_tr_b16 %1, %8 offset:%c10\n\t"
            "ds_read_b64_tr_b16 %2, %8 offset:%c11\n\t"
            "ds_read_b64_tr_b16 %3, %8 offset:%c12\n\t"
            "ds_read_b64_tr_b16 %4, %8 offset:%c13\n\t"
            "ds_read_b64_tr_b16 %5, %8 offset:%c14\n\t"
            "ds_read_b64_tr_b16 %6, %8 offset:%c15\n\t"
            "ds_read_b64_tr_b16 %7, %8 offset:%c16\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2), "=&v"(a3), "=&v"(b3)
            : "v"(addr), "i"(0), "i"(ROW8), "i"(64), "i"(64 + ROW8), "i"(128), "i"(128 + ROW8), "i"(192), "i"(192 + ROW8)
            : "memory");
        out[3] = u32x4{a3[0], a3[1], b3[0], b3[1]};
    } else {
        asm volatile(
            "ds_read_b64_tr_b16 %0, %6 offset:%c7\n\t"
            "ds_read_b64_tr_b16 %1, %6 offset:%c8\n\t"
            "ds_read_b64_tr_b16 %2, %6 offset:%c9\n\t"
            "ds_read_b64_tr_b16 %3, %6 offset:%c10\n\t"
            "ds_read_b64_tr_b16 %4, %6 offset:%c11\n\t"
            "ds_read_b64_tr_b16 %5, %6 offset:%c12\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2)
            : "v"(addr), "i"(0), "i"(ROW8), "i"(64), "i"(64 + ROW8), "i"(128), "i"(128 + ROW8)
            : "memory");
    }
    out[0] = u32x4{a0[0], a0[1], b0[0], b0[1]};
    out[1] = u32x4{a1[0], a1[1], b1[0], b1[1]};
    out[2] = u32x4{a2[0], a2[1], b2[0], b2[1]};
}

// Split form for software pipelining: tr16_issue() starts the 2N transpose reads of one 16-key group (d-block db at
// base + off[db] + imm, second key quad HI bytes further) and returns immediately; the registers may only be consumed
// after lgkm_fence<CNT>() on them, CNT = number of LDS operations issued after the group that may still be in flight
// (LDS returns in order).  Inline asm on purpose: hipcc orders every LDS-reading *intrinsic* behind all outstanding
// LDS-DMA with s_waitcnt vmcnt(0), which would serialise the next tile's DMA with this tile's reads.
template <int N, int HI>
LMI_DEV void tr16_issue(const void* base, const int (&off)[N], int imm, u32x2 (&lo)[N], u32x2 (&hi)[N]) {
    const unsigned b = (unsigned)(size_t)base + (unsigned)imm;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const unsigned a = b + off[i];
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%c3"
                     : "=&v"(lo[i]), "=&v"(hi[i]) : "v"(a), "i"(HI) : "memory");
    }
}
template <int CNT, int N>
LMI_DEV void lgkm_fence(u32x2 (&lo)[N], u32x2 (&hi)[N]) {
    static_assert(N == 3 || N == 4, "d-block count");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%c8)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]),
                     "+v"(lo[3]), "+v"(hi[3]) : "i"(CNT));
    else
        asm volatile("s_waitcnt lgkmcnt(%c6)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2])
                     : "i"(CNT));
}

LMI_DEV int shfl_idx(int v, int src) { return __shfl(v, src, 64); }                       // ds_bpermute_b32: lane <- lane src
LMI_DEV float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
LMI_DEV int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
// v_permlane32_swap_b32: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (no LDS round trip)
LMI_DEV void swap_hi_lo(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
LMI_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
LMI_DEV void lmi_trap() { __builtin_trap(); }
LMI_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
LMI_DEV void setprio_hi() { __builtin_amdgcn_s_setprio(1); }
LMI_DEV void setprio_lo() { __builtin_amdgcn_s_setprio(0); }
LMI_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
LMI_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
LMI_DEV float fexp(float x) { return __expf(x); }
// individually rounded fp32 multiply / subtract: the empty asm makes the result opaque so that hipcc cannot contract
// it into an FMA with a neighbouring operation (HIP's __fmul_rn is a plain `*` and does get contracted)
LMI_DEV float mul_rn(float a, float b) { float r = a * b; asm volatile("" : "+v"(r)); return r; }
LMI_DEV float sub_rn(float a, float b) { float r = a - b; asm volatile("" : "+v"(r)); return r; }
// an fp32 result about to be rounded to a 16-bit type, made opaque first: hipcc otherwise fuses the producing multiply / fma with the conversion
// into v_fma_mixlo_f16 (ONE rounding) in some instantiations of an epilogue and not in others (fp32 rounding, then the conversion: two) — a
// row's 16-bit result then depends on WHICH kernel variant computed it, in one element out of ~3e5 (round 6: the unselected rows of a
// row-selective lo4 launch against the fast kernel, RoPE outputs).  Not volatile: free to move, only not to fuse.
LMI_DEV float sep_rn(float v) { asm("" : "+v"(v)); return v; }

#define LMI_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)

#else
// ------------------------------------------------------------------------------------------------
// host emulation (tools/hipemu/hipemu.cpp)
// ------------------------------------------------------------------------------------------------
#define LMI_DEV inline
#define LMI_DYN_SMEM(name) char* name = hipemu::dyn_smem()

inline int lane_id() { return threadIdx.x & 63; }
inline int wave_id() { return (int)(threadIdx.x >> 6); }
template <int N> inline void wait_vmcnt_barrier() { hipemu::dma_wait(N); hipemu::barrier(); }     // the counted wait is modelled: hipemu.h
template <int N> inline void wait_vmcnt() { hipemu::dma_wait(N); }
inline float emu_fp8_decode(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);                        // subnormal: m * 2^-3 * 2^-6
    else if (e == 15 && m == 7) f = NAN;                         // e4m3fn: the only NaN; no infinities
    else f = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return s ? -f : f;
}
inline uint8_t to_fp8(float x) {                                 // RNE, saturating at +-448 (same results as v_cvt_pk_fp8_f32 after the clamp)
    x = fminf(fmaxf(x, -448.0f), 448.0f);
    const uint8_t sign = __builtin_signbit(x) ? 0x80 : 0;
    float a = fabsf(x);
    if (a == 0) return sign;
    int e;
    (void)frexpf(a, &e);                                         // a = f * 2^e, f in [0.5, 1)
    int E = e - 1;                                               // a = 1.xxx * 2^E
    if (E < -6) E = -6;                                          // subnormal range: fixed scale 2^-6
    const float q = nearbyintf(ldexpf(a, 3 - E));                // mantissa in units of 2^(E-3), ties to even
    int mant = (int)q;
    if (mant >= 16) { mant = 8; E += 1; }                        // rounded up to the next binade
    if (mant < 8) return sign | (uint8_t)mant;                   // subnormal (E == -6 and mant < 8) or zero
    return sign | (uint8_t)(((E + 7) << 3) | (mant - 8));
}
inline f32x16 mfma32_fp8(v8i a, v8i b, f32x16 c, int scale_b) {
    struct Slot { uint8_t a[32], b[32]; };
    Slot* s = (Slot*)hipemu::wave_buf();
    const int l = lane_id();
    __builtin_memcpy(s[l].a, &a, 32);
    __builtin_memcpy(s[l].b, &b, 32);
    hipemu::wave_sync();
    const float sc = ldexpf(1.0f, (scale_b & 255) - 127);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = 0.f;
        for (int k = 0; k < 64; ++k) acc += emu_fp8_decode(s[row + 32 * (k >> 5)].a[k & 31]) * emu_fp8_decode(s[col + 32 * (k >> 5)].b[k & 31]);
        c[r] += acc * sc;
    }
    hipemu::wave_sync();
    return c;
}
inline float emu_fp4_decode(unsigned code) {
    static const float g[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (code & 8) ? -g[code & 7] : g[code & 7];
}
template <int SEL_A, int SEL_B>
inline f32x16 mfma32_fp4(u32x4 a, u32x4 b, f32x16 c, int scale_a, int scale_b) {
    struct Slot { uint8_t a[16], b[16]; int sa, sb; };
    Slot* s = (Slot*)hipemu::wave_buf();
    const int l = lane_id();
    __builtin_memcpy(s[l].a, &a, 16);
    __builtin_memcpy(s[l].b, &b, 16);
    s[l].sa = (scale_a >> (8 * SEL_A)) & 255;
    s[l].sb = (scale_b >> (8 * SEL_B)) & 255;
    hipemu::wave_sync();
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float tot = 0.f;
        for (int kb = 0; kb < 2; ++kb) {
            const Slot &sa = s[row + 32 * kb], &sb = s[col + 32 * kb];
            float acc = 0.f;
            for (int j = 0; j < 32; ++j)
                acc += emu_fp4_decode((sa.a[j >> 1] >> (4 * (j & 1))) & 15) * emu_fp4_decode((sb.b[j >> 1] >> (4 * (j & 1))) & 15);
            tot += acc * ldexpf(1.0f, sa.sa - 127) * ldexpf(1.0f, sb.sb - 127);
        }
        c[r] += tot;
    }
    hipemu::wave_sync();
    return c;
}
inline void raw_barrier() { hipemu::barrier(); }                                                 // s_barrier alone retires no LDS-DMA piece
inline void lds_write_drain() {}
inline void wave_lds_fence() { hipemu::wave_sync(); }

template <typename V8>
inline f32x16 emu_mfma32(V8 a, V8 b, f32x16 c) {
    struct Slot { float a[8], b[8]; };
    Slot* s = (Slot*)hipemu::wave_buf();
    const int l = lane_id();
    for (int j = 0; j < 8; ++j) { s[l].a[j] = (float)a[j]; s[l].b[j] = (float)b[j]; }
    hipemu::wave_sync();
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc += s[row + 32 * (k >> 3)].a[k & 7] * s[col + 32 * (k >> 3)].b[k & 7];
        c[r] = acc;
    }
    hipemu::wave_sync();
    return c;
}
template <typename V8>
inline f32x4 emu_mfma16(V8 a, V8 b, f32x4 c) {
    struct Slot { float a[8], b[8]; };
    Slot* s = (Slot*)hipemu::wave_buf();
    const int l = lane_id();
    for (int j = 0; j < 8; ++j) { s[l].a[j] = (float)a[j]; s[l].b[j] = (float)b[j]; }
    hipemu::wave_sync();
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc += s[row + 16 * (k >> 3)].a[k & 7] * s[col + 16 * (k >> 3)].b[k & 7];
        c[r] = acc;
    }
    hipemu::wave_sync();
    return c;
}
inline f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return emu_mfma16(a, b, c); }
inline f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return emu_mfma16(a, b, c); }
inline f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return emu_mfma32(a, b, c); }
inline f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return emu_mfma32(a, b, c); }

struct BufRsrc {
    const char* base;
    unsigned num_records;
};
inline BufRsrc make_buf(const void* base, unsigned num_records) { return BufRsrc{(const char*)base, num_records}; }
template <int AUX = 0>
inline void glds16_buf(const BufRsrc& b, unsigned voffset, unsigned soffset, void* lds_wave_base) {
    char* dst = (char*)lds_wave_base + lane_id() * 16;
    const unsigned long off = (unsigned long)voffset + soffset;
    hipemu::dma_issue(dst, off + 16 <= b.num_records ? b.base + off : nullptr, 16);
}

inline void glds4_buf(const BufRsrc& b, unsigned voffset, unsigned soffset, void* lds_wave_base) {
    char* dst = (char*)lds_wave_base + lane_id() * 4;
    const unsigned long off = (unsigned long)voffset + soffset;
    hipemu::dma_issue(dst, off + 4 <= b.num_records ? b.base + off : nullptr, 4);
}

inline u32x2 ds_read_tr16_b64(const void* lds_ptr) {
    struct Slot { uint16_t e[4]; };
    Slot* s = (Slot*)hipemu::wave_buf();
    const int l = lane_id();
    __builtin_memcpy(s[l].e, lds_ptr, 8);
    hipemu::wave_sync();
    const int grp = l & ~15, c = l & 15;
    uint16_t o[4];
    for (int j = 0; j < 4; ++j) o[j] = s[grp + 4 * j + (c >> 2)].e[c & 3];
    hipemu::wave_sync();
    u32x2 r;
    r[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
    r[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
    return r;
}


template <int N, int ROW8>
inline void ds_read_tr16_batch(const void* lds_ptr, u32x4* out) {
    for (int i = 0; i < N; ++i) {
        const u32x2 lo = ds_read_tr16_b64((const char*)lds_ptr + i * 64);
        const u32x2 hi = ds_read_tr16_b64((const char*)lds_ptr + i * 64 + ROW8);
        out[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}

template <int N, int HI>
inline void tr16_issue(const void* base, const int (&off)[N], int imm, u32x2 (&lo)[N], u32x2 (&hi)[N]) {
    for (int i = 0; i < N; ++i) {
        lo[i] = ds_read_tr16_b64((const char*)base + imm + off[i]);
        hi[i] = ds_read_tr16_b64((const char*)base + imm + off[i] + HI);
    }
}
template <int CNT, int N>
inline void lgkm_fence(u32x2 (&)[N], u32x2 (&)[N]) {}

template <typename S>
inline S emu_shfl_idx(S v, int src) {
    S* s = (S*)hipemu::wave_buf();
    s[lane_id()] = v;
    hipemu::wave_sync();
    S r = s[src & 63];
    hipemu::wave_sync();
    return r;
}
inline int shfl_idx(int v, int src) { return emu_shfl_idx(v, src); }
inline float shfl_xor(float v, int m) { return emu_shfl_idx(v, lane_id() ^ m); }
inline int shfl_xor(int v, int m) { return emu_shfl_idx(v, lane_id() ^ m); }
inline void swap_hi_lo(unsigned& a, unsigned& b) {
    const unsigned ta = (unsigned)emu_shfl_idx((int)a, lane_id() ^ 32), tb = (unsigned)emu_shfl_idx((int)b, lane_id() ^ 32);
    if (lane_id() < 32) b = ta; else a = tb;
}
inline float quad_max(float v) {
    v = fmaxf(v, emu_shfl_idx(v, lane_id() ^ 1));
    return fmaxf(v, emu_shfl_idx(v, lane_id() ^ 2));
}
template <int SRC> inline unsigned quad_bcast(unsigned v) { return (unsigned)emu_shfl_idx((int)v, (lane_id() & ~3) | SRC); }
inline void lmi_trap() { __builtin_trap(); }
inline bool wave_any(bool p) {
    int v = p ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= emu_shfl_idx(v, lane_id() ^ m);
    return v != 0;
}
inline void sched_fence() {}
inline void setprio_hi() {}
inline void setprio_lo() {}
inline float fast_exp2(float x) { return exp2f(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
inline float fexp(float x) { return expf(x); }
inline float mul_rn(float a, float b) { return a * b; }
inline float sub_rn(float a, float b) { return a - b; }
inline float sep_rn(float v) { return v; }

#define LMI_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipemu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })

#endif  // LMI_EMU

// ------------------------------------------------------------------------------------------------
// shared helpers
// ------------------------------------------------------------------------------------------------
// One-shot streamed operand (a decode step's weights: every byte is read once by one CU): non-temporal policy, so that the stream
// does not push the activations / partial slabs the next launches re-read out of L2 / MALL.  LMI_STREAM_NT=0 builds the default policy
// (A/B only).
#ifndef LMI_STREAM_NT
#define LMI_STREAM_NT 1
#endif
template <typename V> LMI_DEV V ld_stream(const V* p) {
#if !defined(LMI_EMU) && LMI_STREAM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
// GEMM epilogue traffic with a non-temporal hint, by class (LMI_EPI_NT bits; compile-time, so that an A/B is two builds of the library
// on one box — LEOPARD_AMD_LIB): bit 1 = the fp32 residual read-modify-write, bit 2 = the 16-bit / fp8 outputs of the store / GELU /
// SwiGLU epilogues (hundreds of MB per launch, written once: without the hint they are write-allocated in L2 over the operand tiles
// the other workgroups are re-reading).  Measured on the C3 step, two boxes, two rounds each: bit 1 +-0.0 %, bit 2 -0.3 ... -0.6 %
// (production), the same hint on the q | k | v rows / the attention output +0.4 % (the next launch wants them cached), on the prefill's
// KV-cache append +-0.0 %.
#ifndef LMI_EPI_NT
#define LMI_EPI_NT 2
#endif
template <int BIT, typename V> LMI_DEV void st_epi(V* p, V v) {
#if !defined(LMI_EMU)
    if (LMI_EPI_NT & BIT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}
template <int BIT, typename V> LMI_DEV V ld_epi(const V* p) {
#if !defined(LMI_EMU)
    if (LMI_EPI_NT & BIT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
// ---- fp4 e2m1 (the low-bit correction operands) ------------------------------------------------------------------------------------------
// |x| on the grid {0, .5, 1, 1.5, 2, 3, 4, 6} (codes 0..7), sign in bit 3; round to nearest, ties to the even code, saturating at 6.
// Software on purpose: the same bits on the device and in the host emulator (the hardware conversion v_cvt_scalef32_pk_fp4_f32 is
// compared with it by tools/ubench/mfma_fp4_layout.hip); a few VALU operations per element in epilogues that are HBM- or latency-bound.
LMI_DEV unsigned to_fp4(float x) {
    const unsigned s = (__builtin_bit_cast(unsigned, x) >> 28) & 8u;
    const float a = fminf(__builtin_fabsf(x), 6.0f);
    // units of the binade's grid step: < 2 -> halves, < 4 -> ones, else twos; rintf = round half to even, and an even count is an even code
    const float q = a < 2.0f ? __builtin_rintf(a * 2.0f) : (a < 4.0f ? 2.0f + __builtin_rintf(a) : 4.0f + __builtin_rintf(a * 0.5f));
    return s | (unsigned)(int)q;
}
// MX block scale of a block whose largest magnitude is `amax`: E8M0 byte of 2^(floor(log2 amax) - 2) (the block maximum lands in [4, 8):
// values above 6 saturate), that power of two and its exact reciprocal.  amax = 0 (or a denormal): byte 0, scale 1, inv 0 — every code is 0.
LMI_DEV unsigned lo4_scale_byte(float amax, float& scale, float& inv) {
    const int eb = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 255u) - 2;
    if (eb < 1) { scale = 1.0f; inv = 0.f; return 0u; }
    scale = __builtin_bit_cast(float, (unsigned)eb << 23);
    inv = __builtin_bit_cast(float, (unsigned)(254 - eb) << 23);
    return (unsigned)eb;
}
#ifdef LMI_EMU
inline unsigned fp4_pack8(const float (&v)[8], float, float inv) {
    unsigned r = 0;
    for (int e = 0; e < 8; ++e) r |= to_fp4(v[e] * inv) << (4 * e);
    return r;
}
#endif
// One lane's share of a low-bit residual image: y[0..7] are 8 consecutive fp32 values of a row, the 4 lanes of a lane quad hold one
// 32-element block.  Returns the 8 codes of the lane (element e in nibble e) for the residuals y - float(T(y)) and, in every lane of the
// quad, the block's E8M0 scale byte.  hi[e] = T(y[e]) is what the 16-bit pass multiplies.
template <typename T>
LMI_DEV unsigned lo4_encode8(const float (&y)[8], typename vec_of<T>::x8& hi, unsigned& scale_byte) {
    float lo[8], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float ye = sep_rn(y[e]);                               // the value is rounded on its own (see sep_rn), whatever produced it
        hi[e] = (T)ye;
        lo[e] = ye - (float)hi[e];
        amax = fmaxf(amax, __builtin_fabsf(lo[e]));
    }
    amax = quad_max(amax);
    float scale, inv;
    scale_byte = lo4_scale_byte(amax, scale, inv);
    return fp4_pack8(lo, scale, inv);
}

LMI_DEV int imin(int a, int b) { return a < b ? a : b; }
LMI_DEV int imax(int a, int b) { return a > b ? a : b; }
template <typename T> LMI_DEV float to_f32(T v) { return (float)v; }
template <typename T> LMI_DEV T from_f32(float v) { return (T)v; }
// element of an 8-wide output vector from an fp32 value: a cast for the 16-bit / fp32 types, the saturating e4m3 conversion for fp8
template <typename T> struct OutCvt { static LMI_DEV T cvt(float v) { return (T)v; } };
template <> struct OutCvt<fp8_t> { static LMI_DEV uint8_t cvt(float v) { return to_fp8(v); } };
// two fp32 -> one dword of two T (x in the low half)
template <typename T> LMI_DEV unsigned pack2(float x, float y) {
    typename vec_of<T>::x2 v;
    v[0] = (T)x; v[1] = (T)y;
    return __builtin_bit_cast(unsigned, v);
}
// reductions across the two half-wave partners (lane, lane ^ 32): both lanes get the result
LMI_DEV float xhalf_max(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    swap_hi_lo(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
LMI_DEV float xhalf_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    swap_hi_lo(a, b);
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

LMI_DEV float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
LMI_DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

}  // namespace lmi

// ---- optional in-kernel cycle accounting (builds with -DLMI_ATTN_PROF only; tools/attn_prof.py) -----------------------
#if defined(LMI_ATTN_PROF) && !defined(LMI_EMU)
namespace lmi { extern __device__ unsigned long long* g_prof_buf; }
#define LMI_PROF_DECL() unsigned long long prof_t[7] = {0, 0, 0, 0, 0, 0, 0}, prof_last = 0, prof_now = 0; int prof_sink = 0
#define LMI_PROF_MARK(i) do { prof_now = __builtin_amdgcn_s_memtime(); if ((i) != 0) prof_t[i] += prof_now - prof_last; prof_last = prof_now; } while (0)
#define LMI_PROF_TOUCH(x) (prof_sink += __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)(x))))
#define LMI_PROF_DUMP() do { if ((threadIdx.x & 63) == 0 && lmi::g_prof_buf) { \
        unsigned long long* d = lmi::g_prof_buf + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; \
        for (int i = 0; i < 7; ++i) d[i] = prof_t[i]; d[7] = (unsigned long long)(unsigned)prof_sink; } } while (0)
#else
#define LMI_PROF_DECL() ((void)0)
#define LMI_PROF_MARK(i) ((void)0)
#define LMI_PROF_TOUCH(x) ((void)0)
#define LMI_PROF_DUMP() ((void)0)
#endif
