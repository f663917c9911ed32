// elementwise.h — the HBM-bound kernels of the prefill path (SURVEY.md 2.5 "Bound by: HBM" rows):
// tile normalisation + im2col, LayerNorm / RMSNorm over the fp32 residual stream, RoPE (+ KV-cache write),
// embedding gather + image/text merge, last-token lm_head GEMV, synthetic parameter fill.
// All of them move 16 bytes per lane where the layout allows and reduce with wave64 shuffles.
#pragma once
#include "lmi_device.h"

namespace lmi {

// ------------------------------------------------------------------------------------------------
// synthetic fill: element i of tensor `seed` (see leopard_amd/synth.py — same integer hash, bit-exact)
// ------------------------------------------------------------------------------------------------
LMI_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
LMI_DEV float synth_value(uint32_t seed, uint64_t i, int kind) {
    uint32_t h = mix32((uint32_t)i ^ seed);
    h = mix32(h + (uint32_t)(i >> 32) * 0x9E3779B9u + 0x85EBCA6Bu);
    const float b = (float)(h >> 24);
    if (kind == 0) return (2.0f * b - 255.0f) * 0x1p-13f;
    if (kind == 1) return (2.0f * b - 255.0f) * 0x1p-15f;
    return (112.0f + floorf(b * 0.125f)) * 0x1p-7f;
}
template <typename OUT>
__global__ void fill_synth_kernel(OUT* out, long n, uint32_t seed, int kind) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = (OUT)synth_value(seed, (uint64_t)i, kind);
}

// ------------------------------------------------------------------------------------------------
// a5: u8 tiles [N,S,S,3] (HWC) or fp32 pixel_values [N,3,S,S] -> normalised im2col rows
//     out[(n*G + py)*G + px][c*P*P + ky*P + kx]  (K padded with zeros to ldo)
//     value law = SiglipImageProcessor: x*(1/255) then (x-0.5)/0.5   (EVAL:403-404)
// ------------------------------------------------------------------------------------------------
template <typename T, bool FROM_U8>
__global__ void im2col_kernel(const void* in, T* out, int n_tiles, int H, int W, int P, int ldo) {
    const int GH = H / P, GW = W / P, KD = 3 * P * P;         // valid conv: remainder pixels are dropped
    const long total = (long)n_tiles * GH * GW * ldo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % ldo);
        const long row = i / ldo;
        float v = 0.f;
        if (col < KD) {
            const int c = col / (P * P), rem = col - c * P * P, ky = rem / P, kx = rem - ky * P;
            const int px = (int)(row % GW), py = (int)((row / GW) % GH);
            const long n = row / ((long)GH * GW);
            const int y = py * P + ky, x = px * P + kx;
            if (FROM_U8) {
                const float u = (float)((const uint8_t*)in)[((n * H + y) * W + x) * 3 + c];
                v = mul_rn(sub_rn(mul_rn(u, 1.0f / 255.0f), 0.5f), 2.0f);     // no FMA contraction: bit-exact vs the processor
            } else {
                v = ((const float*)in)[((n * 3 + c) * H + y) * (long)W + x];
            }
        }
        out[i] = (T)v;
    }
}

// ------------------------------------------------------------------------------------------------
// K / V rows -> cache rows [pos0, pos0 + S): the cache update of EVAL:291-320 without a rotation (rows that are already
// rotated: moving a packed batch's K/V from the pooled prefill cache into per-sample caches, chunked TP prefill)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void kv_append_kernel(const T* k, const T* v, T* k_cache, T* v_cache, int S, int width, int ld_src, int ld_cache, int pos0) {
    typedef typename vec_of<T>::x8 T8;
    const int cpr = width >> 3;
    const long total = (long)S * cpr * 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int which = (int)(i / ((long)S * cpr));
        const long j = i - (long)which * S * cpr;
        const int s = (int)(j / cpr), c = (int)(j - (long)s * cpr);
        const T* src = (which ? v : k) + (long)s * ld_src + c * 8;
        T* dst = (which ? v_cache : k_cache) + (long)(pos0 + s) * ld_cache + c * 8;
        *(T8*)dst = *(const T8*)src;
    }
}

// ------------------------------------------------------------------------------------------------
// One pass of Pillow's antialiased resampling for 8-bit RGB (a3 / a5: Image.resize inside resize_and_pad_image,
// EVAL:102-140, and the thumbnail squash of SiglipImageProcessor, EVAL:403-404).  Bit-identical to libImaging
// ImagingResampleHorizontal_8bpc / Vertical_8bpc: 22-bit fixed-point taps from the host (leopard_amd/tiler.py
// pil_resample_coeffs, a restatement of precompute_coeffs + normalize_coeffs_8bpc), int32 accumulation from 2^21,
// arithmetic shift, clamp to [0, 255].  AXIS 0: along a row (src [rows, in, 3] -> dst [rows, out, 3]); AXIS 1: down a
// column (src [in, cols, 3] -> dst [out, cols, 3]).  HBM-bound byte work; one thread per output pixel.
// ------------------------------------------------------------------------------------------------
template <int AXIS>
__global__ void resample_u8_kernel(const uint8_t* src, uint8_t* dst, int rows, int cols, int src_pitch, int dst_pitch,
                                   const int* bounds, const int* kk, int ksize) {
    // rows x cols = extent of the OUTPUT image
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / cols), x = (int)(i - (long)y * cols);
        const int o = AXIS == 0 ? x : y;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (long)o * ksize;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        const uint8_t* p = AXIS == 0 ? src + (long)y * src_pitch + (long)lo * 3 : src + (long)lo * src_pitch + (long)x * 3;
        const long step = AXIS == 0 ? 3 : src_pitch;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            s0 += (int)p[0] * w; s1 += (int)p[1] * w; s2 += (int)p[2] * w;
            p += step;
        }
        uint8_t* d = dst + (long)y * dst_pitch + (long)x * 3;
        d[0] = (uint8_t)imin(imax(s0 >> 22, 0), 255);
        d[1] = (uint8_t)imin(imax(s1 >> 22, 0), 255);
        d[2] = (uint8_t)imin(imax(s2 >> 22, 0), 255);
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (SigLIP, eps 1e-6) / RMSNorm (Llama, eps 1e-5): fp32 stream row -> 16-bit GEMM operand row.
// One wave per row, row kept in registers, statistics in fp32 with two passes (mean, then centred variance).
// ------------------------------------------------------------------------------------------------
// LO4 (lmi_norm_lo4): besides T(y) the kernel writes the MX fp4 image of the rounding residuals y - T(y) and its block scales (the A
// operand of a GEMM with the low-bit correction phase, lowbit.h); a lane's 8 elements are a quarter of a 32-element block, D % 32 == 0,
// images K4 = D rounded up to 256 wide (zero padding written here).
struct NormLo4 {
    uint8_t* out4;
    uint8_t* scales;
    int ld4, lds, K4;
    const uint8_t* row_sel;      // [M] or null: rows whose image is wanted (gemm.h GemmArgs::row_sel); the others get T(y) only
};
template <typename T, bool RMS, int MAXV, bool LO4 = false>     // MAXV = max 8-element chunks per lane (D <= MAXV*512)
__global__ void __launch_bounds__(256) norm_kernel(const float* x, const float* w, const float* b, T* out,
                                                   int M, int D, int ldx, int ldo, float eps, float out_scale, NormLo4 lo = NormLo4()) {
    typedef typename vec_of<T>::x8 T8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * ldx;
    const int nvec = D >> 3;                         // 8-element chunks in the row (D % 8 == 0): 32 B in, 16 B out per lane
    f32x4 v[MAXV][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            v[i][0] = *(const f32x4*)(xr + c * 8);
            v[i][1] = *(const f32x4*)(xr + c * 8 + 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                s += RMS ? (v[i][h][0] * v[i][h][0] + v[i][h][1] * v[i][h][1] + v[i][h][2] * v[i][h][2] + v[i][h][3] * v[i][h][3])
                         : (v[i][h][0] + v[i][h][1] + v[i][h][2] + v[i][h][3]);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = 1.0f / sqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nvec) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][h][e] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd = 1.0f / sqrtf(q / (float)D + eps);
    }
    T* orow = out + (long)row * ldo;
    const bool want4 = LO4 && (!lo.row_sel || lo.row_sel[row]);      // (wave-uniform: one wave per row) this row hands over a residual image
    float lo_y[LO4 ? MAXV : 1][8];                                   // LO4: the normalised values, encoded wave-uniformly below
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            T8 o;
            float yv[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ww = *(const f32x4*)(w + c * 8 + 4 * h);
                if (RMS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = ww[e] * (v[i][h][e] * rstd);
                        yv[4 * h + e] = y;
                        o[4 * h + e] = OutCvt<T>::cvt(sizeof(T) == 1 ? y * out_scale : sep_rn(y));       // fp8 operand: static power-of-two scale
                    }
                } else {
                    const f32x4 bb = *(const f32x4*)(b + c * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (v[i][h][e] - mean) * rstd * ww[e] + bb[e];
                        yv[4 * h + e] = y;
                        o[4 * h + e] = OutCvt<T>::cvt(sizeof(T) == 1 ? y * out_scale : sep_rn(y));
                    }
                }
            }
            if constexpr (!(LO4 && sizeof(T) == 2)) *(T8*)(orow + c * 8) = o;
            if constexpr (LO4 && sizeof(T) == 2) {
                if (!want4) *(T8*)(orow + c * 8) = o;                 // the same T(y) lo4_encode8 returns below
#pragma unroll
                for (int e = 0; e < 8; ++e) lo_y[i][e] = yv[e];
            }
        }
    }
    if constexpr (LO4 && sizeof(T) == 2) {
        if (!want4) return;
        // every lane of the wave takes part in the quad exchange (idle lanes on zeros: they also write the zero padding up to K4);
        // nvec % 4 == 0, so a block's four lanes are live or idle together
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (i * 64 >= (lo.K4 >> 3)) break;                       // wave-uniform
            float yv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) yv[e] = c < nvec ? lo_y[i][e] : 0.f;
            T8 o;
            unsigned sb;
            const unsigned codes = lo4_encode8<T>(yv, o, sb);
            if (c < nvec) *(T8*)(orow + c * 8) = o;
            if (c < (lo.K4 >> 3)) {
                *(unsigned*)(lo.out4 + (long)row * lo.ld4 + c * 4) = codes;
                if ((lane & 3) == 0) lo.scales[(long)row * lo.lds + (c >> 2)] = (uint8_t)sb;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same norms for a HANDFUL of rows (decode: 1 .. 16 rows): one 256-thread workgroup per row instead of one wave, so that a row's
// 16 KiB arrive through 4 waves' worth of loads in flight (the one-wave form is latency-bound at ~7.6 us for 8 rows; the batched decode
// step runs 64 of them).  Statistics: per-lane partial -> wave butterfly -> 4 wave sums added in wave order (deterministic).
// ------------------------------------------------------------------------------------------------
template <typename T, bool RMS, int MAXV>     // MAXV = max 8-element chunks per thread (D <= MAXV * 2048)
__global__ void __launch_bounds__(256) norm_rows_kernel(const float* x, const float* w, const float* b, T* out,
                                                        int M, int D, int ldx, int ldo, float eps, float out_scale) {
    typedef typename vec_of<T>::x8 T8;
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float* xr = x + (long)row * ldx;
    const int nvec = D >> 3;
    f32x4 v[MAXV][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + tid;
        if (c < nvec) {
            v[i][0] = *(const f32x4*)(xr + c * 8);
            v[i][1] = *(const f32x4*)(xr + c * 8 + 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                s += RMS ? (v[i][h][0] * v[i][h][0] + v[i][h][1] * v[i][h][1] + v[i][h][2] * v[i][h][2] + v[i][h][3] * v[i][h][3])
                         : (v[i][h][0] + v[i][h][1] + v[i][h][2] + v[i][h][3]);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = 1.0f / sqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + tid;
            if (c < nvec) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][h][e] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        if (lane == 0) red[1][wave] = q;
        __syncthreads();
        q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        rstd = 1.0f / sqrtf(q / (float)D + eps);
    }
    T* orow = out + (long)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + tid;
        if (c < nvec) {
            T8 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ww = *(const f32x4*)(w + c * 8 + 4 * h);
                if (RMS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = ww[e] * (v[i][h][e] * rstd);
                        o[4 * h + e] = OutCvt<T>::cvt(sizeof(T) == 1 ? y * out_scale : sep_rn(y));
                    }
                } else {
                    const f32x4 bb = *(const f32x4*)(b + c * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (v[i][h][e] - mean) * rstd * ww[e] + bb[e];
                        o[4 * h + e] = OutCvt<T>::cvt(sizeof(T) == 1 ? y * out_scale : sep_rn(y));
                    }
                }
            }
            *(T8*)(orow + c * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sequence-parallel residual update (tensor-parallel LLM): x[row] += delta[row] (the reduce-scattered partial
// products of o_proj / down_proj, 16-bit or fp32), x written back, then out[row] = w * (x * rstd) as norm_kernel<RMS> does.
// out == nullptr: the add only.  One wave per row, row in registers.
// ------------------------------------------------------------------------------------------------
template <typename T, typename DT, int MAXV, bool LO4 = false>
__global__ void __launch_bounds__(256) add_rmsnorm_kernel(float* x, const DT* delta, const float* w, T* out, int M, int D, int ldx,
                                                          int ldd, int ldo, float eps, NormLo4 lo = NormLo4()) {
    typedef typename vec_of<T>::x8 T8;
    typedef typename vec_of<DT>::x8 D8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = x + (long)row * ldx;
    const DT* dr = delta + (long)row * ldd;
    const int nvec = D >> 3;
    f32x4 v[MAXV][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            const D8 d = *(const D8*)(dr + c * 8);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[i][h] = *(const f32x4*)(xr + c * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][h][e] += (float)d[4 * h + e];
                *(f32x4*)(xr + c * 8 + 4 * h) = v[i][h];
                s += v[i][h][0] * v[i][h][0] + v[i][h][1] * v[i][h][1] + v[i][h][2] * v[i][h][2] + v[i][h][3] * v[i][h][3];
            }
        }
    }
    if (!out) return;                                    // wave-uniform
    s = wave_sum(s);
    const float rstd = 1.0f / sqrtf(s / (float)D + eps);
    T* orow = out + (long)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            T8 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 ww = *(const f32x4*)(w + c * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = ww[e] * (v[i][h][e] * rstd);
                    o[4 * h + e] = (T)y;
                    if constexpr (LO4) v[i][h][e] = y;               // kept for the residual image below
                }
            }
            if constexpr (!LO4) *(T8*)(orow + c * 8) = o;
        }
    }
    if constexpr (LO4) {                                             // as norm_kernel: every lane of the wave takes part in the quad exchange
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 64 + lane;
            if (i * 64 >= (lo.K4 >> 3)) break;
            float yv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) yv[e] = c < nvec ? v[i][e >> 2][e & 3] : 0.f;
            T8 o;
            unsigned sb;
            const unsigned codes = lo4_encode8<T>(yv, o, sb);
            if (c < nvec) *(T8*)(orow + c * 8) = o;
            if (c < (lo.K4 >> 3)) {
                *(unsigned*)(lo.out4 + (long)row * lo.ld4 + c * 4) = codes;
                if ((lane & 3) == 0) lo.scales[(long)row * lo.lds + (c >> 2)] = (uint8_t)sb;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE (rotate-half, cos/sin tables [S, D/2] fp32 built by the host from position_ids and the llama3-scaled
// inverse frequencies) applied in place to the q and k heads of the packed qkv rows; optionally writes the
// rotated K and the V rows into the KV cache [S_cache, n_kv*D] at row cache_pos0 + s.
// One thread = one 8-element chunk d..d+7 of the first half and its partner chunk d+D/2..  (D = 128)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_kernel(T* qkv, int S, int ld, int n_q, int n_kv, int D, const float* cosT, const float* sinT,
                            T* k_cache, T* v_cache, int ld_cache, int cache_pos0, const int* pos_dev) {
    // pos_dev (decode under a captured graph): the position of row 0 lives in device memory; it indexes the cos/sin
    // tables (built for every position of the cache) and the cache row alike
    const int table0 = pos_dev ? *pos_dev : 0;
    if (pos_dev) cache_pos0 = table0;
    typedef typename vec_of<T>::x8 T8;
    const int half = D >> 1, cpr = half >> 3;                 // chunks per half row
    const int rot_heads = n_q + n_kv;
    const int per_tok = rot_heads * cpr + (v_cache ? n_kv * (D >> 3) : 0);
    const long total = (long)S * per_tok;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int s = (int)(i / per_tok);
        int w = (int)(i - (long)s * per_tok);
        T* row = qkv + (long)s * ld;
        if (w < rot_heads * cpr) {
            const int h = w / cpr, c = w - h * cpr;
            T* p1 = row + h * D + c * 8;
            T* p2 = p1 + half;
            const T8 a = *(const T8*)p1, b = *(const T8*)p2;
            const float* cs = cosT + (long)(table0 + s) * half + c * 8;
            const float* sn = sinT + (long)(table0 + s) * half + c * 8;
            T8 oa, ob;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = (float)a[e], x2 = (float)b[e];
                oa[e] = (T)(x1 * cs[e] - x2 * sn[e]);
                ob[e] = (T)(x2 * cs[e] + x1 * sn[e]);
            }
            *(T8*)p1 = oa;
            *(T8*)p2 = ob;
            if (k_cache && h >= n_q) {
                T* kc = k_cache + (long)(cache_pos0 + s) * ld_cache + (h - n_q) * D + c * 8;
                *(T8*)kc = oa;
                *(T8*)(kc + half) = ob;
            }
        } else {
            w -= rot_heads * cpr;
            const int h = w / (D >> 3), c = w - h * (D >> 3);
            const T8 a = *(const T8*)(row + (n_q + n_kv + h) * D + c * 8);
            *(T8*)(v_cache + (long)(cache_pos0 + s) * ld_cache + h * D + c * 8) = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a10: embedding gather + image/text merge into the fp32 residual stream.
//   src[s] >= 0 : row = embed_table[ids[src[s]]]   (16-bit -> fp32)
//   src[s] <  0 : row = visual_tokens[-src[s]-1]   (fp32 projector output)
// one workgroup per output row
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_merge_kernel(const long* ids, const long* src, const T* table, const float* feats,
                                   float* out, int D, int ld_feats) {
    typedef typename vec_of<T>::x8 T8;
    const long s = blockIdx.x;
    const long sv = src[s];
    float* o = out + s * D;
    if (sv >= 0) {
        const T* t = table + ids[sv] * (long)D;
        for (int c = threadIdx.x; c < (D >> 3); c += blockDim.x) {
            const T8 a = *(const T8*)(t + c * 8);
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = (float)a[e]; hi[e] = (float)a[e + 4]; }
            *(f32x4*)(o + c * 8) = lo;
            *(f32x4*)(o + c * 8 + 4) = hi;
        }
    } else {
        const float* f = feats + (-sv - 1) * (long)ld_feats;
        for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) *(f32x4*)(o + c * 4) = *(const f32x4*)(f + c * 4);
    }
}

// ------------------------------------------------------------------------------------------------
// The tail of a greedy decode step (EVAL:448-452: argmax, stop at eos / max_new_tokens), for B sequences at once and entirely in
// device memory, so that it sits inside the captured step: one workgroup per sequence takes the argmax of its logits row (lowest index on
// ties), writes the token, records it in the sequence's history ring, and applies the stop rule — a sequence that produced an eos id or
// used up its token budget stops advancing (live = 0: its position and key count freeze; what it produces afterwards is ignored).
// ------------------------------------------------------------------------------------------------
struct DecodeAdvanceArgs {
    const float* logits;      // [B, ld_logits] fp32
    int vocab, ld_logits;
    const int64_t* suppress;  // optional: token ids that may never be chosen
    int n_suppress;
    int64_t* tok;             // [B] out: the chosen token
    int* pos;                 // [B] += live
    int* k_len;               // [B] += live (nullable)
    int* live;                // [B] nullable: 1 = running (null: always running)
    int* budget;              // [B] nullable: tokens the sequence may still produce
    const int64_t* eos;       // [n_eos] (entries < 0 are unused)
    int n_eos;
    int64_t* hist;            // [hist_len, B] nullable: hist[hist_pos[b] % hist_len][b] = tok
    int* hist_pos;            // [B] per-sequence step counter of the ring
    int hist_len, B;
};
__global__ void __launch_bounds__(1024) decode_advance_kernel(DecodeAdvanceArgs a) {
    __shared__ float best_v[16];
    __shared__ int best_i[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = a.logits + (long)b * a.ld_logits;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > bv || (v == bv && i < bi)) {                         // only a candidate for the running best is looked up in the suppress list
            for (int j = 0; j < a.n_suppress; ++j)
                if (a.suppress[j] == i) return;                      // (the logits row itself is left as computed: suppressed ids are skipped, not overwritten)
            bv = v; bi = i;
        }
    };
    // 16 bytes per lane, four loads in flight per thread: the row is 0.5 MB and ONE workgroup scans it — the scan is load latency, not bandwidth
    const int v4 = ((((size_t)row) & 15) == 0) ? (a.vocab >> 2) : 0;
    for (int i0 = tid; i0 < v4; i0 += 4 * (int)blockDim.x) {
        f32x4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            q[u] = i < v4 ? *(const f32x4*)(row + 4 * (long)i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < v4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) take(q[u][e], 4 * i + e);
            }
        }
    }
    for (int i = 4 * v4 + tid; i < a.vocab; i += blockDim.x) take(row[i], i);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = shfl_xor(bv, m);
        const int oi = shfl_xor(bi, m);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { best_v[wave] = bv; best_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int w = 1; w < nw; ++w)
            if (best_v[w] > bv || (best_v[w] == bv && best_i[w] < bi)) { bv = best_v[w]; bi = best_i[w]; }
        if (bi == 0x7fffffff) bi = 0;                               // a row of NaN / -inf only
        a.tok[b] = bi;
        int lv = a.live ? a.live[b] : 1;
        if (a.hist) {
            const int hp = a.hist_pos[b];
            a.hist[(long)(hp % a.hist_len) * a.B + b] = bi;
            a.hist_pos[b] = hp + 1;
        }
        int bud = 1;
        if (a.budget) { bud = a.budget[b] - lv; a.budget[b] = bud; }
        bool stop = bud <= 0;
        for (int j = 0; j < a.n_eos; ++j) stop = stop || (a.eos[j] == (int64_t)bi);
        if (stop) lv = 0;
        if (a.live) a.live[b] = lv;
        a.pos[b] += lv;
        if (a.k_len) a.k_len[b] += lv;
    }
}

// ------------------------------------------------------------------------------------------------
// weight-streaming GEMV (M = 1): out[n] = W[n,:] . x  — last-token lm_head and the decode step.
// One wave per output row (per gate/up row pair for SwiGLU); x is held in registers; W streams 16 B / lane.
// ------------------------------------------------------------------------------------------------
enum { GEMV_STORE_F32 = 0, GEMV_STORE_T = 1, GEMV_RESID_F32 = 2, GEMV_SWIGLU_T = 3, GEMV_QKV_ROPE_T = 4 };

// RoPE + KV append riding in a decode projection's epilogue (gemv_split_kernel GEMV_QKV_ROPE_T, skinny_gemm_kernel SK_QKV_ROPE_T):
// the q / k weight rows are in weights.rope_permute_rows order, so rows r and r + 32 of a 64-row group — the SwiGLU pairing — hold
// a first-half element and its rotate-half partner.
struct RopeEpi {
    const float* cos_all;     // [capacity, 64]
    const float* sin_all;
    const int* pos;           // [M] (one entry for the batch-1 GEMV)
    void* k_cache;
    void* v_cache;
    int ld_cache;
    long cache_stride;
    int rope_q, rope_k;       // columns [0, rope_q) are q heads, [rope_q, rope_q + rope_k) k heads, the rest v
};


template <typename T, int EPI, int KCH>       // KCH = 16-byte chunks per lane (K = KCH*512)
__global__ void __launch_bounds__(256) gemv_kernel(const T* W, const T* x, const float* bias, void* out,
                                                   int N, int K, int ldw) {
    typedef typename vec_of<T>::x8 T8;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    T8 xv[KCH];
#pragma unroll
    for (int i = 0; i < KCH; ++i) {
        const int c = i * 64 + lane;
        if (c * 8 < K) xv[i] = *(const T8*)(x + c * 8);
    }
    auto dot = [&](const T* wrow) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int c = i * 64 + lane;
            if (c * 8 < K) {
                const T8 wv = ld_stream((const T8*)(wrow + c * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)wv[e] * (float)xv[i][e];
            }
        }
        return wave_sum(acc);
    };
    if (EPI == GEMV_SWIGLU_T) {
        // rows interleaved in 32-row blocks [gate | up]: output j pairs rows 64*(j/32) + j%32 and +32
        const int n_out = N >> 1;
        for (int j = wave; j < n_out; j += nwaves) {
            const int r0 = ((j >> 5) << 6) + (j & 31);
            const float g = dot(W + (long)r0 * ldw), u = dot(W + (long)(r0 + 32) * ldw);
            if (lane == 0) ((T*)out)[j] = (T)(g / (1.0f + lmi::fexp(-g)) * u);
        }
    } else {
        for (int n = wave; n < N; n += nwaves) {
            float v = dot(W + (long)n * ldw);
            if (lane == 0) {
                if (bias) v += bias[n];
                if (EPI == GEMV_STORE_F32) ((float*)out)[n] = v;
                else if (EPI == GEMV_STORE_T) ((T*)out)[n] = (T)v;
                else ((float*)out)[n] += v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decode GEMV, K split across the 4 waves of a workgroup (hidden sizes K = CPW*2048: 4096, 14336).
//   * each wave keeps only its quarter of x in registers (read once per workgroup, not once per output row) and streams
//     the matching quarter of R weight rows at a time (R*CPW 16-byte loads in flight per lane);
//   * NORM: x arrives as the fp32 residual row and the RMSNorm (same arithmetic as norm_kernel) is applied while it is
//     loaded, so the decode step needs no separate norm launch;
//   * per-row partial sums of the 4 waves meet in LDS once per workgroup, then the epilogue runs.
// Epilogues as gemv_kernel, plus GEMV_QKV_ROPE_T.  `units` = outputs (row pairs for SwiGLU / RoPE); workgroup b owns units
// [b*UPB, (b+1)*UPB); the launcher sizes the grid to a whole number of equal workgroups per CU when the shape allows.
// Measured and dropped (round 3, same box): requesting the next pass's rows before reducing the current one (two register buffers)
// and the first pass's rows before the norm arithmetic: +1.5 % on the decode step — occupancy (3 workgroups per CU) already keeps the
// memory queues full, the extra registers and the raw-barrier prologue only cost.
// ------------------------------------------------------------------------------------------------
template <typename T, int EPI, int CPW, int R, bool NORM>
__global__ void __launch_bounds__(256) gemv_split_kernel(const T* W, const void* xin, const float* gamma, float eps, const float* bias,
                                                         void* out, int N, int K, int ldw, int UPB, RopeEpi rp) {
    typedef typename vec_of<T>::x8 T8;
    constexpr bool PAIR = (EPI == GEMV_SWIGLU_T || EPI == GEMV_QKV_ROPE_T);
    constexpr int RW = PAIR ? 2 : 1;                              // weight rows per unit
    __shared__ float part[4][128];
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int units = PAIR ? N >> 1 : N;
    const int u0 = blockIdx.x * UPB, u1 = imin(units, u0 + UPB);
    // ---- this wave's slice of x ---------------------------------------------------------------------------------
    T8 xv[CPW];
    if (NORM) {
        const float* xf = (const float*)xin;
        f32x4 lo[CPW], hi[CPW];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int e = ((wave * CPW + i) * 64 + lane) * 8;
            lo[i] = *(const f32x4*)(xf + e);
            hi[i] = *(const f32x4*)(xf + e + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) ss += lo[i][j] * lo[i][j] + hi[i][j] * hi[i][j];
        }
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int e = ((wave * CPW + i) * 64 + lane) * 8;
            const f32x4 g0 = *(const f32x4*)(gamma + e), g1 = *(const f32x4*)(gamma + e + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[i][j] = (T)(g0[j] * (lo[i][j] * rstd));
                xv[i][j + 4] = (T)(g1[j] * (hi[i][j] * rstd));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPW; ++i) xv[i] = *(const T8*)((const T*)xin + ((wave * CPW + i) * 64 + lane) * 8);
    }
    // ---- stream the weight rows, R at a time -------------------------------------------------------------------
    auto row_of = [&](int u, int half) -> int {                    // weight row of output u (SwiGLU: gate / up interleaved by 32)
        if (PAIR) return ((u >> 5) << 6) + (u & 31) + 32 * half;
        return u;
    };
    for (int ub = u0; ub < u1; ub += R / RW) {
        T8 wv[R][CPW];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int u = imin(ub + r / RW, units - 1);
            const T* wrow = W + (long)row_of(u, r % RW) * ldw;
#pragma unroll
            for (int i = 0; i < CPW; ++i) wv[r][i] = ld_stream((const T8*)(wrow + ((wave * CPW + i) * 64 + lane) * 8));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < CPW; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)wv[r][i][e] * (float)xv[i][e];
            acc = wave_sum(acc);
            const int slot = (ub - u0) * RW + r;
            if (lane == 0 && slot < 128) part[wave][slot] = acc;
        }
    }
    __syncthreads();
    // ---- epilogue: one thread per output of the workgroup ----------------------------------------------------
    const int t = threadIdx.x;
    if (t < u1 - u0) {
        const int u = u0 + t;
        if (EPI == GEMV_SWIGLU_T) {
            const float g = part[0][2 * t] + part[1][2 * t] + part[2][2 * t] + part[3][2 * t];
            const float up = part[0][2 * t + 1] + part[1][2 * t + 1] + part[2][2 * t + 1] + part[3][2 * t + 1];
            ((T*)out)[u] = (T)(g / (1.0f + lmi::fexp(-g)) * up);
        } else if (EPI == GEMV_QKV_ROPE_T) {
            // unit u = columns c1 = row_of(u, 0) and c1 + 32 of the permuted q | k | v row: rotate (q, k) from the fp32 sums, restore the
            // natural order, append K / V to the cache row of this step's position
            const float a = part[0][2 * t] + part[1][2 * t] + part[2][2 * t] + part[3][2 * t];
            const float b = part[0][2 * t + 1] + part[1][2 * t + 1] + part[2][2 * t + 1] + part[3][2 * t + 1];
            const int c1 = ((u >> 5) << 6) + (u & 31), p = *rp.pos;
            T* o = (T*)out;
            if (c1 < rp.rope_q + rp.rope_k) {
                const int hb = c1 & ~127, j = c1 & 127, d1 = ((j >> 6) << 5) + (j & 31);
                const float cs = rp.cos_all[(long)p * 64 + d1], sn = rp.sin_all[(long)p * 64 + d1];
                const T o1 = (T)(a * cs - b * sn), o2 = (T)(b * cs + a * sn);
                o[hb + d1] = o1;
                o[hb + d1 + 64] = o2;
                if (hb >= rp.rope_q) {
                    T* kc = (T*)rp.k_cache + (long)p * rp.ld_cache + (hb - rp.rope_q) + d1;
                    kc[0] = o1;
                    kc[64] = o2;
                }
            } else {
                const T va = (T)a, vb = (T)b;
                o[c1] = va;
                o[c1 + 32] = vb;
                T* vc = (T*)rp.v_cache + (long)p * rp.ld_cache + (c1 - rp.rope_q - rp.rope_k);
                vc[0] = va;
                vc[32] = vb;
            }
        } else {
            float v = part[0][t] + part[1][t] + part[2][t] + part[3][t];
            if (bias) v += bias[u];
            if (EPI == GEMV_STORE_F32) ((float*)out)[u] = v;
            else if (EPI == GEMV_STORE_T) ((T*)out)[u] = (T)v;
            else ((float*)out)[u] += v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// lmi_split_hi_lo: fp32 [M, K] -> 16-bit [M, 2K] = [ hi | lo ], hi = T(x), lo = T(x - hi).  The "split operand" precision mode
// (engine.split_operands): a GEMM over the 2K-wide operand against [W | W] computes hi.W + lo.W, i.e. the product of the UNROUNDED
// activation (to ~2^-22) on the 16-bit matrix pipe — the hand-over rounding that dominates the distance to the fp32 reference is gone.
// HBM-bound: 4 B in, 4 B out per element.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) split_hi_lo_kernel(const float* x, T* out, int M, int K, int ldx, int ldo, long lo_off) {
    typedef typename vec_of<T>::x8 T8;
    const int cpr = K >> 3;
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / cpr), c = (int)(i - (long)m * cpr);
        const float* src = x + (long)m * ldx + c * 8;
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
        T8 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (T)a[e]; hi[4 + e] = (T)b[e];
            lo[e] = (T)sub_rn(a[e], (float)hi[e]); lo[4 + e] = (T)sub_rn(b[e], (float)hi[4 + e]);
        }
        T* o = out + (long)m * ldo + c * 8;
        *(T8*)o = hi;
        *(T8*)(o + lo_off) = lo;                                      // K: [hi | lo] columns (lmi_split_hi_lo); M * ldo: lo rows below the hi rows (lmi_split_rows_hl)
    }
}

// ------------------------------------------------------------------------------------------------
// lmi_debug_copy: grid-stride 16-byte copy on a caller-chosen number of 256-thread workgroups.  Diagnostics only
// (tools/overlap_probe.py): a stand-in for a collective's transport kernel — few workgroups, no LDS, few registers — to observe
// whether such a kernel gets CU time beside a GEMM that holds one 128 KiB-LDS workgroup on every CU.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) debug_copy_kernel(const u32x4* src, u32x4* dst, long n16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// lmi_quantize_fp8: out[m, d] = fp8_e4m3(x[m, d] * scale) — the hand-over of an activation to an fp8 GEMM operand (static
// per-tensor power-of-two scale).  8 elements per lane per step (32 / 16 bytes in, 8 bytes out); HBM-bound.
// ------------------------------------------------------------------------------------------------
template <typename TIN>
__global__ void __launch_bounds__(256) quantize_fp8_kernel(const TIN* x, uint8_t* out, int M, int D, int ldx, int ldo, float scale) {
    const int cpr = D >> 3;                                        // 8-element chunks per row
    const long total = (long)M * cpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / cpr), c = (int)(i - (long)m * cpr);
        const TIN* src = x + (long)m * ldx + c * 8;
        u8x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_fp8((float)src[e] * scale);
        *(u8x8*)(out + (long)m * ldo + c * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// lmi_lm_head_last: logits of a few selected rows of the fp32 residual stream (the last position of every packed sequence
// in prefill, EVAL:333 restricted to what generate() consumes; the single row of a decode step).  The final RMSNorm is
// applied in the launch and the normalised row STAYS fp32 — it is never rounded to the 16-bit compute type — so the head
// is w(T, exact) x x(fp32) with fp32 FMAs: the one place on the path where full activation precision costs nothing,
// because the kernel is bound by the 1.05 GB weight stream (2 B / weight), not by arithmetic.
// grid = (row blocks of the vocabulary, selected rows); the normalised row lives in LDS (K floats), each wave streams R
// weight rows at a time, 16 bytes per lane per load, and reduces with wave64 shuffles.
// ------------------------------------------------------------------------------------------------
template <typename T, int R>
__global__ void __launch_bounds__(256) lm_head_rows_kernel(const T* W, const float* x, const long* rows, const float* gamma,
                                                           float eps, float* out, int N, int K, int ldw, int ldx, int ldo,
                                                           int rows_per_wg) {
    typedef typename vec_of<T>::x8 T8;
    LMI_DYN_SMEM(smem);
    float* xs = (float*)smem;                                      // [K] normalised row, + 4 floats of reduction scratch
    float* red = xs + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long src = rows ? rows[blockIdx.y] : (long)blockIdx.y;
    const float* xr = x + src * ldx;
    float ss = 0.f;
    for (int i = tid * 4; i < K; i += 1024) {
        const f32x4 v = *(const f32x4*)(xr + i);
        *(f32x4*)(xs + i) = v;
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (gamma) {
        const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
        for (int i = tid * 4; i < K; i += 1024) {                  // same element arithmetic as norm_kernel: w * (x * rstd)
            const f32x4 v = *(const f32x4*)(xs + i), g = *(const f32x4*)(gamma + i);
            *(f32x4*)(xs + i) = f32x4{g[0] * (v[0] * rstd), g[1] * (v[1] * rstd), g[2] * (v[2] * rstd), g[3] * (v[3] * rstd)};
        }
        __syncthreads();
    }
    const int n0 = blockIdx.x * rows_per_wg, n1 = imin(N, n0 + rows_per_wg);
    float* orow = out + (long)blockIdx.y * ldo;
    for (int nb = n0 + wave * R; nb < n1; nb += 4 * R) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        for (int c = lane * 8; c < K; c += 512) {
            const f32x4 x0 = *(const f32x4*)(xs + c), x1 = *(const f32x4*)(xs + c + 4);
            T8 wv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) wv[r] = ld_stream((const T8*)(W + (long)imin(nb + r, N - 1) * ldw + c));
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r] += (float)wv[r][e] * x0[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r] += (float)wv[r][4 + e] * x1[e];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0 && nb + r < n1) orow[nb + r] = v;
        }
    }
}

}  // namespace lmi
