// lowbit.h — operands of the low-bit correction phase (gemm.h, GemmArgs::A4 / W4): MX fp4 (e2m1) images.
//
// Reference arithmetic served: none new — this is how the HIP path gets from "one rounding of every activation to the 16-bit compute
// type" (1.1 - 1.7e-3 of the logit scale at the full 59-layer depth) to north_star's "logits within 1e-3" without a second 16-bit pass:
// the rounding residual x - T(x) of every layer-linear A operand is handed over as well, as a 4-bit image with one power-of-two scale per
// 32 elements, and multiplied with a 4-bit image of the weight at 4 x the 16-bit MFMA rate into the same accumulators.
//   lmi_split_lo4     fp32 [M, K] -> T [M, K] (= T(x)), fp4 [M, K4 / 2 bytes] (image of x - T(x)), E8M0 [M, K4 / 32]   (attention outputs)
//   lmi_quantize_w4   T [N, K]    -> fp4 [N, K4 / 2 bytes], E8M0 [N] (one scale per weight row)                         (once, at load)
// The norm kernels (elementwise.h) and the GEMM epilogues (gemm.h) write the same images directly.  K4 = K rounded up to 256; the
// padding holds zero codes and zero scale bytes.  HBM-bound: 4 B in, 2 + 0.5 + 1/32 B out per element.
#pragma once
#include "lmi_device.h"

namespace lmi {

// 8 elements per thread, a lane quad = one 32-element block (K % 32 == 0 and K4 % 256 == 0 keep quads inside a row)
template <typename T>
__global__ void __launch_bounds__(256) split_lo4_kernel(const float* x, T* hi, uint8_t* lo4, uint8_t* scales, int M, int K, int K4, int ldx,
                                                        int ldh, int ld4, int lds) {
    typedef typename vec_of<T>::x8 T8;
    const int cpr = K4 >> 3, creal = K >> 3;
    const long total = (long)M * cpr;                                // total % 4 == 0: a lane quad is live or idle as a whole
    // wave-uniform trip count (the quad exchange is a wave-level operation in the host emulator): idle lanes run on zeros and store nothing
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i - (threadIdx.x & 63) < total; i += (long)gridDim.x * blockDim.x) {
        const bool live = i < total;
        const long ii = live ? i : 0;
        const int m = (int)(ii / cpr), c = (int)(ii - (long)m * cpr);
        float y[8];
        if (live && c < creal) {
            const float* src = x + (long)m * ldx + c * 8;
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { y[e] = a[e]; y[4 + e] = b[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = 0.f;
        }
        T8 h;
        unsigned sb;
        const unsigned codes = lo4_encode8<T>(y, h, sb);
        if (!live) continue;
        if (c < creal) *(T8*)(hi + (long)m * ldh + c * 8) = h;
        *(unsigned*)(lo4 + (long)m * ld4 + c * 4) = codes;
        if ((c & 3) == 0) scales[(long)m * lds + (c >> 2)] = (uint8_t)sb;
    }
}

// one wave per weight row: amax of the row, then the codes under the row's scale.  Rows are `ldw` elements apart (row-major W).
template <typename T>
__global__ void __launch_bounds__(256) quantize_w4_kernel(const T* W, uint8_t* w4, uint8_t* scales, int N, int K, int K4, int ldw, int ld4) {
    typedef typename vec_of<T>::x8 T8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const T* wr = W + (long)row * ldw;
    const int creal = K >> 3, cpr = K4 >> 3;
    float amax = 0.f;
    for (int c = lane; c < creal; c += 64) {
        const T8 v = *(const T8*)(wr + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, __builtin_fabsf((float)v[e]));
    }
    amax = wave_max(amax);
    float scale, inv;
    const unsigned sb = lo4_scale_byte(amax, scale, inv);
    for (int c = lane; c < cpr; c += 64) {
        unsigned codes = 0;
        if (c < creal) {
            const T8 v = *(const T8*)(wr + c * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
            codes = fp4_pack8(f, scale, inv);
        }
        *(unsigned*)(w4 + (long)row * ld4 + c * 4) = codes;
    }
    if (lane == 0) scales[row] = (uint8_t)sb;
}

}  // namespace lmi
