// Microbenchmark: what matrix rate does a power-capped MI355X sustain on a pure MFMA stream (no memory traffic)?
// Variants: v_mfma_f32_32x32x16 vs 16x16x32, f16 vs bf16, 1 or 2 waves per SIMD, operands = gaussian-like values.
// Each launch runs ~100+ ms so that the power controller settles.   build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned& s) {          // cheap gaussian-ish in [-2, 2]
    s = s * 1664525u + 1013904223u;
    const float u1 = (s >> 8) * (1.0f / 16777216.0f);
    s = s * 1664525u + 1013904223u;
    const float u2 = (s >> 8) * (1.0f / 16777216.0f);
    return (u1 + u2 - 1.0f) * 2.0f;
}

template <int KIND>   // 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16, 3: 16x16x32 bf16
__global__ void __launch_bounds__(512) k(float* out, int iters, float wscale) {
    unsigned s = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
    f16x8 ah[4], bh[4];
    bf16x8 ab[4], bb[4];
    for (int f = 0; f < 4; ++f)
        for (int i = 0; i < 8; ++i) {
            const float x = rnd(s), w = rnd(s) * wscale;
            ah[f][i] = (_Float16)x; bh[f][i] = (_Float16)w; ab[f][i] = (__bf16)x; bb[f][i] = (__bf16)w;
        }
    f32x16 c[8] = {};
    f32x4 d[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (KIND == 0) c[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 3], bh[m >> 2], c[m & 7], 0, 0, 0);
            if (KIND == 2) c[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[m & 3], bb[m >> 2], c[m & 7], 0, 0, 0);
            if (KIND == 1) {
                d[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m & 3], bh[m >> 2], d[m], 0, 0, 0);
                d[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(m + 1) & 3], bh[m >> 2], d[m], 0, 0, 0);
            }
            if (KIND == 3) {
                d[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[m & 3], bb[m >> 2], d[m], 0, 0, 0);
                d[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[(m + 1) & 3], bb[m >> 2], d[m], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
    for (int m = 0; m < 16; ++m) r += c[m & 7][m] + d[m][m & 3];
    if (r == 123.456f) out[0] = r;
}

template <int KIND>
void run(const char* name, int threads, float wscale, float* out) {
    const int blocks = 256;                                    // one workgroup per CU
    const int iters = (threads == 512 ? 300000 : 600000);   // >= 150 ms per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(out, 2000, wscale);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, threads>>>(out, iters, wscale);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_iter = (KIND & 1) ? 16.0 * 2 * 2 * 16 * 16 * 32 : 16.0 * 2 * 32 * 32 * 16;
    const double tf = flop_per_iter * iters * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma32 = 32.0;                        // 8 passes x 4 cycles for 16K MACs
    const double ghz = tf * 1e12 / (256.0 * 4 * 1024) / 1e9 / ((threads / 64) >= 4 ? 1.0 : (threads / 64) / 4.0);
    printf("%-28s waves/CU %d  w_scale %.2f : %7.1f ms  %7.1f TFLOP/s  (= %.2f GHz at 100 %% pipe)\n", name, threads / 64, wscale, ms, tf, ghz);
    (void)cyc_per_mfma32;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    for (float ws : {1.0f, 0.02f}) {
        run<0>("32x32x16 f16", 256, ws, out);
        run<0>("32x32x16 f16", 512, ws, out);
        run<1>("16x16x32 f16", 256, ws, out);
        run<1>("16x16x32 f16", 512, ws, out);
        run<2>("32x32x16 bf16", 256, ws, out);
        run<3>("16x16x32 bf16", 256, ws, out);
    }
    return 0;
}
