// Microbenchmark: do MFMA and VALU work of co-resident waves (and of one wave) overlap on a gfx950 SIMD?
// The instruction streams are pinned with asm volatile + sched_barrier so the compiler cannot reshape them.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/overlap tools/ubench/overlap.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP>
__device__ __forceinline__ void vop(float& x, float c1, float c2) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
}

// role 0: every wave runs NM MFMAs with NV/NM VALU ops after each; role 1: waves 0..3 MFMA only, waves 4..7 VALU only
template <int NM, int NV, int OP, int NC = 4>
__global__ void __launch_bounds__(512, 1) k(float* out, int iters, int role, int nwaves, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[4] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float k1 = 1.0001f + out[0] * 0.f, k2 = 0.5f;
    const bool do_m = role == 0 ? (NM > 0) : (wave < 4);
    const bool do_v = role == 0 ? (NV > 0) : (wave >= 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                c[m % NC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[m % NC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < (NM > 0 ? NV / NM : 0); ++j) vop<OP>(v[j & 7], k1, k2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < (NM > 0 ? NM : 16); ++m) {
                c[m % NC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[m % NC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < (NV > 0 ? NV : 96); ++j) vop<OP>(v[j & 7], k1, k2);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    float s = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NM, int NV, int OP = 0, int NC = 4>
void run(float* out, int role, int nwaves, const char* what) {
    static unsigned long long* cyc = nullptr;
    if (!cyc) hipMalloc(&cyc, 64);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NM, NV, OP, NC><<<256, 512>>>(out, 10, role, nwaves, cyc);
    hipEventRecord(e0);
    k<NM, NV, OP, NC><<<256, 512>>>(out, iters, role, nwaves, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-62s %7.1f ns/iter | cycles/iter wave0 %6.0f wave%d %6.0f\n", what, ms * 1e6 / iters,
           (double)h[0] / iters, nwaves - 1, (double)h[nwaves - 1] / iters);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4);
    run<16, 0>(out, 0, 4, "1 wave/SIMD: 16 MFMA");
    run<0, 96>(out, 0, 4, "1 wave/SIMD: 96 v_fma");
    run<16, 16>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 1 v_fma)");
    run<16, 32>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 2 v_fma)");
    run<16, 64>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 4 v_fma)");
    run<16, 96>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 6 v_fma)");
    run<16, 128>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 8 v_fma)");
    run<16, 192>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 12 v_fma)");
    run<4, 96>(out, 0, 4, "1 wave/SIMD: 4 x (MFMA + 24 v_fma)");
    run<16, 96, 1>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 6 v_add_u32)");
    run<16, 96, 2>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 6 v_max_f32)");
    run<0, 96, 3>(out, 0, 4, "1 wave/SIMD: 96 v_exp_f32");
    run<16, 96, 3>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 6 v_exp_f32)");
    run<16, 96, 4>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 6 v_cvt_pk_f16_f32)");
    run<16, 0, 0, 1>(out, 0, 4, "1 wave/SIMD: 16 MFMA on ONE accumulator");
    run<16, 0, 0, 2>(out, 0, 4, "1 wave/SIMD: 16 MFMA on TWO accumulators (alternating)");
    run<16, 64, 0, 2>(out, 0, 4, "1 wave/SIMD: 16 x (MFMA + 4 v_fma), TWO accumulators");
    run<16, 0>(out, 0, 8, "2 waves/SIMD: each 16 MFMA");
    run<0, 96>(out, 0, 8, "2 waves/SIMD: each 96 v_fma");
    run<16, 96>(out, 1, 8, "2 waves/SIMD: one 16 MFMA, the other 96 v_fma");
    run<16, 96>(out, 0, 8, "2 waves/SIMD: each 16 x (MFMA + 6 v_fma)");
    return 0;
}
