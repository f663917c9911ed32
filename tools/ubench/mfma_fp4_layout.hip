// Pins, on the device, what the low-bit correction phase of csrc/gemm.h assumes about v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1)
// operands (cbsz = blgp = 4):
//   1. operand layout: lane l holds row l & 31, k = 32 (l >> 5) + j, j = 0..31, in the FIRST FOUR dwords of the operand, element j in
//      nibble j & 1 of byte j >> 1 (hypothesis LOW: even element in the low nibble; HIGH: in the high nibble);
//   2. E8M0 scales: lane l's scale byte (selected by op_sel from its scale dword) multiplies that lane's 32 elements: per (row, 32-k block);
//   3. software e2m1 rounding (lmi::to_fp4, round to nearest even, saturate at 6) == v_cvt_scalef32_pk_fp4_f32;
//   4. issue rate: cycles per MFMA, fp4 32x32x64 against f16 32x32x16 and fp8 32x32x64.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_fp4_layout mfma_fp4_layout.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static const float E2M1[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float dec4(int code) { return (code & 8 ? -1.f : 1.f) * E2M1[code & 7]; }

template <int OPA, int OPB>
__global__ void probe(const int* A, const int* B, const int* SA, const int* SB, float* out) {
    const int l = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    for (int i = 0; i < 4; ++i) { a[i] = A[l * 4 + i]; b[i] = B[l * 4 + i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, OPA, SA[l], OPB, SB[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

// software e2m1 quantiser (the one csrc uses): |x| on the grid {0, .5, 1, 1.5, 2, 3, 4, 6}, RNE, saturating
__host__ __device__ inline int to_fp4_sw(float x) {
    const int s = x < 0.f ? 8 : 0;
    float a = fabsf(x);
    if (!(a < 6.0f)) return s | 7;                       // saturate (NaN too)
    int code;
    if (a < 2.0f) { code = (int)rintf(a * 2.0f); }       // 0, .5, 1, 1.5, (2.0 -> code 4)
    else if (a < 4.0f) { code = 2 + (int)rintf(a); }     // 2 -> 4, 3 -> 5, 4 -> 6
    else { code = 4 + (int)rintf(a * 0.5f); }            // 4 -> 6, 6 -> 7
    return s | code;
}
__global__ void cvt_probe(const float* x, int n, float scale, int* hw, int* sw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x[i], 0.0f, scale, 0);
    hw[i] = (int)(r & 15);
    sw[i] = to_fp4_sw(x[i] / scale);
}

template <int MODE>
__global__ void rate(float* out, unsigned long long* cyc, int iters) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    v8i a = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0x22222222, 0x22222222, 0x22222222, 0x22222222}, b = a;
    f16x8 ah = {1, 1, 1, 1, 1, 1, 1, 1}, bh = ah;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OPA, int OPB>
static void run_layout(const std::vector<int>& codesA, const std::vector<int>& codesB, const std::vector<int>& sa, const std::vector<int>& sb) {
    // pack under hypothesis LOW (even element in the low nibble) and compute host expectations under LOW and HIGH
    std::vector<int> A(64 * 4), B(64 * 4);
    for (int l = 0; l < 64; ++l)
        for (int d = 0; d < 4; ++d) {
            unsigned wa = 0, wb = 0;
            for (int e = 0; e < 8; ++e) { wa |= (unsigned)codesA[l * 32 + d * 8 + e] << (4 * e); wb |= (unsigned)codesB[l * 32 + d * 8 + e] << (4 * e); }
            A[l * 4 + d] = (int)wa; B[l * 4 + d] = (int)wb;
        }
    int *dA, *dB, *dSA, *dSB; float* dO;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dO, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dSA, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dSB, sb.data(), 256, hipMemcpyHostToDevice);
    probe<OPA, OPB><<<1, 64>>>(dA, dB, dSA, dSB, dO);
    std::vector<float> O(1024);
    hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
    for (int hyp = 0; hyp < 2; ++hyp) {        // 0 = LOW nibble first, 1 = HIGH nibble first (elements of a byte swapped)
        double maxd = 0, maxv = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                double acc = 0;
                for (int kb = 0; kb < 2; ++kb) {
                    const int la = row + 32 * kb, lb = col + 32 * kb;
                    const double fa = ldexp(1.0, ((sa[la] >> (8 * OPA)) & 255) - 127), fb = ldexp(1.0, ((sb[lb] >> (8 * OPB)) & 255) - 127);
                    double s = 0;
                    for (int j = 0; j < 32; ++j) {
                        const int jj = hyp ? (j ^ 1) : j;
                        s += (double)dec4(codesA[la * 32 + jj]) * dec4(codesB[lb * 32 + jj]);
                    }
                    acc += s * fa * fb;
                }
                maxd = fmax(maxd, fabs(acc - O[l * 16 + r]));
                maxv = fmax(maxv, fabs(acc));
            }
        printf("  op_sel a=%d b=%d  hypothesis %s: max|device - host| = %.3e (max|host| = %.3e)\n", OPA, OPB, hyp ? "HIGH-nibble-first" : "LOW-nibble-first", maxd, maxv);
    }
    hipFree(dA); hipFree(dB); hipFree(dSA); hipFree(dSB); hipFree(dO);
}

int main() {
    srand(12345);
    std::vector<int> ca(64 * 32), cb(64 * 32), sa(64), sb(64);
    for (auto& v : ca) v = rand() & 15;
    for (auto& v : cb) v = rand() & 15;
    // symmetric-in-nibble data could hide the order: make sure A is asymmetric (it is random), B likewise
    for (int l = 0; l < 64; ++l) {
        sa[l] = 0; sb[l] = 0;
        for (int k = 0; k < 4; ++k) { sa[l] |= (120 + (rand() % 14)) << (8 * k); sb[l] |= (120 + (rand() % 14)) << (8 * k); }
    }
    printf("== 1/2. fp4 operand layout and per-lane E8M0 scales (random codes, random per-lane scale bytes) ==\n");
    run_layout<0, 0>(ca, cb, sa, sb);
    run_layout<1, 2>(ca, cb, sa, sb);
    run_layout<3, 1>(ca, cb, sa, sb);
    run_layout<2, 3>(ca, cb, sa, sb);

    printf("== 3. software e2m1 rounding vs v_cvt_scalef32_pk_fp4_f32 ==\n");
    {
        std::vector<float> x;
        for (int i = -1400; i <= 1400; ++i) x.push_back(i * (1.0f / 200.0f));           // -7 .. 7 step 0.005: every tie of the grid included
        for (int i = 0; i < 4000; ++i) x.push_back(((rand() & 0xffff) / 65536.0f - 0.5f) * 16.0f);
        x.push_back(1e30f); x.push_back(-1e30f); x.push_back(1e-30f);
        const int n = (int)x.size();
        float* dx; int *dh, *ds;
        hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&ds, n * 4);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        for (float scale : {1.0f, 0.25f, 8.0f}) {
            cvt_probe<<<(n + 255) / 256, 256>>>(dx, n, scale, dh, ds);
            std::vector<int> h(n), s(n);
            hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int i = 0; i < n; ++i) {
                const bool same = h[i] == s[i] || ((h[i] & 7) == 0 && (s[i] & 7) == 0);      // +0 / -0 both fine
                if (!same && bad++ < 8) printf("    x = %g scale %g: hw code %d (%g) sw code %d (%g)\n", x[i], scale, h[i], dec4(h[i]), s[i], dec4(s[i]));
            }
            printf("  scale %g: %d values, %d mismatches\n", scale, n, bad);
        }
        hipFree(dx); hipFree(dh); hipFree(ds);
    }

    printf("== 4. issue rate (one wave, 4 independent accumulators, s_memtime ticks at 100 MHz) ==\n");
    {
        float* dout; unsigned long long* dc;
        hipMalloc(&dout, 256); hipMalloc(&dc, 8);
        const int iters = 20000;
        const char* names[3] = {"f16 32x32x16", "fp8 32x32x64 (scaled)", "fp4 32x32x64 (scaled)"};
        for (int m = 0; m < 3; ++m) {
            unsigned long long c = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) rate<0><<<1, 64>>>(dout, dc, iters);
                if (m == 1) rate<1><<<1, 64>>>(dout, dc, iters);
                if (m == 2) rate<2><<<1, 64>>>(dout, dc, iters);
                hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            }
            const double flops = (m == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 64) * 4.0 * iters;
            printf("  %-24s %llu ticks for %d x 4 MFMAs = %.2f ns per MFMA, %.1f GFLOP/s per wave\n", names[m], c, iters, c * 10.0 / (4.0 * iters), flops / (c * 10.0));
        }
    }
    return 0;
}
