#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A, B: per lane 32 bytes (fp8 e4m3).  out: per lane 16 floats.
__global__ void probe(const uint8_t* A, const uint8_t* B, float* out, int scale_a, int scale_b) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)A)[l * 8 + i]; b[i] = ((const int*)B)[l * 8 + i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0 /*cbsz: A fp8*/, 0 /*blgp: B fp8*/, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}
int main() {
    std::vector<uint8_t> A(64 * 32), B(64 * 32);
    uint8_t *dA, *dB; float* dO;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dO, 64 * 16 * 4);
    std::vector<float> O(64 * 16);
    const uint8_t ONE = 0x38;      // e4m3 1.0 = 0 0111 000
    const uint8_t TWO = 0x40;      // 2.0
    // Experiment 1: A all ones, B all ones, scales 127 -> every output should be 64
    for (int sa : {127, 128, 126}) {
        std::fill(A.begin(), A.end(), ONE); std::fill(B.begin(), B.end(), ONE);
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        const int s = sa | (sa << 8) | (sa << 16) | (sa << 24);
        probe<<<1, 64>>>(dA, dB, dO, s, 0x7f7f7f7f);
        hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
        printf("scale_a=%d: out[0]=%g out[last]=%g\n", sa, O[0], O[1023]);
    }
    // Experiment 2: A layout.  B = all ones.  A: lane la, byte ba = 2.0 (others 0) -> which output rows/cols light up (value 2)?
    // C/D layout known: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).   A . B: out[i][j] = sum_k A[i][k] B[k][j]
    printf("A probe (lane, byte) -> row i that lights up (all cols)\n");
    for (int la : {0, 1, 31, 32, 33, 63}) for (int ba : {0, 1, 15, 16, 31}) {
        std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), ONE);
        A[la * 32 + ba] = TWO;
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dO, 0x7f7f7f7f, 0x7f7f7f7f);
        hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
        int rows = 0, first = -1, cols = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (O[l * 16 + r] != 0) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            if (first < 0) first = row;
            if (row != first) rows++;
            cols++;
        }
        printf("  A lane %2d byte %2d -> row %d (other rows %d, nonzero outputs %d, value %g)\n", la, ba, first, rows, cols, first >= 0 ? 2.0 : 0.0);
    }
    // Experiment 3: k index.  A lane la byte ba = 2 ; B lane lb byte bb = 2; others 0: nonzero (=4) iff same k.
    printf("k mapping: for A(lane 0, byte ba) find B(lane lb in {0,32}, byte bb) with nonzero product\n");
    for (int la : {0, 32}) for (int ba : {0, 1, 7, 8, 16, 31}) {
        for (int lb : {0, 32}) for (int bb = 0; bb < 32; ++bb) {
            std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0);
            A[la * 32 + ba] = TWO; B[lb * 32 + bb] = TWO;
            hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
            probe<<<1, 64>>>(dA, dB, dO, 0x7f7f7f7f, 0x7f7f7f7f);
            hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 1024; ++i) if (O[i] != 0) { printf("  A(lane %d, byte %d) x B(lane %d, byte %d) -> out lane %d reg %d = %g\n", la, ba, lb, bb, i / 16, i % 16, O[i]); break; }
        }
    }
    // Experiment 4: per-lane scale: scale_a differs per lane? set A ones, B ones, scale_a byte0 = 127 + (lane & 1)
    return 0;
}
