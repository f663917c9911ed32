// hipemu — a tiny lock-step emulator of the HIP execution model, for unit-testing kernel LOGIC on a
// host without a GPU.  One workgroup at a time; every thread of the workgroup is a ucontext fibre;
// __syncthreads() and wave-collective primitives (MFMA, shuffles, LDS transpose reads) are rendezvous
// points.  DEVELOPMENT TOOL ONLY: never linked into libleopard_amd.so, never loaded by leopard_amd.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }


void __syncthreads();

namespace hipemu {
void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);
void barrier();            // workgroup rendezvous and nothing else (s_barrier)
void wave_sync();          // rendezvous of the 64 lanes of the calling thread's wave
// LDS-DMA (buffer_load ... lds) with its asynchrony modelled at BOTH worst cases: the destination is poisoned (0xFF: NaN in every operand
// type) the moment the piece is issued — the hardware may overwrite it at any time from then on — and receives its bytes only when a
// vmcnt wait of the issuing thread retires the piece (in order, oldest first) — the hardware may deliver it that late.  A kernel that
// reads a ring slot before its counted wait covers the slot's pieces, or still reads the previous tenant after issuing into the slot, gets
// NaNs on the host exactly where it has a race on the device.  HIPEMU_SYNC_DMA=1 restores immediate copies (debugging).
void dma_issue(void* lds_dst, const void* src, int bytes);     // src == nullptr: zeros (out-of-range buffer read)
void dma_wait(int max_outstanding);                            // s_waitcnt vmcnt(N) of the calling thread
void* wave_buf();          // per-wave scratch (64 lanes x 256 B)
char* dyn_smem();
}  // namespace hipemu
