// What would a PERSISTENT per-layer decode kernel pay per phase boundary?  (round 5; DESIGN.md 4 "Decode")
//
// The decode step is ~200 dependent launches inside one HIP graph, ~3.5 us of ramp + tail each.  A persistent kernel replaces a launch
// boundary by a grid-wide barrier with release / acquire at agent scope (the phases hand data from every workgroup to every workgroup, across
// the eight XCD L2s).  This program measures both on the device, with the same payload:
//   persistent   G workgroups x 256 threads, R rounds: write 1 KiB, grid barrier (atomic arrive + acquire spin), read another workgroup's 1 KiB
//   graph        the same round as R dependent kernel launches captured into one hipGraph
// and checks that the barrier version observed every hand-over (a checksum), so the fences measured are the ones a real kernel needs.
// Build: hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// STREAM floats per thread of read-only traffic per round (the weight stream of a decode phase keeps the memory system busy while the
// barrier traffic goes through it); 0 = barrier latency alone
template <int STREAM, int VARIANT>
__global__ void __launch_bounds__(256) persistent(float* buf, unsigned* counter, const float* stream, size_t stream_elems, float* out, int R) {
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    float carry = 0.f, acc = 0.f;
    for (int r = 0; r < R; ++r) {
        buf[((size_t)(r & 1) * G + b) * 256 + t] = carry + (float)(b + r);
        if (STREAM) {
            const size_t base = ((size_t)r * G + b) * 256 * STREAM % (stream_elems - 256 * STREAM);
#pragma unroll
            for (int i = 0; i < STREAM; ++i) acc += __builtin_nontemporal_load(stream + base + i * 256 + t);
        }
        __syncthreads();
        if (t == 0) {
            const unsigned want = (unsigned)G * (unsigned)(r + 1);
            int spins = 0;                                                          // bounded: a lost arrival must not hang the box
            if (VARIANT == 0) {            // textbook: release arrive, acquire poll (every poll carries the cache invalidate)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            } else if (VARIANT == 1) {     // relaxed polls, one acquire fence at the end (below)
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            } else {                       // two levels: 8 group counters (workgroup b -> group b & 7, its XCD under round-robin dispatch), the last
                                           // arrival of a group arrives at the top counter, the last one there publishes the round in 8 flag lines
                unsigned* grp = counter + 32 * (1 + (b & 7));
                unsigned* flag = counter + 32 * (9 + (b & 7));
                const unsigned gsz = (unsigned)((G - (b & 7) + 7) / 8);
                const unsigned prev = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1 == gsz * (unsigned)(r + 1)) {
                    const unsigned top = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    if (top + 1 == 8u * (unsigned)(r + 1))
                        for (int x = 0; x < 8; ++x) __hip_atomic_store(counter + 32 * (9 + x), (unsigned)(r + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                                   // (every thread reads other workgroups' data)
        carry = __hip_atomic_load(&buf[((size_t)(r & 1) * G + (b + 37) % G) * 256 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)b * 256 + t] = carry + acc * 0.f;
}

// the runtime's own grid barrier (cooperative launch, cooperative_groups::this_grid().sync()) with the same payload
template <int STREAM>
__global__ void __launch_bounds__(256) persistent_cg(float* buf, const float* stream, size_t stream_elems, float* out, int R) {
    namespace cg = cooperative_groups;
    cg::grid_group grid = cg::this_grid();
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    float carry = 0.f, acc = 0.f;
    for (int r = 0; r < R; ++r) {
        buf[((size_t)(r & 1) * G + b) * 256 + t] = carry + (float)(b + r);
        if (STREAM) {
            const size_t base = ((size_t)r * G + b) * 256 * STREAM % (stream_elems - 256 * STREAM);
#pragma unroll
            for (int i = 0; i < STREAM; ++i) acc += __builtin_nontemporal_load(stream + base + i * 256 + t);
        }
        grid.sync();
        carry = __hip_atomic_load(&buf[((size_t)(r & 1) * G + (b + 37) % G) * 256 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)b * 256 + t] = carry + acc * 0.f;
}

template <int STREAM>
__global__ void __launch_bounds__(256) one_round(float* buf, const float* stream, size_t stream_elems, float* carry_io, int r) {
    const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
    const float carry = r == 0 ? 0.f : buf[((size_t)((r - 1) & 1) * G + (b + 37) % G) * 256 + t];
    if (STREAM) {
        const size_t base = ((size_t)r * G + b) * 256 * STREAM % (stream_elems - 256 * STREAM);
#pragma unroll
        for (int i = 0; i < STREAM; ++i) acc += __builtin_nontemporal_load(stream + base + i * 256 + t);
    }
    buf[((size_t)(r & 1) * G + b) * 256 + t] = carry + (float)(b + r) + acc * 0.f;
    (void)carry_io;
}

static double expected(int G, int b, int R) {       // carry after R rounds at workgroup b: sum over the chain of (b_k + r)
    double c = 0;                                   // value written at round r by workgroup x: carry_x(r) + x + r; carry_b(r+1) = that of x = b + 37
    std::vector<double> cur(G, 0.0), nxt(G);
    for (int r = 0; r < R; ++r) {
        for (int x = 0; x < G; ++x) nxt[x] = cur[(x + 37) % G] + ((x + 37) % G) + r;
        cur.swap(nxt);
    }
    c = cur[b];
    return c;
}

template <int STREAM>
static void run(int G, int R, float* stream, size_t stream_elems) {
    float *buf, *out;
    unsigned* counter;
    CK(hipMalloc(&buf, (size_t)2 * G * 256 * 4));
    CK(hipMalloc(&out, (size_t)G * 256 * 4));
    CK(hipMalloc(&counter, 32 * 17 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_v[3] = {1e30f, 1e30f, 1e30f}, ms_g = 1e30f;
    int bad = 0;
    std::vector<float> h((size_t)G * 256);
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(counter, 0, 32 * 17 * 4, s));
            CK(hipEventRecord(e0, s));
            if (v == 0) persistent<STREAM, 0><<<G, 256, 0, s>>>(buf, counter, stream, stream_elems, out, R);
            if (v == 1) persistent<STREAM, 1><<<G, 256, 0, s>>>(buf, counter, stream, stream_elems, out, R);
            if (v == 2) persistent<STREAM, 2><<<G, 256, 0, s>>>(buf, counter, stream, stream_elems, out, R);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < ms_v[v]) ms_v[v] = ms;
        }
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < G; b += 17)
            if ((double)h[(size_t)b * 256 + 5] != (double)(float)expected(G, b, R)) ++bad;
    }
    float ms_cg = 1e30f;
    {
        int Rv = R;
        void* kargs[] = {&buf, &stream, &stream_elems, &out, &Rv};
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipLaunchCooperativeKernel((const void*)persistent_cg<STREAM>, dim3(G), dim3(256), kargs, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < ms_cg) ms_cg = ms;
        }
        CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < G; b += 17)
            if ((double)h[(size_t)b * 256 + 5] != (double)(float)expected(G, b, R)) ++bad;
    }
    // the same rounds as a graph of dependent launches
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < R; ++r) one_round<STREAM><<<G, 256, 0, s>>>(buf, stream, stream_elems, out, r);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < ms_g) ms_g = ms;
    }
    printf("G = %4d workgroups, %3d rounds, %5.1f KiB streamed per workgroup and round:  grid barrier %6.2f (acquire polls) %6.2f (relaxed polls) %6.2f (two-level) %6.2f (cooperative-groups grid.sync) us / round   graph of launches %6.2f us / round   hand-overs %s\n",
           G, R, STREAM * 1.0, ms_v[0] * 1e3 / R, ms_v[1] * 1e3 / R, ms_v[2] * 1e3 / R, ms_cg * 1e3 / R, ms_g * 1e3 / R, bad ? "WRONG" : "verified");
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    CK(hipFree(buf));
    CK(hipFree(out));
    CK(hipFree(counter));
    CK(hipStreamDestroy(s));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t stream_elems = (size_t)1 << 29;                  // 2 GiB of read-only "weights"
    float* stream;
    CK(hipMalloc(&stream, stream_elems * 4));
    CK(hipMemset(stream, 0, stream_elems * 4));
    const int R = 200;
    for (int G : {256, 768}) run<0>(G, R, stream, stream_elems);
    for (int G : {256, 768}) run<64>(G, R, stream, stream_elems);         // 64 KiB per workgroup and round (16 - 50 MB per phase)
    return 0;
}
