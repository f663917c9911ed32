// capi.hip — extern "C" launchers of libleopard_amd.so (see include/leopard_amd.h for the contract).
// Built by hipcc for gfx950; the same file builds against tools/hipemu with -DLMI_EMU for CPU logic tests.
#include "../../include/leopard_amd.h"

#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "attention.h"
#include "attention64.h"
#include "attention_fp8.h"
#include "elementwise.h"
#include "patch_embed.h"
#include "gemm.h"
#include "lowbit.h"
#include "skinny.h"

using namespace lmi;

#ifdef LMI_ATTN_PROF
namespace lmi { __device__ unsigned long long* g_prof_buf = nullptr; }
extern "C" int lmi_debug_set_prof_buffer(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(lmi::g_prof_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LMI_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return LMI_OK;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Raise the dynamic-LDS limit of `kernel` on the CURRENT device.  The attribute is per device, so the "already done" flag is a
// per-instantiation bit mask indexed by the device ordinal (a process may drive several GPUs, one thread each); atomic because
// the header promises re-entrancy per stream.
template <typename K>
void allow_big_lds(K kernel, int bytes, std::atomic<uint64_t>& done) {
#ifndef LMI_EMU
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_relaxed);
#else
    (void)kernel; (void)bytes; (void)done;
#endif
}

int grid_for(long n, int block, int cap = 256 * 8) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- GEMM dispatch ---------------------------------------------------------------------------------------
// tile geometries (see gemm.h GemmCfg<BM, BN, WAVES_M, WAVES_N, STAGES>)
typedef GemmCfg<128, 128, 2, 2, 2> Cfg0;   //  64 KiB LDS, 256 threads, 2 workgroups / CU
typedef GemmCfg<256, 256, 2, 4, 2> Cfg1;   // 128 KiB LDS, 512 threads, wave tile 128x64
typedef GemmCfg<256, 128, 4, 2, 3> Cfg2;   // 144 KiB LDS, 512 threads, wave tile 64x64, 3-slot ring
typedef GemmCfg<256, 128, 4, 2, 2> Cfg3;   //  96 KiB LDS
typedef GemmCfg<128, 256, 2, 4, 3> Cfg4;   // 144 KiB LDS, wave tile 64x64, 3-slot ring
typedef GemmCfg<64, 128, 2, 2, 3> CfgS;    //  72 KiB LDS, 256 threads, wave tile 32x64: small-M problems
typedef GemmCfg<64, 128, 2, 2, 6> CfgS6;   // 144 KiB LDS: 5 k-tiles in flight for latency-bound weight streaming at small M
typedef GemmCfg<384, 128, 4, 2, 2> CfgM;   // 128 KiB LDS, wave tile 96x64: M-complete tiles for 256 < M <= 384 (every weight tile is fetched ONCE)
constexpr int kNumGemmCfg = 11;             // 0-4 ring geometries; 5-7 staggered 256x256 schedules; 8-9 small-M rings; 10 M-complete 384x128
std::atomic<int> g_gemm_cfg{-1};            // -1 = choose per shape (options are process-wide; relaxed atomics: launches on other threads read them)
std::atomic<int> g_gemm_group_m{GEMM_GROUP_M};
std::atomic<int> g_gemm_order{0};
// per-shape-class geometry (end-to-end A/B knobs; defaults = best measured on the C3 prefill, which runs at the power cap
// and does not always agree with isolated bursts of one GEMM)
std::atomic<int> g_gemm_wide{5};                        // N >= 2048, K > 1536 (Llama projections)
std::atomic<int> g_gemm_short{5};                       // N >= 2048, K <= 1536 (SigLIP qkv, fc1)
std::atomic<int> g_gemm_narrow{2};                      // N < 2048, K >= 2048 (SigLIP fc2)
std::atomic<int> g_gemm_small{0};                       // N < 2048, K < 2048 (SigLIP out_proj, patch embedding)

template <typename T, int EPI, int ACT, int AMODE, typename C>
int launch_gemm_cfg(const GemmArgs& a, void* stream) {
    int tiles = ((a.M + C::BM - 1) / C::BM) * ((a.N + C::BN - 1) / C::BN);
    if (a.order == 1) tiles = (tiles + 255) / 256 * 256;
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(gemm_kernel<T, EPI, ACT, AMODE, C>, C::SMEM_TOTAL, attr_done);
    LMI_LAUNCH((gemm_kernel<T, EPI, ACT, AMODE, C>), dim3(tiles), dim3(C::NT), C::SMEM_TOTAL, stream, a);
    return check_launch("lmi_gemm");
}

template <typename T, int EPI, int ACT, int AMODE, typename C, int VAR>
int launch_gemm_stagger(const GemmArgs& a, void* stream) {
    int tiles = ((a.M + C::BM - 1) / C::BM) * ((a.N + C::BN - 1) / C::BN);
    if (a.order == 1) tiles = (tiles + 255) / 256 * 256;
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(gemm_stagger_kernel<T, EPI, ACT, AMODE, C, VAR>, C::SMEM_TOTAL, attr_done);
    LMI_LAUNCH((gemm_stagger_kernel<T, EPI, ACT, AMODE, C, VAR>), dim3(tiles), dim3(C::NT), C::SMEM_TOTAL, stream, a);
    return check_launch("lmi_gemm");
}

// Geometry per shape class, from end-to-end A/B runs of the C3 prefill (`bench.py --opt`, profiles/README.md): N >= 2048 takes the
// staggered 256x256 schedule; narrow-N / deep-K (SigLIP fc2) the 256x128 3-slot ring; small problems 128x128.
// Problems that leave the class geometry fewer than two rounds of tiles on the 256 CUs (one image instead of six: BASELINE
// configs C1 / C2, Idefics2's text side) are decided by the tile count instead: cost = tiles per CU x tile area / the geometry's
// relative efficiency on large problems (256x256 staggered 1.0, 256x128 0.85, 128x128 0.75, 64x128 0.55; tools/sweep_fp8_cfg.py and
// the round-1 sweeps).  Every C3 / C5 shape has >= 464 tiles and evaluates to its class geometry, so the headline path is unchanged.
std::atomic<int> g_gemm_mid_m{1};            // lmi_set_option("gemm.mid_m", 0) = 64x128 tiles for every M < 512 (A/B)
std::atomic<int> g_gemm_auto_small{1};                  // lmi_set_option("gemm.auto_small", 0) = class rules only (A/B)
int choose_gemm_cfg(const GemmArgs& a) {
    const int forced = g_gemm_cfg.load(std::memory_order_relaxed);
    if (forced >= 0) return forced;
    // 256 < M <= 384 (Idefics2's text side: S = 312) on a wide N: one M-complete 384x128 tile per 128 weight rows — the weights are streamed
    // once instead of once per 64-row tile (5x at M = 312); narrower N leaves too few tiles for 256 CUs and stays on the 64x128 ring
    if (g_gemm_mid_m.load() && a.M > 256 && a.M <= 384 && a.N >= 128 * 160) return 10;
    // 128 < M <= 256 on a wide N (C1's gate/up at S = 228): the 256 x 128 ring is M-complete there too — 224 tiles of 8 waves, every weight tile
    // fetched once — against 896 tiles of the 64 x 128 ring (isolated launches: 61 vs 79 us, profiles/r06_small_m_gemm_sweep.txt)
    if (g_gemm_mid_m.load() && a.M > 128 && a.M <= 256 && a.N >= 128 * 160) return 2;
    if (a.M < 512) return 8;
    const int cls = a.N >= 2048 ? (a.K <= 1536 ? g_gemm_short.load() : g_gemm_wide.load()) : (a.K >= 2048 ? g_gemm_narrow.load() : g_gemm_small.load());
    if (!g_gemm_auto_small.load()) return cls;
    struct Geo { int cfg, bm, bn; float eff; };
    auto geo_of = [](int cfg) -> Geo {
        switch (cfg) {
            case 1: case 5: case 6: case 7: return {cfg, 256, 256, 1.0f};
            case 2: case 3: return {cfg, 256, 128, 0.85f};
            case 4: return {cfg, 128, 256, 0.85f};
            case 8: case 9: return {cfg, 64, 128, 0.55f};
            default: return {cfg, 128, 128, 0.75f};
        }
    };
    auto tiles = [&](const Geo& g) { return (long)((a.M + g.bm - 1) / g.bm) * ((a.N + g.bn - 1) / g.bn); };
    const Geo c0 = geo_of(cls);
    if (tiles(c0) >= 512) return cls;
    auto cost = [&](const Geo& g) { return (float)((tiles(g) + 255) / 256) * (float)(g.bm * g.bn) / g.eff; };
    Geo best = c0;
    float best_cost = cost(c0);
    for (int cfg : {2, 0, 8}) {
        const Geo g = geo_of(cfg);
        const float c = cost(g);
        if (c < best_cost * 0.97f) { best = g; best_cost = c; }      // ties stay with the larger tile
    }
    return best.cfg;
}

template <typename T, int EPI, int ACT, int AMODE>
int launch_gemm(const GemmArgs& a, void* stream) {
    switch (choose_gemm_cfg(a)) {
        case 1: return launch_gemm_cfg<T, EPI, ACT, AMODE, Cfg1>(a, stream);
        case 2: return launch_gemm_cfg<T, EPI, ACT, AMODE, Cfg2>(a, stream);
        case 3: return launch_gemm_cfg<T, EPI, ACT, AMODE, Cfg3>(a, stream);
        case 4: return launch_gemm_cfg<T, EPI, ACT, AMODE, Cfg4>(a, stream);
        case 5: case 7: return launch_gemm_stagger<T, EPI, ACT, AMODE, Cfg1, 0>(a, stream);   // staggered two-group schedule (production)
        case 6: return launch_gemm_stagger<T, EPI, ACT, AMODE, Cfg1, 1>(a, stream);   // the same with the single-tile prologue of rounds 1-2 (A/B)
        case 8: return launch_gemm_cfg<T, EPI, ACT, AMODE, CfgS>(a, stream);
        case 9: return launch_gemm_cfg<T, EPI, ACT, AMODE, CfgS6>(a, stream);
        case 10: return launch_gemm_cfg<T, EPI, ACT, AMODE, CfgM>(a, stream);
        default: return launch_gemm_cfg<T, EPI, ACT, AMODE, Cfg0>(a, stream);
    }
}

// ---- low-bit correction phase (lmi_gemm_lo4): the production geometries only, mapped like the fp8 family -------------------------------------
// host-side row ranges of the row selection for the launch in flight on this thread (lmi_lo4.sel_ranges): the launcher that knows the tile
// height turns them into ranges of row tiles (gemm.h gemm_fill_sel)
thread_local const int* t_sel_ranges = nullptr;
thread_local int t_sel_n = 0;

template <typename T, int EPI, int ACT, typename C>
int launch_gemm_lo4_ring(const GemmArgs& a0, void* stream) {
    GemmArgs a = a0;
    gemm_fill_sel(a, t_sel_ranges, t_sel_n, C::BM);
    const int tiles = gemm_grid_size(a, C::BM, C::BN);
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(gemm_kernel<T, EPI, ACT, AMODE_PLAIN, C, T, true>, C::SMEM_LO4, attr_done);
    LMI_LAUNCH((gemm_kernel<T, EPI, ACT, AMODE_PLAIN, C, T, true>), dim3(tiles), dim3(C::NT), C::SMEM_LO4, stream, a);
    return check_launch("lmi_gemm_lo4");
}
template <typename T, int EPI, int ACT>
int launch_gemm_lo4(const GemmArgs& a, void* stream) {
    switch (choose_gemm_cfg(a)) {
        case 2: case 3: case 4: return launch_gemm_lo4_ring<T, EPI, ACT, Cfg2>(a, stream);
        case 5: case 6: case 7: case 1: {
            GemmArgs b = a;
            gemm_fill_sel(b, t_sel_ranges, t_sel_n, Cfg1::BM);
            const int tiles = gemm_grid_size(b, Cfg1::BM, Cfg1::BN);
            static std::atomic<uint64_t> attr_done{0};
            allow_big_lds(gemm_stagger_kernel<T, EPI, ACT, AMODE_PLAIN, Cfg1, 0, T, true>, Cfg1::SMEM_LO4, attr_done);
            LMI_LAUNCH((gemm_stagger_kernel<T, EPI, ACT, AMODE_PLAIN, Cfg1, 0, T, true>), dim3(tiles), dim3(Cfg1::NT), Cfg1::SMEM_LO4, stream, b);
            return check_launch("lmi_gemm_lo4");
        }
        case 8: case 9: return launch_gemm_lo4_ring<T, EPI, ACT, CfgS>(a, stream);
        case 10: return launch_gemm_lo4_ring<T, EPI, ACT, CfgM>(a, stream);
        default: return launch_gemm_lo4_ring<T, EPI, ACT, Cfg0>(a, stream);
    }
}
template <typename T>
int dispatch_gemm_lo4(const GemmArgs& a, int epi, int act, void* stream) {
    switch (epi) {
        case LMI_EPI_STORE:
            if (act == LMI_ACT_NONE) return launch_gemm_lo4<T, EPI_STORE_T, ACT_NONE>(a, stream);
            if (act == LMI_ACT_GELU_TANH) return launch_gemm_lo4<T, EPI_STORE_T, ACT_GELU_TANH>(a, stream);
            break;
        case LMI_EPI_RESIDUAL:
            if (act == LMI_ACT_NONE) return launch_gemm_lo4<T, EPI_RESID_F32, ACT_NONE>(a, stream);
            break;
        case LMI_EPI_STORE_F32:
            if (act == LMI_ACT_NONE) return launch_gemm_lo4<T, EPI_STORE_F32, ACT_NONE>(a, stream);
            break;
        case LMI_EPI_SWIGLU:
            if (act == LMI_ACT_NONE) return launch_gemm_lo4<T, EPI_SWIGLU_T, ACT_NONE>(a, stream);
            break;
        case LMI_EPI_QKV_ROPE:
            if (act == LMI_ACT_NONE) return launch_gemm_lo4<T, EPI_QKV_ROPE_T, ACT_NONE>(a, stream);
            break;
    }
    return fail(LMI_EINVAL, "lmi_gemm_lo4: unsupported epilogue/act combination (%d, %d)", epi, act);
}

// ---- fp8 operands (lmi_gemm_fp8): the production geometries only ---------------------------------------------------------------
template <typename T, int EPI, int ACT, typename C>
int launch_gemm_fp8_ring(const GemmArgs& a, void* stream) {
    const int tiles = ((a.M + C::BM - 1) / C::BM) * ((a.N + C::BN - 1) / C::BN);
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(gemm_kernel<T, EPI, ACT, AMODE_PLAIN, C, fp8_t>, C::SMEM_TOTAL, attr_done);
    LMI_LAUNCH((gemm_kernel<T, EPI, ACT, AMODE_PLAIN, C, fp8_t>), dim3(tiles), dim3(C::NT), C::SMEM_TOTAL, stream, a);
    return check_launch("lmi_gemm_fp8");
}
template <typename T, int EPI, int ACT>
int launch_gemm_fp8(const GemmArgs& a, void* stream) {
    switch (choose_gemm_cfg(a)) {
        case 2: case 3: case 4: return launch_gemm_fp8_ring<T, EPI, ACT, Cfg2>(a, stream);
        case 5: case 6: case 7: case 1: {
            const int tiles = ((a.M + Cfg1::BM - 1) / Cfg1::BM) * ((a.N + Cfg1::BN - 1) / Cfg1::BN);
            static std::atomic<uint64_t> attr_done{0};
            allow_big_lds(gemm_stagger_kernel<T, EPI, ACT, AMODE_PLAIN, Cfg1, 0, fp8_t>, Cfg1::SMEM_TOTAL, attr_done);
            LMI_LAUNCH((gemm_stagger_kernel<T, EPI, ACT, AMODE_PLAIN, Cfg1, 0, fp8_t>), dim3(tiles), dim3(Cfg1::NT), Cfg1::SMEM_TOTAL, stream, a);
            return check_launch("lmi_gemm_fp8");
        }
        case 8: case 9: case 10: return launch_gemm_fp8_ring<T, EPI, ACT, CfgS>(a, stream);
        default: return launch_gemm_fp8_ring<T, EPI, ACT, Cfg0>(a, stream);
    }
}
template <typename T>
int dispatch_gemm_fp8(const GemmArgs& a, int epi, int act, void* stream) {
    switch (epi) {
        case LMI_EPI_STORE:
            if (act == LMI_ACT_NONE) return launch_gemm_fp8<T, EPI_STORE_T, ACT_NONE>(a, stream);
            if (act == LMI_ACT_GELU_TANH) return launch_gemm_fp8<T, EPI_STORE_T, ACT_GELU_TANH>(a, stream);
            break;
        case LMI_EPI_RESIDUAL:
            if (act == LMI_ACT_NONE) return launch_gemm_fp8<T, EPI_RESID_F32, ACT_NONE>(a, stream);
            break;
        case LMI_EPI_STORE_F32:
            if (act == LMI_ACT_NONE) return launch_gemm_fp8<T, EPI_STORE_F32, ACT_NONE>(a, stream);
            break;
        case LMI_EPI_SWIGLU:
            if (act == LMI_ACT_NONE) return launch_gemm_fp8<T, EPI_SWIGLU_T, ACT_NONE>(a, stream);
            break;
    }
    return fail(LMI_EINVAL, "lmi_gemm_fp8: unsupported epilogue/act combination (%d, %d)", epi, act);
}

template <typename T>
int dispatch_gemm(const GemmArgs& a, int epi, int act, int amode, void* stream) {
    if (amode == LMI_A_PIXEL_SHUFFLE) {
        if (epi == LMI_EPI_STORE && act == LMI_ACT_GELU_ERF) return launch_gemm<T, EPI_STORE_T, ACT_GELU_ERF, AMODE_PIXSHUF>(a, stream);
        if (epi == LMI_EPI_STORE && act == LMI_ACT_NONE) return launch_gemm<T, EPI_STORE_T, ACT_NONE, AMODE_PIXSHUF>(a, stream);
        return fail(LMI_EINVAL, "lmi_gemm: pixel-shuffle A mode supports epilogue STORE with act none|gelu_erf");
    }
    switch (epi) {
        case LMI_EPI_STORE:
            if (act == LMI_ACT_NONE) return launch_gemm<T, EPI_STORE_T, ACT_NONE, AMODE_PLAIN>(a, stream);
            if (act == LMI_ACT_GELU_TANH) return launch_gemm<T, EPI_STORE_T, ACT_GELU_TANH, AMODE_PLAIN>(a, stream);
            if (act == LMI_ACT_GELU_ERF) return launch_gemm<T, EPI_STORE_T, ACT_GELU_ERF, AMODE_PLAIN>(a, stream);
            break;
        case LMI_EPI_RESIDUAL:
            if (act == LMI_ACT_NONE) return launch_gemm<T, EPI_RESID_F32, ACT_NONE, AMODE_PLAIN>(a, stream);
            break;
        case LMI_EPI_STORE_F32:
            if (act == LMI_ACT_NONE) return launch_gemm<T, EPI_STORE_F32, ACT_NONE, AMODE_PLAIN>(a, stream);
            if (act == LMI_ACT_GELU_TANH) return launch_gemm<T, EPI_STORE_F32, ACT_GELU_TANH, AMODE_PLAIN>(a, stream);   // split-operand mode: fc1 -> fp32
            break;
        case LMI_EPI_SWIGLU:
            if (act == LMI_ACT_NONE) return launch_gemm<T, EPI_SWIGLU_T, ACT_NONE, AMODE_PLAIN>(a, stream);
            break;
        case LMI_EPI_QKV_ROPE:
            if (act == LMI_ACT_NONE) return launch_gemm<T, EPI_QKV_ROPE_T, ACT_NONE, AMODE_PLAIN>(a, stream);
            break;
    }
    return fail(LMI_EINVAL, "lmi_gemm: unsupported epilogue/act combination (%d, %d)", epi, act);
}

// ---- attention dispatch ------------------------------------------------------------------------------------
template <typename T, int D, bool CAUSAL, bool USE_TR>
int launch_attn(const AttnArgs& a, int n_seq, int max_q, void* stream) {
    const int qblocks = (max_q + ATT_BQ - 1) / ATT_BQ;
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(attn_fwd_kernel<T, D, CAUSAL, USE_TR>, AttnGeom<D>::SMEM, attr_done);
    LMI_LAUNCH((attn_fwd_kernel<T, D, CAUSAL, USE_TR>), dim3(qblocks, a.n_heads, n_seq), dim3(ATT_THREADS),
               AttnGeom<D>::SMEM, stream, a);
    return check_launch("lmi_attn_varlen_fwd");
}
std::atomic<int> g_attn_lds_pad{0};                      // experiment knob: extra dynamic LDS per workgroup (lowers residency)
std::atomic<int> g_attn_stream_kv{1};        // decode, GQA-packed: non-temporal K / V tile loads (A/B knob attn.stream_kv)
std::atomic<int> g_gemv_plan{1};             // decode GEMV grid: 1 = a whole number of equal workgroups per CU when the shape allows (A/B knob gemv.plan)
std::atomic<int> g_attn_split_tiles{0};      // decode: 64-key tiles per split-KV workgroup; 0 = decode_splits chooses (A/B knob attn.decode_split_tiles)
std::atomic<int> g_attn_gqa_pack{1};         // decode: 1 = a workgroup's waves take the query heads of one kv head (A/B knob attn.gqa_pack)
std::atomic<int> g_attn_dma{1};                          // 1 = LDS-DMA kernel (production), 0 = register-staged kernel (cross-checks)

template <typename T, int D, bool CAUSAL>
int launch_attn_dma(const AttnArgs& a, int n_seq, int max_q, void* stream) {
    const int qblocks = (max_q + ATT_BQ - 1) / ATT_BQ;
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(attn_fwd_dma_kernel<T, D, CAUSAL>, 160 * 1024, attr_done);
    AttnArgs b = a;
    b.n_qblocks = qblocks;
    LMI_LAUNCH((attn_fwd_dma_kernel<T, D, CAUSAL>), dim3(qblocks * a.n_heads * n_seq), dim3(ATT_THREADS),
               AttnDmaGeom<D>::SMEM + g_attn_lds_pad, stream, b);
    return check_launch("lmi_attn_varlen_fwd");
}

// Software-pipelined attention (attention64.h): long self-attention prefills at head_dim 128 writing the plain 16-bit output.  OPT-IN
// (attn.rows64; 0 = attention.h's kernel, the default): measured on the Llama shape (S = 7187, f16; profiles/r04_attention64_study.txt)
// 1 = two 32-row blocks per wave, one wave per SIMD: 0.572 ms; 2 = one block per wave, two waves per SIMD: 0.505 ms; attention.h 0.442 ms
// — a single hand-interleaved instruction stream per wave loses to two independent workgroups per CU on this part (the issue port, not
// the matrix pipe, is what the softmax competes for).  attn.rows64_min: shortest max_seqlen_q that takes the opt-in kernel.
std::atomic<int> g_attn_rows64{0};
std::atomic<int> g_attn_rows64_min{1024};
template <typename T, bool CAUSAL, int NBLK>
int launch_attn_r64(const AttnArgs& a, int n_seq, int max_q, void* stream) {
    typedef Attn64Geom<NBLK> G;
    const int qblocks = (max_q + G::BQ - 1) / G::BQ;
    static std::atomic<uint64_t> attr_done{0};
    allow_big_lds(attn_fwd_r64_kernel<T, CAUSAL, NBLK>, G::SMEM, attr_done);
    AttnArgs b = a;
    b.n_qblocks = qblocks;
    LMI_LAUNCH((attn_fwd_r64_kernel<T, CAUSAL, NBLK>), dim3(qblocks * a.n_heads * n_seq), dim3(G::NT), G::SMEM, stream, b);
    return check_launch("lmi_attn_varlen_fwd");
}
template <typename T, int D>
bool attn_r64_applies(const AttnArgs& a, int max_q, int use_tr) {
    if (D != 128 || !use_tr || !g_attn_dma.load() || !g_attn_rows64.load() || max_q < g_attn_rows64_min.load()) return false;
    if (!a.out || a.out_fp8 || a.out_f32 || a.out4 || a.n_splits > 1 || a.k_len || a.gqa_pack || a.check_k_extent) return false;   // self-attention, 16-bit output only
    // the ring requests tiles up to two past the end (range-checked to zeros): their 32-bit offsets must not wrap
    return ((long)(max_q + 4 * ATT_BKV) * a.ldk + D) * 2 < (1L << 32) && ((long)(max_q + 4 * ATT_BKV) * a.ldv + D) * 2 < (1L << 32);
}

template <typename T, int D>
int dispatch_attn(const AttnArgs& a, int n_seq, int max_q, int causal, int use_tr, void* stream) {
    if constexpr (D == 128) {
        if (attn_r64_applies<T, D>(a, max_q, use_tr)) {
            if (g_attn_rows64.load() == 1)
                return causal ? launch_attn_r64<T, true, 2>(a, n_seq, max_q, stream) : launch_attn_r64<T, false, 2>(a, n_seq, max_q, stream);
            return causal ? launch_attn_r64<T, true, 1>(a, n_seq, max_q, stream) : launch_attn_r64<T, false, 1>(a, n_seq, max_q, stream);
        }
    }
    if (use_tr && g_attn_dma)
        return causal ? launch_attn_dma<T, D, true>(a, n_seq, max_q, stream) : launch_attn_dma<T, D, false>(a, n_seq, max_q, stream);
    if (causal) return use_tr ? launch_attn<T, D, true, true>(a, n_seq, max_q, stream)
                              : launch_attn<T, D, true, false>(a, n_seq, max_q, stream);
    return use_tr ? launch_attn<T, D, false, true>(a, n_seq, max_q, stream)
                  : launch_attn<T, D, false, false>(a, n_seq, max_q, stream);
}

// units per workgroup of the K-split kernel: a multiple of the units of one pass.  Preferred: a grid of exactly 3, 2 or 1 workgroups
// per CU (768 / 512 / 256) with the same number of passes in every workgroup — the stream then ends everywhere at once instead of in a
// partial last round; otherwise about a thousand workgroups.  At most 128 partial slots per wave.
static int gemv_upb(int units, int per_pass, int max_units) {
    if (g_gemv_plan.load()) {
        const int chunks = (units + per_pass - 1) / per_pass;
        for (int g : {768, 512, 256})
            if (units % per_pass == 0 && chunks % g == 0 && (chunks / g) * per_pass <= max_units) return (chunks / g) * per_pass;
    }
    int upb = per_pass;
    while ((units + upb - 1) / upb > 1024 && upb * 2 <= max_units) upb *= 2;
    return upb;
}
template <typename T, int EPI, bool NORM>
int launch_gemv(const void* W, const void* x, const float* gamma, float eps, const float* bias, void* out, int N, int K, int ldw,
                void* stream, const RopeEpi& rp = RopeEpi()) {
    constexpr bool PAIR = (EPI == GEMV_SWIGLU_T || EPI == GEMV_QKV_ROPE_T);
    const int rows = PAIR ? N / 2 : N;
    constexpr int RW = PAIR ? 2 : 1;
    if (K == 4096) {
        const int upb = gemv_upb(rows, 8 / RW, 128 / RW);
        LMI_LAUNCH((gemv_split_kernel<T, EPI, 2, 8, NORM>), dim3((rows + upb - 1) / upb), dim3(256), 0, stream, (const T*)W, x, gamma, eps,
                   bias, out, N, K, ldw, upb, rp);
        return check_launch("lmi_gemv");
    }
    if (EPI == GEMV_QKV_ROPE_T) return fail(LMI_EINVAL, "lmi_gemv_rmsnorm_rope: K = %d is not a supported hidden size (4096)", K);
    if (K == 14336 && !NORM) {
        const int upb = gemv_upb(rows, 4 / RW, 128 / RW);
        LMI_LAUNCH((gemv_split_kernel<T, EPI, 7, 4, false>), dim3((rows + upb - 1) / upb), dim3(256), 0, stream, (const T*)W, x, gamma, eps,
                   bias, out, N, K, ldw, upb, rp);
        return check_launch("lmi_gemv");
    }
    if (NORM) return fail(LMI_EINVAL, "lmi_gemv_rmsnorm: K = %d is not a supported hidden size (4096)", K);
    const int grid = grid_for(rows, 4);
    if (K <= 4096) {
        LMI_LAUNCH((gemv_kernel<T, EPI, 8>), dim3(grid), dim3(256), 0, stream, (const T*)W, (const T*)x, bias, out, N, K, ldw);
    } else {
        LMI_LAUNCH((gemv_kernel<T, EPI, 28>), dim3(grid), dim3(256), 0, stream, (const T*)W, (const T*)x, bias, out, N, K, ldw);
    }
    return check_launch("lmi_gemv");
}
template <typename T, bool NORM>
int dispatch_gemv(const void* W, const void* x, const float* gamma, float eps, const float* bias, void* out, int N, int K, int ldw,
                  int epi, void* stream) {
    switch (epi) {
        case 0: return launch_gemv<T, GEMV_STORE_F32, NORM>(W, x, gamma, eps, bias, out, N, K, ldw, stream);
        case 1: return launch_gemv<T, GEMV_STORE_T, NORM>(W, x, gamma, eps, bias, out, N, K, ldw, stream);
        case 2: return launch_gemv<T, GEMV_RESID_F32, NORM>(W, x, gamma, eps, bias, out, N, K, ldw, stream);
        case 3: return launch_gemv<T, GEMV_SWIGLU_T, NORM>(W, x, gamma, eps, bias, out, N, K, ldw, stream);
    }
    return fail(LMI_EINVAL, "lmi_gemv: bad epilogue %d", epi);
}
template <typename T, bool RMS>
int norm_impl(const float* x, const float* w, const float* b, void* out, int M, int D, int ldx, int ldo,
                     float eps, void* stream, const char* what, float out_scale = 1.0f) {
    if (!x || !w || !out || (!RMS && !b) || M < 0 || D <= 0 || (D & 7) || D > 4096 || (ldx & 3) || (ldo & 7) ||
        !aligned16(x) || !aligned16(w) || !aligned16(out))
        return fail(LMI_EINVAL, "%s: bad argument (M=%d D=%d ldx=%d ldo=%d; D%%8==0, D<=4096)", what, M, D, ldx, ldo);
    if (M == 0) return LMI_OK;
    if (M <= 32) {                                                  // a handful of rows (decode): one workgroup per row (norm_rows_kernel)
        if (D <= 2048) LMI_LAUNCH((norm_rows_kernel<T, RMS, 1>), dim3(M), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, out_scale);
        else LMI_LAUNCH((norm_rows_kernel<T, RMS, 2>), dim3(M), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, out_scale);
        return check_launch(what);
    }
    const int grid = (M + 3) / 4;
    if (D <= 1536) LMI_LAUNCH((norm_kernel<T, RMS, 3>), dim3(grid), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, out_scale);
    else LMI_LAUNCH((norm_kernel<T, RMS, 8>), dim3(grid), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, out_scale);
    return check_launch(what);
}

template <typename T, bool RMS>
int norm_lo4_impl(const float* x, const float* w, const float* b, void* out, NormLo4 lo, int M, int D, int ldx, int ldo, float eps, void* stream) {
    const int grid = (M + 3) / 4;
    if (D <= 1536) LMI_LAUNCH((norm_kernel<T, RMS, 3, true>), dim3(grid), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, 1.0f, lo);
    else LMI_LAUNCH((norm_kernel<T, RMS, 8, true>), dim3(grid), dim3(256), 0, stream, x, w, b, (T*)out, M, D, ldx, ldo, eps, 1.0f, lo);
    return check_launch("lmi_norm_lo4");
}
template <typename T, typename DT>
int add_rmsnorm_impl(float* x, const void* delta, const float* w, void* out, int M, int D, int ldx, int ldd, int ldo, float eps,
                            void* stream) {
    const int grid = (M + 3) / 4;
    if (D <= 1536) LMI_LAUNCH((add_rmsnorm_kernel<T, DT, 3>), dim3(grid), dim3(256), 0, stream, x, (const DT*)delta, w, (T*)out, M, D, ldx, ldd, ldo, eps);
    else LMI_LAUNCH((add_rmsnorm_kernel<T, DT, 8>), dim3(grid), dim3(256), 0, stream, x, (const DT*)delta, w, (T*)out, M, D, ldx, ldd, ldo, eps);
    return check_launch("lmi_add_rmsnorm");
}

template <typename T>
int im2col_impl(const void* in, int from_u8, void* out, int n_tiles, int H, int W, int P, int ldo, int grid, void* stream) {
    if (from_u8) LMI_LAUNCH((im2col_kernel<T, true>), dim3(grid), dim3(256), 0, stream, in, (T*)out, n_tiles, H, W, P, ldo);
    else LMI_LAUNCH((im2col_kernel<T, false>), dim3(grid), dim3(256), 0, stream, in, (T*)out, n_tiles, H, W, P, ldo);
    return check_launch("lmi_preprocess_tiles");
}
template <typename T>
int rope_impl(void* qkv, int S, int ld, int nq, int nkv, int D, const float* c, const float* s, void* kc, void* vc,
              int ldc, int pos0, const int* pos_dev, int grid, void* stream) {
    LMI_LAUNCH((rope_kernel<T>), dim3(grid), dim3(256), 0, stream, (T*)qkv, S, ld, nq, nkv, D, c, s, (T*)kc, (T*)vc, ldc, pos0, pos_dev);
    return check_launch("lmi_rope_qk");
}
template <typename T>
int patch_embed_impl(const PatchEmbedArgs& a, int from_u8, void* stream) {
    static std::atomic<uint64_t> done_u8{0}, done_f32{0};
    const int grid = ((a.M + PE_BM - 1) / PE_BM) * (a.N / PE_BN);
    if (from_u8) {
        allow_big_lds(patch_embed_kernel<T, true>, PE_SMEM, done_u8);
        LMI_LAUNCH((patch_embed_kernel<T, true>), dim3(grid), dim3(256), PE_SMEM, stream, a);
    } else {
        allow_big_lds(patch_embed_kernel<T, false>, PE_SMEM, done_f32);
        LMI_LAUNCH((patch_embed_kernel<T, false>), dim3(grid), dim3(256), PE_SMEM, stream, a);
    }
    return check_launch("lmi_patch_embed");
}
template <typename T>
int kv_append_impl(const void* k, const void* v, void* kc, void* vc, int S, int width, int ld_src, int ld_cache, int pos0, void* stream) {
    const long total = (long)S * (width >> 3) * 2;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    LMI_LAUNCH((kv_append_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)k, (const T*)v, (T*)kc, (T*)vc, S, width, ld_src, ld_cache, pos0);
    return check_launch("lmi_kv_append");
}
template <typename T>
int merge_impl(const int64_t* ids, const int64_t* src, const void* table, const float* feats, float* out, int S, int D,
               int ld_feats, void* stream) {
    LMI_LAUNCH((embed_merge_kernel<T>), dim3(S), dim3(256), 0, stream, (const long*)ids, (const long*)src, (const T*)table,
               feats, out, D, ld_feats);
    return check_launch("lmi_embed_merge");
}

}  // namespace

#define LMI_DISPATCH_T(dtype, CALL_F16, CALL_BF16)                                   \
    do {                                                                              \
        if ((dtype) == LMI_F16) return CALL_F16;                                      \
        if ((dtype) == LMI_BF16) return CALL_BF16;                                    \
        return fail(LMI_EINVAL, "%s: dtype must be LMI_F16 or LMI_BF16", __func__);   \
    } while (0)

// ---- decode attention: split-KV LDS-DMA kernel + merge ------------------------------------------------------------------
// Key tiles per workgroup: the largest of 8, 4, 2, 1 (512 ... 64 keys) that still gives about one workgroup per CU for ONE sequence
// (grid_heads workgroups per split: the kv heads when the blocks are GQA-packed, else the query heads), at most 64 splits (the merge
// kernel gives every split one lane).  A function of the launch shape only — never of device data (graph capture) or of the number
// of sequences (a pooled launch and its sequences one by one split alike and agree bit for bit).
static int decode_splits(int max_seqlen_k, int grid_heads, int* split_tiles) {
    const int tiles = (max_seqlen_k + ATT_BKV - 1) / ATT_BKV;
    int st = g_attn_split_tiles.load() > 0 ? g_attn_split_tiles.load() : 8;
    if (g_attn_split_tiles.load() <= 0)
        while (st > 1 && (long)grid_heads * ((tiles + st - 1) / st) < 256) st >>= 1;
    while ((tiles + st - 1) / st > 64) st *= 2;
    *split_tiles = st;
    const int n = (tiles + st - 1) / st;
    return n < 2 ? 2 : n;      // a single chunk also goes through the partial + merge pair: one code path
}
static int decode_grid_heads(int n_heads, int n_kv_heads, int max_q) {
    return (g_attn_gqa_pack.load() && n_heads == 4 * n_kv_heads && max_q <= 32) ? n_kv_heads : n_heads;
}

template <typename T>
int attn_decode_impl(AttnArgs a, int n_seq, int max_q, int q_rows, void* out, int ldo, void* stream, int lo_rows = 0) {
    static std::atomic<uint64_t> attr_done{0};
    static std::atomic<uint64_t> attr_done_s{0};
    allow_big_lds(attn_fwd_dma_kernel<T, 128, true>, 160 * 1024, attr_done);
    allow_big_lds(attn_fwd_dma_kernel<T, 128, true, true>, 160 * 1024, attr_done_s);
    // GQA-packed blocks when a kv head serves exactly 4 query heads (Llama-3.1-8B, Mistral-7B) and a sequence has at most 32 query rows
    a.gqa_pack = decode_grid_heads(a.n_heads, a.n_kv_heads, max_q) != a.n_heads ? 1 : 0;
    a.n_qblocks = a.gqa_pack ? (max_q + 31) / 32 : (max_q + ATT_BQ - 1) / ATT_BQ;
    const dim3 grid(a.n_qblocks * (a.gqa_pack ? a.n_kv_heads : a.n_heads) * n_seq * a.n_splits);
    // GQA-packed blocks read every K / V tile exactly once: non-temporal loads; head-per-block launches re-read them from L2 (4 query heads)
    if (a.gqa_pack && g_attn_stream_kv.load()) LMI_LAUNCH((attn_fwd_dma_kernel<T, 128, true, true>), grid, dim3(ATT_THREADS), AttnDmaGeom<128>::SMEM, stream, a);
    else LMI_LAUNCH((attn_fwd_dma_kernel<T, 128, true>), grid, dim3(ATT_THREADS), AttnDmaGeom<128>::SMEM, stream, a);
    const long items = (long)q_rows * a.n_heads;
    LMI_LAUNCH((attn_combine_kernel<T, 128>), dim3((unsigned)items), dim3(256), 0, stream, (const float*)a.part_o,
               (const float*)a.part_ml, (T*)out, a.cu_q, n_seq, a.n_heads, a.n_splits, a.part_rows, ldo, a.scale, lo_rows);
    return check_launch("lmi_attn_decode_fwd");
}

std::atomic<int> g_skinny_coalesce{1};       // nn.Linear-layout weights of the M <= 16 kernel: 1 = coalescing lane order + ds_bpermute (LAYOUT 2), 0 = MFMA lane order
template <typename T, int LAYOUT>
int skinny_impl(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, void* stream,
                const RopeEpi& rp = RopeEpi(), const SkinnyNorm& nm = SkinnyNorm()) {
    const int units = (epilogue == LMI_SKINNY_SWIGLU || epilogue == 4) ? N / 32 : N / 16;
#define LMI_SK(E) LMI_LAUNCH((skinny_gemm_kernel<T, E, LAYOUT>), dim3(units), dim3(512), 0, stream, (const T*)W, (const T*)X, out, M, N, K, ldw, ldx, ldo, rp, nm)
    switch (epilogue) {
        case LMI_SKINNY_STORE: LMI_SK(SK_STORE_T); break;
        case LMI_SKINNY_RESIDUAL: LMI_SK(SK_RESID_F32); break;
        case LMI_SKINNY_SWIGLU: LMI_SK(SK_SWIGLU_T); break;
        case 4: LMI_SK(SK_QKV_ROPE_T); break;                        // lmi_rope_qkv_skinny only
        default: LMI_SK(SK_STORE_F32); break;
    }
#undef LMI_SK
    return check_launch("lmi_gemm_skinny");
}

// the folded-norm arguments of lmi_gemm_skinny_ex / lmi_rope_qkv_skinny, checked
int skinny_norm_args(const char* who, SkinnyNorm& nm, int M, int N, int epilogue, const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps,
                     void* norm_out, int ld_norm, const float* norm_gamma, float* rowsq_out) {
    nm = SkinnyNorm();
    if (rowsq_in) {
        if (rowsq_parts <= 0 || norm_dim <= 0) return fail(LMI_EINVAL, "%s: rowsq_in needs rowsq_parts > 0 and norm_dim > 0", who);
        // the producer (a residual lmi_gemm_skinny_ex over the norm_dim-wide stream) writes one partial per 16-column workgroup and row
        if (rowsq_parts != norm_dim / 16 || (norm_dim % 16) || ((uintptr_t)rowsq_in & 3))
            return fail(LMI_EINVAL, "%s: rowsq_parts %d must be norm_dim / 16 = %d (4-byte aligned partials)", who, rowsq_parts, norm_dim / 16);
        nm.rowsq_in = rowsq_in; nm.parts_in = rowsq_parts; nm.inv_dim = 1.0f / (float)norm_dim; nm.eps = norm_eps;
    }
    if (norm_out || rowsq_out || norm_gamma) {
        if (epilogue != LMI_SKINNY_RESIDUAL || !norm_out || !rowsq_out || !norm_gamma || ld_norm < N || ((uintptr_t)rowsq_out & 3) || ((uintptr_t)norm_out & 1) ||
            ((uintptr_t)norm_gamma & 3))
            return fail(LMI_EINVAL, "%s: norm_out / norm_gamma / rowsq_out go together, with the residual epilogue (ld_norm >= N, aligned pointers)", who);
        nm.norm_out = norm_out; nm.ld_norm = ld_norm; nm.gamma = norm_gamma; nm.rowsq_out = rowsq_out;
    }
    (void)M;
    return LMI_OK;
}

template <typename T>
int rope_rows_impl(void* qkv, int S, int ld, int nq, int nkv, int D, const float* c, const float* s, void* kc, void* vc, int ldc,
                          long cache_stride, const int* pos, int grid, void* stream) {
    LMI_LAUNCH((rope_rows_kernel<T>), dim3(grid), dim3(256), 0, stream, (T*)qkv, S, ld, nq, nkv, D, c, s, (T*)kc, (T*)vc, ldc, cache_stride, pos);
    return check_launch("lmi_rope_qk_rows");
}

int rope_entry(const char* who, void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_table,
                      const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, const int* pos_dev,
                      int dtype, void* stream) {
    if (!qkv || !cos_table || !sin_table || S < 0 || (head_dim & 15) || (ld & 7) || ((k_cache || v_cache) && (ld_cache & 7)) ||
        ((k_cache != nullptr) != (v_cache != nullptr)))
        return fail(LMI_EINVAL, "%s: bad argument", who);
    if (S == 0) return LMI_OK;
    const long work = (long)S * ((n_q_heads + n_kv_heads) * (head_dim / 16) + (v_cache ? n_kv_heads * head_dim / 8 : 0));
    const int grid = grid_for(work, 256);
    LMI_DISPATCH_T(dtype, (rope_impl<f16_t>(qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_table, sin_table, k_cache, v_cache, ld_cache, cache_pos0, pos_dev, grid, stream)),
                   (rope_impl<bf16_t>(qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_table, sin_table, k_cache, v_cache, ld_cache, cache_pos0, pos_dev, grid, stream)));
}


template <typename T>
static int attn_fp8_impl(const AttnFp8Args& a, int n_seq, int causal, void* stream) {
    const dim3 grid((unsigned)(a.n_qblocks * a.n_heads * n_seq));
    if (causal) LMI_LAUNCH((attn_fwd_fp8_kernel<T, true>), grid, dim3(ATT_THREADS), 4 * ATT8_IMG, stream, a);
    else LMI_LAUNCH((attn_fwd_fp8_kernel<T, false>), grid, dim3(ATT_THREADS), 4 * ATT8_IMG, stream, a);
    return check_launch("lmi_attn_fp8_fwd");
}

extern "C" {

const char* lmi_last_error(void) { return g_err; }
int lmi_abi_version(void) { return 1; }

int lmi_set_option(const char* key, int value) {
    if (!key) return fail(LMI_EINVAL, "lmi_set_option: null key");
    if (!strcmp(key, "gemm.config")) {
        if (value < -1 || value >= kNumGemmCfg) return fail(LMI_EINVAL, "lmi_set_option: gemm.config in [-1, %d)", kNumGemmCfg);
        g_gemm_cfg = value;
        return LMI_OK;
    }
    if (!strcmp(key, "gemm.group_m")) {
        if (value < 1 || value > 64) return fail(LMI_EINVAL, "lmi_set_option: gemm.group_m in [1, 64]");
        g_gemm_group_m = value;
        return LMI_OK;
    }
    if (!strcmp(key, "gemm.order")) { g_gemm_order = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "gemm.sel_ragged_last")) { g_sel_keep_ragged_last = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "gemm.auto_small")) { g_gemm_auto_small = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "gemm.mid_m")) { g_gemm_mid_m = value ? 1 : 0; return LMI_OK; }
    {
        struct { const char* key; std::atomic<int>* var; } classes[] = {{"gemm.wide", &g_gemm_wide}, {"gemm.short_k", &g_gemm_short},
                                                          {"gemm.narrow_n", &g_gemm_narrow}, {"gemm.small", &g_gemm_small}};
        for (auto& c : classes)
            if (!strcmp(key, c.key)) {
                if (value < 0 || value > 7) return fail(LMI_EINVAL, "lmi_set_option: %s in [0, 7]", key);
                *c.var = value;
                return LMI_OK;
            }
    }
    if (!strcmp(key, "skinny.coalesce")) { g_skinny_coalesce = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "attn.dma")) { g_attn_dma = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "attn.rows64")) {
        if (value < 0 || value > 2) return fail(LMI_EINVAL, "lmi_set_option: attn.rows64 in {0, 1, 2}");
        g_attn_rows64 = value;
        return LMI_OK;
    }
    if (!strcmp(key, "attn.rows64_min")) { g_attn_rows64_min = value < 0 ? 0 : value; return LMI_OK; }
    if (!strcmp(key, "attn.stream_kv")) { g_attn_stream_kv = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "gemv.plan")) { g_gemv_plan = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "attn.gqa_pack")) { g_attn_gqa_pack = value ? 1 : 0; return LMI_OK; }
    if (!strcmp(key, "attn.decode_split_tiles")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(LMI_EINVAL, "lmi_set_option: attn.decode_split_tiles in {0, 1, 2, 4, 8}");
        g_attn_split_tiles = value;
        return LMI_OK;
    }
    if (!strcmp(key, "attn.lds_pad")) { g_attn_lds_pad = value < 0 ? 0 : (value > 90 * 1024 ? 90 * 1024 : value); return LMI_OK; }
    return fail(LMI_EINVAL, "lmi_set_option: unknown key %s", key);
}

int lmi_debug_copy(const void* src, void* dst, int64_t bytes, int n_workgroups, void* stream) {
    if (!src || !dst || bytes < 0 || (bytes & 15) || n_workgroups <= 0 || !aligned16(src) || !aligned16(dst))
        return fail(LMI_EINVAL, "lmi_debug_copy: bad argument (bytes %% 16 == 0, 16-byte aligned pointers, n_workgroups > 0)");
    if (bytes == 0) return LMI_OK;
    LMI_LAUNCH(debug_copy_kernel, dim3(n_workgroups), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, (long)(bytes >> 4));
    return check_launch("lmi_debug_copy");
}

int lmi_fill_synthetic(void* out, int64_t n, uint32_t seed, int kind, int out_dtype, void* stream) {
    if (!out || n < 0 || kind < 0 || kind > 2) return fail(LMI_EINVAL, "lmi_fill_synthetic: bad argument");
    if (n == 0) return LMI_OK;
    const int grid = grid_for(n, 256);
    if (out_dtype == LMI_F16) LMI_LAUNCH((fill_synth_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, (f16_t*)out, (long)n, seed, kind);
    else if (out_dtype == LMI_BF16) LMI_LAUNCH((fill_synth_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, (bf16_t*)out, (long)n, seed, kind);
    else if (out_dtype == LMI_F32) LMI_LAUNCH((fill_synth_kernel<float>), dim3(grid), dim3(256), 0, stream, (float*)out, (long)n, seed, kind);
    else return fail(LMI_EINVAL, "lmi_fill_synthetic: bad out_dtype %d", out_dtype);
    return check_launch("lmi_fill_synthetic");
}

int lmi_resample_u8(const void* src, void* dst, int axis, int out_rows, int out_cols, int src_pitch, int dst_pitch,
                    const int* bounds, const int* taps, int ksize, void* stream) {
    if (!src || !dst || !bounds || !taps) return fail(LMI_EINVAL, "lmi_resample_u8: null pointer");
    if ((axis != 0 && axis != 1) || out_rows < 0 || out_cols < 0 || ksize <= 0 || src_pitch <= 0 || dst_pitch < out_cols * 3)
        return fail(LMI_EINVAL, "lmi_resample_u8: bad arguments (axis=%d rows=%d cols=%d ksize=%d)", axis, out_rows, out_cols, ksize);
    if (out_rows == 0 || out_cols == 0) return LMI_OK;
    const int grid = grid_for((long)out_rows * out_cols, 256);
    if (axis == 0)
        LMI_LAUNCH((resample_u8_kernel<0>), dim3(grid), dim3(256), 0, stream, (const uint8_t*)src, (uint8_t*)dst, out_rows, out_cols,
                   src_pitch, dst_pitch, bounds, taps, ksize);
    else
        LMI_LAUNCH((resample_u8_kernel<1>), dim3(grid), dim3(256), 0, stream, (const uint8_t*)src, (uint8_t*)dst, out_rows, out_cols,
                   src_pitch, dst_pitch, bounds, taps, ksize);
    return check_launch("lmi_resample_u8");
}

int lmi_preprocess_images(const void* in, int from_u8, void* out, int n_images, int height, int width, int patch, int ldo,
                          int dtype, void* stream) {
    if (!in || !out || n_images < 0 || patch <= 0 || height < patch || width < patch || ldo < 3 * patch * patch || (ldo & 7))
        return fail(LMI_EINVAL, "lmi_preprocess_images: bad shape (H=%d W=%d P=%d ldo=%d)", height, width, patch, ldo);
    if (n_images == 0) return LMI_OK;
    const int grid = grid_for((long)n_images * (height / patch) * (width / patch) * ldo, 256);
    LMI_DISPATCH_T(dtype, (im2col_impl<f16_t>(in, from_u8, out, n_images, height, width, patch, ldo, grid, stream)),
                   (im2col_impl<bf16_t>(in, from_u8, out, n_images, height, width, patch, ldo, grid, stream)));
}

int lmi_preprocess_tiles(const void* in, int from_u8, void* out, int n_tiles, int image_size, int patch, int ldo,
                         int dtype, void* stream) {
    if (image_size <= 0 || patch <= 0 || image_size % patch)
        return fail(LMI_EINVAL, "lmi_preprocess_tiles: bad shape (S=%d P=%d ldo=%d)", image_size, patch, ldo);
    return lmi_preprocess_images(in, from_u8, out, n_tiles, image_size, image_size, patch, ldo, dtype, stream);
}

int lmi_patch_embed(const void* pixels, int from_u8, const void* W, const float* bias, const float* pos_emb, float* out, int n_tiles,
                    int image_size, int patch, int N, int ldw, int ldo, int dtype, void* stream) {
    if (!pixels || !W || !bias || !pos_emb || !out) return fail(LMI_EINVAL, "lmi_patch_embed: null pointer");
    if (n_tiles < 0 || image_size <= 0 || patch < 3 || image_size % patch || N <= 0 || (N % 128) || (ldo & 3) || ldo < N)
        return fail(LMI_EINVAL, "lmi_patch_embed: bad shape (n=%d S=%d P=%d N=%d ldo=%d; P >= 3, S %% P == 0, N %% 128 == 0)", n_tiles, image_size, patch, N,
                    ldo);
    PatchEmbedArgs a;
    a.pix = pixels; a.W = W; a.bias = bias; a.pos = pos_emb; a.out = out;
    a.S = image_size; a.P = patch; a.G = image_size / patch; a.N = N;
    a.RP = (3 * patch + 7) & ~7;
    a.KP = (patch * a.RP + 63) & ~63;
    if (a.KP / PE_BK > PE_MAX_KT) return fail(LMI_EINVAL, "lmi_patch_embed: patch size %d is beyond the kernel's gather table", patch);
    a.ldw = ldw; a.ldo = ldo;
    const long M = (long)n_tiles * a.G * a.G;
    if (M > 0x7fffffffL) return fail(LMI_EINVAL, "lmi_patch_embed: too many patches");
    a.M = (int)M;
    if (ldw < a.KP || (ldw & 7) || !aligned16(W) || !aligned16(out) || !aligned16(bias) || !aligned16(pos_emb))
        return fail(LMI_EINVAL, "lmi_patch_embed: weight rows must hold %d elements ((ky, kx, c) order, pixel rows padded to %d) and be 16-byte aligned",
                    a.KP, a.RP);
    if (M == 0) return LMI_OK;
    LMI_DISPATCH_T(dtype, patch_embed_impl<f16_t>(a, from_u8, stream), patch_embed_impl<bf16_t>(a, from_u8, stream));
}

int lmi_kv_append(const void* k, const void* v, void* k_cache, void* v_cache, int S, int width, int ld_src, int ld_cache, int cache_pos0,
                  int dtype, void* stream) {
    if (!k || !v || !k_cache || !v_cache || S < 0 || width <= 0 || (width & 7) || (ld_src & 7) || (ld_cache & 7) || cache_pos0 < 0 ||
        !aligned16(k) || !aligned16(v) || !aligned16(k_cache) || !aligned16(v_cache))
        return fail(LMI_EINVAL, "lmi_kv_append: bad argument (S=%d width=%d ld_src=%d ld_cache=%d pos0=%d)", S, width, ld_src, ld_cache, cache_pos0);
    if (S == 0) return LMI_OK;
    LMI_DISPATCH_T(dtype, kv_append_impl<f16_t>(k, v, k_cache, v_cache, S, width, ld_src, ld_cache, cache_pos0, stream),
                   kv_append_impl<bf16_t>(k, v, k_cache, v_cache, S, width, ld_src, ld_cache, cache_pos0, stream));
}

int lmi_layernorm(const float* x, const float* w, const float* b, void* out, int M, int D, int ldx, int ldo, float eps,
                  int dtype, void* stream) {
    if (dtype == LMI_F32) return norm_impl<float, false>(x, w, b, out, M, D, ldx, ldo, eps, stream, "lmi_layernorm");
    if (dtype == LMI_FP8) return fail(LMI_EINVAL, "lmi_layernorm: fp8 output needs a scale: use lmi_norm_fp8");
    LMI_DISPATCH_T(dtype, (norm_impl<f16_t, false>(x, w, b, out, M, D, ldx, ldo, eps, stream, "lmi_layernorm")),
                   (norm_impl<bf16_t, false>(x, w, b, out, M, D, ldx, ldo, eps, stream, "lmi_layernorm")));
}

int lmi_rmsnorm(const float* x, const float* w, void* out, int M, int D, int ldx, int ldo, float eps, int dtype,
                void* stream) {
    if (dtype == LMI_F32) return norm_impl<float, true>(x, w, nullptr, out, M, D, ldx, ldo, eps, stream, "lmi_rmsnorm");
    if (dtype == LMI_FP8) return fail(LMI_EINVAL, "lmi_rmsnorm: fp8 output needs a scale: use lmi_norm_fp8");
    LMI_DISPATCH_T(dtype, (norm_impl<f16_t, true>(x, w, nullptr, out, M, D, ldx, ldo, eps, stream, "lmi_rmsnorm")),
                   (norm_impl<bf16_t, true>(x, w, nullptr, out, M, D, ldx, ldo, eps, stream, "lmi_rmsnorm")));
}

// everything lmi_gemm / lmi_gemm_ex / lmi_rmsnorm_rope share: argument checks, buffer extents, dispatch
struct GemmExtras {
    const float* rowsq_in = nullptr; int rowsq_parts = 0; int norm_dim = 0; float norm_eps = 0.f;
    void* norm_out = nullptr; const float* norm_gamma = nullptr; float* rowsq_out = nullptr; int ld_norm = 0;
    const float* rope_cos = nullptr; const float* rope_sin = nullptr; void* k_cache = nullptr; void* v_cache = nullptr;
    int ld_cache = 0, cache_pos0 = 0, rope_q = 0, rope_k = 0;
};
static int gemm_entry(const char* who, const void* A, const void* W, void* out, const float* bias, const float* addmat, const int* add_rows,
                      const int* row_map, int M, int N, int K, int lda, int ldw, int ldo, int add_period, int epilogue, int act, int a_mode,
                      int ps_grid, int dtype, void* stream, const GemmExtras& x, const lmi_lo4* lo = nullptr) {
    if (!A || !W || !out) return fail(LMI_EINVAL, "%s: null pointer", who);
    // ldw = LMI_LDW_PACKED(K) = -K: W in the operand order of lmi_gemm_skinny (one copy of the weights for prefill and decode)
    const bool w_packed = ldw < 0;
    if (w_packed) {
        if (ldw != -K || (K % 128) || (N % 16) || a_mode != LMI_A_PLAIN)
            return fail(LMI_EINVAL, "%s: packed W needs ldw == -K, K %% 128 == 0, N %% 16 == 0 and a plain A (N=%d K=%d ldw=%d)", who, N, K, ldw);
        ldw = K;
    }
    if (M < 0 || N <= 0 || K <= 0 || (N % 128) || (K % GEMM_BK))
        return fail(LMI_EINVAL, "%s: need N %% 128 == 0 and K %% 64 == 0 (M=%d N=%d K=%d)", who, M, N, K);
    if ((lda & 7) || (ldw & 7) || (ldo & 3) || !aligned16(A) || !aligned16(W) || !aligned16(out) ||
        (bias && !aligned16(bias)) || (addmat && !aligned16(addmat)))
        return fail(LMI_EINVAL, "%s: pointers must be 16-byte aligned, lda/ldw multiples of 8, ldo of 4", who);
    if (addmat && !add_rows && add_period <= 0) return fail(LMI_EINVAL, "%s: addmat needs add_period > 0 or add_rows", who);
    if (a_mode == LMI_A_PIXEL_SHUFFLE) {
        if (ps_grid <= 0 || (ps_grid & 1) || (K & 3) || ((K / 4) % GEMM_BK) || (M % ((ps_grid / 2) * (ps_grid / 2))))
            return fail(LMI_EINVAL, "%s: pixel-shuffle needs even grid, (K/4) %% 64 == 0, M %% (G/2)^2 == 0", who);
    } else if (a_mode != LMI_A_PLAIN) {
        return fail(LMI_EINVAL, "%s: bad a_mode %d", who, a_mode);
    }
    if (x.rowsq_in && (x.rowsq_parts <= 0 || (x.rowsq_parts & 3) || x.norm_dim <= 0 || !aligned16(x.rowsq_in)))
        return fail(LMI_EINVAL, "%s: rowsq_in needs rowsq_parts > 0 and a multiple of 4, norm_dim > 0 and 16-byte alignment", who);
    if (x.rowsq_in && epilogue != LMI_EPI_STORE && epilogue != LMI_EPI_SWIGLU && epilogue != LMI_EPI_QKV_ROPE)
        return fail(LMI_EINVAL, "%s: rowsq_in (row scale) is supported by the STORE, SWIGLU and q|k|v + RoPE epilogues", who);
    if (x.norm_out && (epilogue != LMI_EPI_RESIDUAL || !x.norm_gamma || !x.rowsq_out || (x.ld_norm & 7) || x.ld_norm < N || row_map ||
                       !aligned16(x.norm_out) || !aligned16(x.norm_gamma)))
        return fail(LMI_EINVAL, "%s: norm_out needs the RESIDUAL epilogue, norm_gamma, rowsq_out, ld_norm %% 8 == 0 and no row_map", who);
    if (M == 0) return LMI_OK;
    GemmArgs a;
    a.swiglu_f32 = 0;
    if (epilogue == LMI_EPI_SWIGLU_F32) { epilogue = LMI_EPI_SWIGLU; a.swiglu_f32 = 1; }
    a.A = A; a.W = W; a.out = out; a.bias = bias; a.addmat = addmat; a.add_rows = add_rows; a.row_map = row_map;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.add_period = add_period; a.ps_grid = ps_grid; a.group_m = g_gemm_group_m; a.order = g_gemm_order;
    a.norm_out = x.norm_out; a.norm_gamma = x.norm_gamma; a.rowsq_out = x.rowsq_out; a.ld_norm = x.ld_norm;
    a.rowsq_in = x.rowsq_in; a.rowsq_parts = x.rowsq_parts; a.norm_dim = x.norm_dim; a.norm_eps = x.norm_eps;
    a.rope_cos = x.rope_cos; a.rope_sin = x.rope_sin; a.k_cache = x.k_cache; a.v_cache = x.v_cache;
    a.ld_cache = x.ld_cache; a.cache_pos0 = x.cache_pos0; a.rope_q = x.rope_q; a.rope_k = x.rope_k;
    a.scale_e8m0 = 0x7f7f7f7f;
    a.out_scale = 1.0f;
    a.w_packed = w_packed ? 1 : 0;
    // extents for the buffer resources the LDS-DMA reads through (32-bit offsets)
    const long a_rows = (a_mode == LMI_A_PIXEL_SHUFFLE) ? (long)(M / ((ps_grid / 2) * (ps_grid / 2))) * ps_grid * ps_grid : (long)M;
    const long a_cols = (a_mode == LMI_A_PIXEL_SHUFFLE) ? K / 4 : K;
    const long a_bytes = a_rows > 0 ? ((a_rows - 1) * lda + a_cols) * 2 : 0, w_bytes = ((long)(N - 1) * ldw + K) * 2;
    if (a_bytes >= (1L << 32) || w_bytes >= (1L << 32))
        return fail(LMI_EINVAL, "%s: operand extent >= 4 GiB (A %ld, W %ld bytes)", who, a_bytes, w_bytes);
    a.a_bytes = (unsigned)a_bytes; a.w_bytes = (unsigned)w_bytes;
    a.A4 = nullptr; a.W4 = nullptr; a.a4_scale = nullptr; a.w4_scale = nullptr; a.lda4 = a.ldw4 = a.lds4 = a.K4 = 0;
    a.a4_bytes = a.w4_bytes = a.a4s_bytes = 0;
    a.out4 = nullptr; a.out4_scale = nullptr; a.ld_out4 = a.ld_out4s = 0;
    a.row_sel = nullptr; a.unit_sel = nullptr; a.sel_n = a.sel_total = 0;
    if (lo) {                                                       // low-bit correction phase (lmi_gemm_lo4 / lmi_rmsnorm_rope_lo4)
        const int k4 = lo->k4;          // K rounded up to 256, or wider: the images may carry their own (padded) k order — see lmi_attn_varlen_fwd_lo4
        if (!lo->a4 || !lo->a4_scale || !lo->w4 || !lo->w4_scale || k4 < K || (k4 & 255) || K < 128 || a_mode != LMI_A_PLAIN || (lo->lda4 & 15) ||
            (lo->ldw4 & 15) || lo->lda4 < k4 / 2 || lo->ldw4 < k4 / 2 || (lo->lds4 & 3) || lo->lds4 < k4 / 32 || !aligned16(lo->a4) || !aligned16(lo->w4) ||
            ((uintptr_t)lo->a4_scale & 3))
            return fail(LMI_EINVAL, "%s: lo4 needs all four images, k4 %% 256 == 0 and >= K (%d), K >= 128, a plain A, 16-byte aligned images with "
                        "lda4 / ldw4 %% 16 == 0 and >= k4 / 2, lds4 %% 4 == 0 and >= k4 / 32", who, k4);
        if ((lo->out4 != nullptr) != (lo->out4_scale != nullptr) ||
            (lo->out4 && (((uintptr_t)lo->out4 & 3) || (lo->ld_out4 & 3) || lo->ld_out4s <= 0 ||
                          !(epilogue == LMI_EPI_SWIGLU || epilogue == LMI_EPI_STORE || (epilogue == LMI_EPI_RESIDUAL && x.norm_out)) || a.swiglu_f32)))
            return fail(LMI_EINVAL, "%s: lo4 output needs out4 and out4_scale, ld_out4 %% 4 == 0, and the STORE / SWIGLU epilogue or the RESIDUAL producer mode", who);
        if (lo->out4) {                                             // rows of the output image must not overlap (advisor r05)
            const int out_cols = epilogue == LMI_EPI_SWIGLU ? N / 2 : N;
            if (lo->ld_out4 < out_cols / 2 || lo->ld_out4s < out_cols / 32)
                return fail(LMI_EINVAL, "%s: lo4 output rows overlap: ld_out4 (%d) must be >= %d bytes and ld_out4s (%d) >= %d for %d output columns", who,
                            lo->ld_out4, out_cols / 2, lo->ld_out4s, out_cols / 32, out_cols);
        }
        const long a4b = (long)(M - 1) * lo->lda4 + k4 / 2, w4b = (long)(N - 1) * lo->ldw4 + k4 / 2, sb = (long)(M - 1) * lo->lds4 + k4 / 32;
        if (a4b >= (1L << 32) || w4b >= (1L << 32)) return fail(LMI_EINVAL, "%s: lo4 image extent >= 4 GiB", who);
        a.A4 = lo->a4; a.W4 = lo->w4; a.a4_scale = (const uint8_t*)lo->a4_scale; a.w4_scale = (const uint8_t*)lo->w4_scale;
        a.lda4 = lo->lda4; a.ldw4 = lo->ldw4; a.lds4 = lo->lds4; a.K4 = k4;
        a.a4_bytes = (unsigned)a4b; a.w4_bytes = (unsigned)w4b; a.a4s_bytes = (unsigned)sb;
        a.out4 = lo->out4; a.out4_scale = (uint8_t*)lo->out4_scale; a.ld_out4 = lo->ld_out4; a.ld_out4s = lo->ld_out4s;
        if ((lo->row_sel != nullptr) != (lo->unit_sel != nullptr))
            return fail(LMI_EINVAL, "%s: lo4 row selection needs row_sel [M] and unit_sel [ceil(M / 64)] together (or neither: every row)", who);
        if (lo->row_sel && row_map) return fail(LMI_EINVAL, "%s: lo4 row selection does not combine with row_map", who);
        a.row_sel = (const uint8_t*)lo->row_sel; a.unit_sel = (const uint8_t*)lo->unit_sel;
        if (lo->n_sel_ranges < 0 || (lo->n_sel_ranges > 0 && (!lo->sel_ranges || !lo->row_sel)))
            return fail(LMI_EINVAL, "%s: lo4 sel_ranges needs n_sel_ranges >= 0, a host array of [begin, end) row pairs and row_sel / unit_sel", who);
        struct SelScope {                                          // visible to the launcher of THIS call only
            SelScope(const int* r, int n) { t_sel_ranges = r; t_sel_n = n; }
            ~SelScope() { t_sel_ranges = nullptr; t_sel_n = 0; }
        } scope(lo->sel_ranges, lo->n_sel_ranges);
        LMI_DISPATCH_T(dtype, dispatch_gemm_lo4<f16_t>(a, epilogue, act, stream), dispatch_gemm_lo4<bf16_t>(a, epilogue, act, stream));
    }
    // (Measured and dropped, profiles/r03_ab_tail_split_and_prologue.txt: computing the last 256-column tile of SigLIP q|k|v / fc1 with a
    // second, finer launch so that the tile count drops from 6.07 / 7.37 to 5.64 / 6.94 rounds of 256 workgroups — 0.13 % SLOWER over the
    // C3 step.  An almost-empty last round is cheap on this part: the idle CUs draw no power and the rest clock up.)
    LMI_DISPATCH_T(dtype, dispatch_gemm<f16_t>(a, epilogue, act, a_mode, stream),
                   dispatch_gemm<bf16_t>(a, epilogue, act, a_mode, stream));
}

int lmi_add_rmsnorm(float* x, const void* delta, int delta_dtype, const float* w, void* out, int M, int D, int ldx, int ldd, int ldo,
                    float eps, int dtype, void* stream) {
    if (!x || !delta || (out && !w) || M < 0 || D <= 0 || (D & 7) || D > 4096 || (ldx & 3) || (ldd & 7) || (out && (ldo & 7)) ||
        !aligned16(x) || !aligned16(delta) || (out && !aligned16(out)) || (w && !aligned16(w)))
        return fail(LMI_EINVAL, "lmi_add_rmsnorm: bad argument (M=%d D=%d; D %% 8 == 0, D <= 4096)", M, D);
    if (dtype != LMI_F16 && dtype != LMI_BF16) return fail(LMI_EINVAL, "lmi_add_rmsnorm: dtype must be LMI_F16 or LMI_BF16");
    if (delta_dtype != LMI_F32 && delta_dtype != dtype) return fail(LMI_EINVAL, "lmi_add_rmsnorm: delta_dtype must be LMI_F32 or dtype");
    if (M == 0) return LMI_OK;
    if (dtype == LMI_F16)
        return delta_dtype == LMI_F32 ? add_rmsnorm_impl<f16_t, float>(x, delta, w, out, M, D, ldx, ldd, ldo, eps, stream)
                                      : add_rmsnorm_impl<f16_t, f16_t>(x, delta, w, out, M, D, ldx, ldd, ldo, eps, stream);
    return delta_dtype == LMI_F32 ? add_rmsnorm_impl<bf16_t, float>(x, delta, w, out, M, D, ldx, ldd, ldo, eps, stream)
                                  : add_rmsnorm_impl<bf16_t, bf16_t>(x, delta, w, out, M, D, ldx, ldd, ldo, eps, stream);
}

int lmi_add_rmsnorm_lo4(float* x, const void* delta, int delta_dtype, const float* w, void* out, void* out4, void* scales, int M, int D, int K4,
                        int ldx, int ldd, int ldo, int ld4, int lds, float eps, int dtype, void* stream) {
    if (!x || !delta || !w || !out || !out4 || !scales || M < 0 || D <= 0 || (D & 31) || D > 4096 || K4 != (D + 255) / 256 * 256 || (ldx & 3) || (ldd & 7) ||
        (ldo & 7) || (ld4 & 3) || ld4 < K4 / 2 || lds < K4 / 32 || !aligned16(x) || !aligned16(delta) || !aligned16(out) || !aligned16(w) || ((uintptr_t)out4 & 3))
        return fail(LMI_EINVAL, "lmi_add_rmsnorm_lo4: bad argument (M=%d D=%d K4=%d; D %% 32 == 0, D <= 4096, K4 = D rounded up to 256)", M, D, K4);
    if (dtype != LMI_F16 && dtype != LMI_BF16) return fail(LMI_EINVAL, "lmi_add_rmsnorm_lo4: dtype must be LMI_F16 or LMI_BF16");
    if (delta_dtype != LMI_F32 && delta_dtype != dtype) return fail(LMI_EINVAL, "lmi_add_rmsnorm_lo4: delta_dtype must be LMI_F32 or dtype");
    if (M == 0) return LMI_OK;
    NormLo4 lo;
    lo.out4 = (uint8_t*)out4; lo.scales = (uint8_t*)scales; lo.ld4 = ld4; lo.lds = lds; lo.K4 = K4; lo.row_sel = nullptr;
    const int grid = (M + 3) / 4;
#define LMI_ADDNORM4(T_, DT_)                                                                                                                     \
    do {                                                                                                                                          \
        if (D <= 1536) LMI_LAUNCH((add_rmsnorm_kernel<T_, DT_, 3, true>), dim3(grid), dim3(256), 0, stream, x, (const DT_*)delta, w, (T_*)out, M, D, ldx, ldd, ldo, eps, lo); \
        else LMI_LAUNCH((add_rmsnorm_kernel<T_, DT_, 8, true>), dim3(grid), dim3(256), 0, stream, x, (const DT_*)delta, w, (T_*)out, M, D, ldx, ldd, ldo, eps, lo);          \
    } while (0)
    if (dtype == LMI_F16) { if (delta_dtype == LMI_F32) LMI_ADDNORM4(f16_t, float); else LMI_ADDNORM4(f16_t, f16_t); }
    else { if (delta_dtype == LMI_F32) LMI_ADDNORM4(bf16_t, float); else LMI_ADDNORM4(bf16_t, bf16_t); }
#undef LMI_ADDNORM4
    return check_launch("lmi_add_rmsnorm_lo4");
}

int lmi_gemm(const void* A, const void* W, void* out, const float* bias, const float* addmat, const int* add_rows,
             const int* row_map,
             int M, int N, int K, int lda, int ldw, int ldo, int add_period, int epilogue, int act, int a_mode,
             int ps_grid, int dtype, void* stream) {
    if (epilogue == LMI_EPI_QKV_ROPE) return fail(LMI_EINVAL, "lmi_gemm: the q|k|v + RoPE epilogue is reached through lmi_rmsnorm_rope");
    return gemm_entry("lmi_gemm", A, W, out, bias, addmat, add_rows, row_map, M, N, K, lda, ldw, ldo, add_period, epilogue, act, a_mode, ps_grid,
                      dtype, stream, GemmExtras());
}

int lmi_gemm_bias_act(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int act,
                      int residual, int ps_grid, int dtype, void* stream) {
    if (act < LMI_ACT_NONE || act > LMI_ACT_SWIGLU || (act == LMI_ACT_SWIGLU && residual))
        return fail(LMI_EINVAL, "lmi_gemm_bias_act: act must be LMI_ACT_NONE / GELU_TANH / GELU_ERF / SWIGLU (SwiGLU has no residual form)");
    const int epilogue = residual ? LMI_EPI_RESIDUAL : (act == LMI_ACT_SWIGLU ? LMI_EPI_SWIGLU : LMI_EPI_STORE);
    return gemm_entry("lmi_gemm_bias_act", A, W, out, bias, nullptr, nullptr, nullptr, M, N, K, lda, ldw, ldo, 0, epilogue,
                      act == LMI_ACT_SWIGLU ? LMI_ACT_NONE : act, ps_grid > 0 ? LMI_A_PIXEL_SHUFFLE : LMI_A_PLAIN, ps_grid, dtype, stream, GemmExtras());
}

int lmi_gemm_ex(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, const float* norm_gamma, float* rowsq_out,
                int ld_norm, int dtype, void* stream) {
    if (epilogue == LMI_EPI_QKV_ROPE) return fail(LMI_EINVAL, "lmi_gemm_ex: the q|k|v + RoPE epilogue is reached through lmi_rmsnorm_rope");
    GemmExtras x;
    x.rowsq_in = rowsq_in; x.rowsq_parts = rowsq_parts; x.norm_dim = norm_dim; x.norm_eps = norm_eps;
    x.norm_out = norm_out; x.norm_gamma = norm_gamma; x.rowsq_out = rowsq_out; x.ld_norm = ld_norm;
    return gemm_entry("lmi_gemm_ex", A, W, out, bias, nullptr, nullptr, nullptr, M, N, K, lda, ldw, ldo, 0, epilogue, act, LMI_A_PLAIN, 0, dtype,
                      stream, x);
}

int lmi_rmsnorm_rope(const void* A, const void* Wqkv, void* qkv, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_table,
                     const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads,
                     int head_dim, int K, int lda, int ldw, int ldo, int dtype, void* stream) {
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_rmsnorm_rope: head_dim %d (only 128: a wave's 64 columns hold half a head)", head_dim);
    if (!cos_table || !sin_table || n_q_heads <= 0 || n_kv_heads <= 0 || ((k_cache != nullptr) != (v_cache != nullptr)) ||
        (k_cache && ((ld_cache & 7) || !aligned16(k_cache) || !aligned16(v_cache))) || (ldo & 7) || !aligned16(cos_table) || !aligned16(sin_table))
        return fail(LMI_EINVAL, "lmi_rmsnorm_rope: bad argument");
    GemmExtras x;
    x.rowsq_in = rowsq_in; x.rowsq_parts = rowsq_parts; x.norm_dim = K; x.norm_eps = norm_eps;
    x.rope_cos = cos_table; x.rope_sin = sin_table; x.k_cache = k_cache; x.v_cache = v_cache; x.ld_cache = ld_cache; x.cache_pos0 = cache_pos0;
    x.rope_q = n_q_heads * head_dim; x.rope_k = n_kv_heads * head_dim;
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    return gemm_entry("lmi_rmsnorm_rope", A, Wqkv, qkv, nullptr, nullptr, nullptr, nullptr, M, N, K, lda, ldw, ldo, 0, LMI_EPI_QKV_ROPE, LMI_ACT_NONE,
                      LMI_A_PLAIN, 0, dtype, stream, x);
}

int lmi_gemm_lo4(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                 const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, const float* norm_gamma, float* rowsq_out,
                 int ld_norm, const lmi_lo4* lo, int dtype, void* stream) {
    if (!lo) return fail(LMI_EINVAL, "lmi_gemm_lo4: null lo4 descriptor (use lmi_gemm_ex)");
    if (epilogue == LMI_EPI_QKV_ROPE) return fail(LMI_EINVAL, "lmi_gemm_lo4: the q|k|v + RoPE epilogue is reached through lmi_rmsnorm_rope_lo4");
    GemmExtras x;
    x.rowsq_in = rowsq_in; x.rowsq_parts = rowsq_parts; x.norm_dim = norm_dim; x.norm_eps = norm_eps;
    x.norm_out = norm_out; x.norm_gamma = norm_gamma; x.rowsq_out = rowsq_out; x.ld_norm = ld_norm;
    return gemm_entry("lmi_gemm_lo4", A, W, out, bias, nullptr, nullptr, nullptr, M, N, K, lda, ldw, ldo, 0, epilogue, act, LMI_A_PLAIN, 0, dtype,
                      stream, x, lo);
}

int lmi_rmsnorm_rope_lo4(const void* A, const void* Wqkv, void* qkv, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_table,
                         const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads,
                         int head_dim, int K, int lda, int ldw, int ldo, const lmi_lo4* lo, int dtype, void* stream) {
    if (!lo) return fail(LMI_EINVAL, "lmi_rmsnorm_rope_lo4: null lo4 descriptor (use lmi_rmsnorm_rope)");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_rmsnorm_rope_lo4: head_dim %d (only 128: a wave's 64 columns hold half a head)", head_dim);
    if (!cos_table || !sin_table || n_q_heads <= 0 || n_kv_heads <= 0 || ((k_cache != nullptr) != (v_cache != nullptr)) ||
        (k_cache && ((ld_cache & 7) || !aligned16(k_cache) || !aligned16(v_cache))) || (ldo & 7) || !aligned16(cos_table) || !aligned16(sin_table))
        return fail(LMI_EINVAL, "lmi_rmsnorm_rope_lo4: bad argument");
    GemmExtras x;
    x.rowsq_in = rowsq_in; x.rowsq_parts = rowsq_parts; x.norm_dim = K; x.norm_eps = norm_eps;
    x.rope_cos = cos_table; x.rope_sin = sin_table; x.k_cache = k_cache; x.v_cache = v_cache; x.ld_cache = ld_cache; x.cache_pos0 = cache_pos0;
    x.rope_q = n_q_heads * head_dim; x.rope_k = n_kv_heads * head_dim;
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    return gemm_entry("lmi_rmsnorm_rope_lo4", A, Wqkv, qkv, nullptr, nullptr, nullptr, nullptr, M, N, K, lda, ldw, ldo, 0, LMI_EPI_QKV_ROPE, LMI_ACT_NONE,
                      LMI_A_PLAIN, 0, dtype, stream, x, lo);
}

int lmi_split_lo4(const float* x, void* hi, void* lo4, void* scales, int M, int K, int K4, int ldx, int ldh, int ld4, int lds, int dtype, void* stream) {
    if (!x || !hi || !lo4 || !scales || M < 0 || K <= 0 || (K & 31) || K4 != (K + 255) / 256 * 256 || (ldx & 3) || (ldh & 7) || ldh < K || (ld4 & 3) ||
        ld4 < K4 / 2 || lds < K4 / 32 || !aligned16(x) || !aligned16(hi) || ((uintptr_t)lo4 & 3))
        return fail(LMI_EINVAL, "lmi_split_lo4: bad argument (M=%d K=%d K4=%d; K %% 32 == 0, K4 = K rounded up to 256, ld4 >= K4 / 2, lds >= K4 / 32)", M, K, K4);
    if (M == 0) return LMI_OK;
    const int grid = grid_for((long)M * (K4 >> 3), 256);
    if (dtype == LMI_F16) LMI_LAUNCH((split_lo4_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, x, (f16_t*)hi, (uint8_t*)lo4, (uint8_t*)scales, M, K, K4, ldx, ldh, ld4, lds);
    else if (dtype == LMI_BF16) LMI_LAUNCH((split_lo4_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, x, (bf16_t*)hi, (uint8_t*)lo4, (uint8_t*)scales, M, K, K4, ldx, ldh, ld4, lds);
    else return fail(LMI_EINVAL, "lmi_split_lo4: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_split_lo4");
}

int lmi_norm_lo4_rows(const float* x, const float* w, const float* b, void* out, void* out4, void* scales, int M, int D, int K4, int ldx, int ldo,
                      int ld4, int lds, float eps, const void* row_sel, int dtype, void* stream) {
    if (!x || !w || !out || !out4 || !scales || M < 0 || D <= 0 || (D & 31) || D > 4096 || K4 != (D + 255) / 256 * 256 || (ldx & 3) || (ldo & 7) ||
        (ld4 & 3) || ld4 < K4 / 2 || lds < K4 / 32 || !aligned16(x) || !aligned16(w) || !aligned16(out) || ((uintptr_t)out4 & 3) || (b && !aligned16(b)))
        return fail(LMI_EINVAL, "lmi_norm_lo4[_rows]: bad argument (M=%d D=%d K4=%d; D %% 32 == 0, D <= 4096, K4 = D rounded up to 256)", M, D, K4);
    if (M == 0) return LMI_OK;
    NormLo4 lo;
    lo.out4 = (uint8_t*)out4; lo.scales = (uint8_t*)scales; lo.ld4 = ld4; lo.lds = lds; lo.K4 = K4; lo.row_sel = (const uint8_t*)row_sel;
    if (dtype == LMI_F16) return b ? norm_lo4_impl<f16_t, false>(x, w, b, out, lo, M, D, ldx, ldo, eps, stream) : norm_lo4_impl<f16_t, true>(x, w, b, out, lo, M, D, ldx, ldo, eps, stream);
    if (dtype == LMI_BF16) return b ? norm_lo4_impl<bf16_t, false>(x, w, b, out, lo, M, D, ldx, ldo, eps, stream) : norm_lo4_impl<bf16_t, true>(x, w, b, out, lo, M, D, ldx, ldo, eps, stream);
    return fail(LMI_EINVAL, "lmi_norm_lo4[_rows]: dtype must be LMI_F16 or LMI_BF16");
}

int lmi_norm_lo4(const float* x, const float* w, const float* b, void* out, void* out4, void* scales, int M, int D, int K4, int ldx, int ldo,
                 int ld4, int lds, float eps, int dtype, void* stream) {
    return lmi_norm_lo4_rows(x, w, b, out, out4, scales, M, D, K4, ldx, ldo, ld4, lds, eps, nullptr, dtype, stream);
}

int lmi_quantize_w4(const void* W, void* w4, void* scales, int N, int K, int K4, int ldw, int ld4, int dtype, void* stream) {
    if (!W || !w4 || !scales || N <= 0 || K <= 0 || (K & 7) || K4 < K || (K4 & 255) || (ldw & 7) || ldw < K || (ld4 & 15) || ld4 < K4 / 2 ||
        !aligned16(W) || !aligned16(w4))
        return fail(LMI_EINVAL, "lmi_quantize_w4: bad argument (N=%d K=%d K4=%d; K %% 8 == 0, K4 = K rounded up to 256, ld4 %% 16 == 0 and >= K4 / 2)", N, K, K4);
    const int grid = (N + 3) / 4;
    if (dtype == LMI_F16) LMI_LAUNCH((quantize_w4_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, (const f16_t*)W, (uint8_t*)w4, (uint8_t*)scales, N, K, K4, ldw, ld4);
    else if (dtype == LMI_BF16) LMI_LAUNCH((quantize_w4_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)W, (uint8_t*)w4, (uint8_t*)scales, N, K, K4, ldw, ld4);
    else return fail(LMI_EINVAL, "lmi_quantize_w4: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_quantize_w4");
}

int lmi_rope_qkv_fp8(const void* A8, const void* Wqkv8, void* qkv, int scale_exp, const float* cos_table, const float* sin_table, void* k_cache,
                     void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int lda, int ldw, int ldo,
                     int out_dtype, void* stream) {
    if (!A8 || !Wqkv8 || !qkv || !cos_table || !sin_table) return fail(LMI_EINVAL, "lmi_rope_qkv_fp8: null pointer");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_rope_qkv_fp8: head_dim %d (only 128: a wave's 64 columns hold half a head)", head_dim);
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    if (M < 0 || n_q_heads <= 0 || n_kv_heads <= 0 || K <= 0 || (K % 128) || (lda & 15) || (ldw & 15) || (ldo & 7) || !aligned16(A8) || !aligned16(Wqkv8) ||
        !aligned16(qkv) || ((k_cache != nullptr) != (v_cache != nullptr)) || (k_cache && ((ld_cache & 7) || !aligned16(k_cache) || !aligned16(v_cache))) ||
        !aligned16(cos_table) || !aligned16(sin_table) || scale_exp < -120 || scale_exp > 120)
        return fail(LMI_EINVAL, "lmi_rope_qkv_fp8: bad argument (K %% 128 == 0, lda / ldw %% 16 == 0, ldo %% 8 == 0, 16-byte aligned pointers)");
    if (M == 0) return LMI_OK;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A8; a.W = Wqkv8; a.out = qkv;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.group_m = g_gemm_group_m; a.order = g_gemm_order;
    const int e = 127 + scale_exp;
    a.scale_e8m0 = e | (e << 8) | (e << 16) | (e << 24);
    a.out_scale = 1.0f;
    a.rope_cos = cos_table; a.rope_sin = sin_table; a.k_cache = k_cache; a.v_cache = v_cache; a.ld_cache = ld_cache; a.cache_pos0 = cache_pos0;
    a.rope_q = n_q_heads * head_dim; a.rope_k = n_kv_heads * head_dim;
    const long a_bytes = ((long)(M - 1) * lda + K), w_bytes = ((long)(N - 1) * ldw + K);
    if (a_bytes >= (1L << 32) || w_bytes >= (1L << 32)) return fail(LMI_EINVAL, "lmi_rope_qkv_fp8: operand extent >= 4 GiB");
    a.a_bytes = (unsigned)a_bytes; a.w_bytes = (unsigned)w_bytes;
    LMI_DISPATCH_T(out_dtype, (launch_gemm_fp8<f16_t, EPI_QKV_ROPE_T, ACT_NONE>(a, stream)), (launch_gemm_fp8<bf16_t, EPI_QKV_ROPE_T, ACT_NONE>(a, stream)));
}

int lmi_norm_fp8(const float* x, const float* w, const float* b, void* out, int M, int D, int ldx, int ldo, float eps, float out_scale,
                 void* stream) {
    if (ldo & 15) return fail(LMI_EINVAL, "lmi_norm_fp8: ldo must be a multiple of 16 (fp8 GEMM operand rows)");
    if (b) return norm_impl<fp8_t, false>(x, w, b, out, M, D, ldx, ldo, eps, stream, "lmi_norm_fp8", out_scale);
    return norm_impl<fp8_t, true>(x, w, nullptr, out, M, D, ldx, ldo, eps, stream, "lmi_norm_fp8", out_scale);
}

int lmi_quantize_fp8(const void* x, int x_dtype, void* out, int M, int D, int ldx, int ldo, float scale, void* stream) {
    if (!x || !out || M < 0 || D <= 0 || (D & 7) || (ldx & 7) || (ldo & 15) || !aligned16(x) || !aligned16(out))
        return fail(LMI_EINVAL, "lmi_quantize_fp8: bad argument (M=%d D=%d ldx=%d ldo=%d; D %% 8 == 0, ldx %% 8 == 0, ldo %% 16 == 0)", M, D, ldx, ldo);
    if (M == 0) return LMI_OK;
    const int grid = grid_for((long)M * (D >> 3), 256);
    uint8_t* o = (uint8_t*)out;
    if (x_dtype == LMI_F32) LMI_LAUNCH((quantize_fp8_kernel<float>), dim3(grid), dim3(256), 0, stream, (const float*)x, o, M, D, ldx, ldo, scale);
    else if (x_dtype == LMI_F16) LMI_LAUNCH((quantize_fp8_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, (const f16_t*)x, o, M, D, ldx, ldo, scale);
    else if (x_dtype == LMI_BF16) LMI_LAUNCH((quantize_fp8_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, o, M, D, ldx, ldo, scale);
    else return fail(LMI_EINVAL, "lmi_quantize_fp8: x_dtype must be LMI_F32, LMI_F16 or LMI_BF16");
    return check_launch("lmi_quantize_fp8");
}

int lmi_gemm_fp8(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                 int scale_exp, int out_dtype, float out_scale, void* stream) {
    if (!A || !W || !out) return fail(LMI_EINVAL, "lmi_gemm_fp8: null pointer");
    if (M < 0 || N <= 0 || K <= 0 || (N % 128) || (K % 128))
        return fail(LMI_EINVAL, "lmi_gemm_fp8: need N %% 128 == 0 and K %% 128 == 0 (M=%d N=%d K=%d)", M, N, K);
    if ((lda & 15) || (ldw & 15) || (ldo & 3) || !aligned16(A) || !aligned16(W) || !aligned16(out) || (bias && !aligned16(bias)))
        return fail(LMI_EINVAL, "lmi_gemm_fp8: pointers must be 16-byte aligned, lda/ldw multiples of 16, ldo of 4");
    if (scale_exp < -120 || scale_exp > 120) return fail(LMI_EINVAL, "lmi_gemm_fp8: scale_exp out of range");
    if (M == 0) return LMI_OK;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.out = out; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.group_m = g_gemm_group_m; a.order = g_gemm_order;
    const int e = 127 + scale_exp;
    a.scale_e8m0 = e | (e << 8) | (e << 16) | (e << 24);
    a.out_scale = out_scale;
    const long a_bytes = ((long)(M - 1) * lda + K), w_bytes = ((long)(N - 1) * ldw + K);
    if (a_bytes >= (1L << 32) || w_bytes >= (1L << 32)) return fail(LMI_EINVAL, "lmi_gemm_fp8: operand extent >= 4 GiB");
    a.a_bytes = (unsigned)a_bytes; a.w_bytes = (unsigned)w_bytes;
    if (out_dtype == LMI_FP8) {                                    // fp8 results for the next fp8 GEMM: GELU (ViT fc1) and SwiGLU (gate/up) only
        if ((ldo & 7) || out_scale <= 0.f) return fail(LMI_EINVAL, "lmi_gemm_fp8: fp8 output needs ldo %% 8 == 0 and out_scale > 0");
        if (epilogue == LMI_EPI_STORE && act == LMI_ACT_GELU_TANH) return launch_gemm_fp8<fp8_t, EPI_STORE_T, ACT_GELU_TANH>(a, stream);
        if (epilogue == LMI_EPI_STORE && act == LMI_ACT_NONE) return launch_gemm_fp8<fp8_t, EPI_STORE_T, ACT_NONE>(a, stream);
        if (epilogue == LMI_EPI_SWIGLU && act == LMI_ACT_NONE) return launch_gemm_fp8<fp8_t, EPI_SWIGLU_T, ACT_NONE>(a, stream);
        return fail(LMI_EINVAL, "lmi_gemm_fp8: fp8 output is supported for STORE (+GELU-tanh) and SWIGLU");
    }
    LMI_DISPATCH_T(out_dtype, dispatch_gemm_fp8<f16_t>(a, epilogue, act, stream), dispatch_gemm_fp8<bf16_t>(a, epilogue, act, stream));
}

static int attn_varlen_entry(const char* who, const void* q, const void* k, const void* v, void* out, float* out_f32, int ldo32, void* out_fp8, int ldo8, float out_fp8_scale,
                             const int* cu_seqlens_q, const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                             int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, int use_tr, int dtype, void* stream,
                             void* out4 = nullptr, void* out4_scale = nullptr, int ld_out4 = 0, int ld_out4s = 0, const uint8_t* row_sel = nullptr) {
    if (!q || !k || !v || (!out && !out_fp8 && !out_f32) || !cu_seqlens_q || !cu_seqlens_k) return fail(LMI_EINVAL, "%s: null pointer", who);
    if (out4) {
        const int ndb = (head_dim + 31) / 32;
        if (!out || !out4_scale || !use_tr || !g_attn_dma.load() || ((uintptr_t)out4 & 7) || (ld_out4 & 7) || ld_out4 < n_heads * ndb * 16 ||
            ld_out4s < n_heads * ndb)
            return fail(LMI_EINVAL, "%s: the residual image needs the 16-bit output, the LDS-DMA kernel, an 8-byte aligned image with ld_out4 %% 8 == 0 and "
                        ">= n_heads * ceil(head_dim / 32) * 16 bytes, ld_out4s >= n_heads * ceil(head_dim / 32)", who);
    }
    if (n_seq < 0 || max_seqlen_q < 0 || n_heads <= 0 || n_kv_heads <= 0 || (n_heads % n_kv_heads))
        return fail(LMI_EINVAL, "%s: bad head counts (%d, %d)", who, n_heads, n_kv_heads);
    if (head_dim != 128 && head_dim != 96 && head_dim != 72)
        return fail(LMI_EINVAL, "%s: head_dim %d not in {72, 96, 128}", who, head_dim);
    if (window < 0) return fail(LMI_EINVAL, "%s: window must be >= 0", who);
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || !aligned16(q) || !aligned16(k) || !aligned16(v) || (out && !aligned16(out)))
        return fail(LMI_EINVAL, "%s: alignment", who);
    if (out_fp8 && (!use_tr || !g_attn_dma.load() || (ldo8 & 7) || ((uintptr_t)out_fp8 & 7) || !(out_fp8_scale > 0.f)))
        return fail(LMI_EINVAL, "%s: the fp8 output needs the LDS-DMA kernel (use_tr), ldo8 %% 8 == 0, an 8-byte aligned pointer and a positive scale", who);
    if (out_f32 && (!use_tr || !g_attn_dma.load() || (ldo32 & 3) || !aligned16(out_f32)))
        return fail(LMI_EINVAL, "%s: the fp32 output needs the LDS-DMA kernel (use_tr), ldo32 %% 4 == 0 and a 16-byte aligned pointer", who);
    if (n_seq == 0 || max_seqlen_q == 0) return LMI_OK;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.cu_q = cu_seqlens_q; a.cu_k = cu_seqlens_k; a.k_len = nullptr;
    a.out_fp8 = out_fp8; a.ldo8 = ldo8; a.out_fp8_scale = out_fp8_scale; a.out_f32 = out_f32; a.ldo32 = ldo32;
    a.out4 = (uint8_t*)out4; a.out4_scale = (uint8_t*)out4_scale; a.ld_out4 = ld_out4; a.ld_out4s = ld_out4s; a.row_sel = row_sel;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.scale = scale; a.window = window; a.n_qblocks = 0;
    a.n_splits = 1; a.split_tiles = 0; a.part_rows = 0; a.part_o = nullptr; a.part_ml = nullptr; a.gqa_pack = 0;
    a.check_k_extent = 1;
    if (cu_seqlens_k == cu_seqlens_q) {                              // self-attention: the longest key sequence is max_seqlen_q
        if (((long)max_seqlen_q * ldk + head_dim) * 2 >= (1L << 32) || ((long)max_seqlen_q * ldv + head_dim) * 2 >= (1L << 32))
            return fail(LMI_EINVAL, "%s: one sequence's K / V rows span >= 4 GiB (max_seqlen %d, ldk %d, ldv %d)", who, max_seqlen_q, ldk, ldv);
        a.check_k_extent = 0;
    }
    if (head_dim == 128)
        LMI_DISPATCH_T(dtype, (dispatch_attn<f16_t, 128>(a, n_seq, max_seqlen_q, causal, use_tr, stream)),
                       (dispatch_attn<bf16_t, 128>(a, n_seq, max_seqlen_q, causal, use_tr, stream)));
    if (head_dim == 96)
        LMI_DISPATCH_T(dtype, (dispatch_attn<f16_t, 96>(a, n_seq, max_seqlen_q, causal, use_tr, stream)),
                       (dispatch_attn<bf16_t, 96>(a, n_seq, max_seqlen_q, causal, use_tr, stream)));
    LMI_DISPATCH_T(dtype, (dispatch_attn<f16_t, 72>(a, n_seq, max_seqlen_q, causal, use_tr, stream)),
                   (dispatch_attn<bf16_t, 72>(a, n_seq, max_seqlen_q, causal, use_tr, stream)));
}

int lmi_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q,
                        const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                        int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, int use_tr, int dtype, void* stream) {
    if (!out) return fail(LMI_EINVAL, "lmi_attn_varlen_fwd: null pointer");
    return attn_varlen_entry("lmi_attn_varlen_fwd", q, k, v, out, nullptr, 0, nullptr, 0, 0.f, cu_seqlens_q, cu_seqlens_k, n_seq, max_seqlen_q, n_heads, n_kv_heads,
                             head_dim, ldq, ldk, ldv, ldo, scale, causal, window, use_tr, dtype, stream);
}

int lmi_attn_varlen_fwd_fp8(const void* q, const void* k, const void* v, void* out_fp8, int ldo8, float out_scale, const int* cu_seqlens_q,
                            const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, float scale, int causal, int window, int dtype, void* stream) {
    if (!out_fp8) return fail(LMI_EINVAL, "lmi_attn_varlen_fwd_fp8: null pointer");
    return attn_varlen_entry("lmi_attn_varlen_fwd_fp8", q, k, v, nullptr, nullptr, 0, out_fp8, ldo8, out_scale, cu_seqlens_q, cu_seqlens_k, n_seq, max_seqlen_q, n_heads,
                             n_kv_heads, head_dim, ldq, ldk, ldv, 0, scale, causal, window, 1, dtype, stream);
}

// ---- fp8 attention arithmetic (attention_fp8.h): operand preparation + forward ----------------------------------------------------
int lmi_attn_prep_fp8(const void* qkv, int ld, const int* cu_seqlens, const int* tile_base, int n_seq, int n_tiles, int n_q_heads, int n_kv_heads,
                      int head_dim, float q_scale, float k_scale, float v_scale, void* q8, int ldq8, void* k_img, void* v_img, int dtype,
                      void* stream) {
    if (!qkv || !cu_seqlens || !tile_base || !q8 || !k_img || !v_img) return fail(LMI_EINVAL, "lmi_attn_prep_fp8: null pointer");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_attn_prep_fp8: head_dim %d (only 128)", head_dim);
    if (n_seq < 0 || n_tiles < 0 || n_q_heads <= 0 || n_kv_heads <= 0 || (n_q_heads % n_kv_heads) || (ld & 7) || (ldq8 & 15) ||
        ld < (n_q_heads + 2 * n_kv_heads) * 128 || ldq8 < n_q_heads * 128 || !aligned16(qkv) || !aligned16(q8) || !aligned16(k_img) || !aligned16(v_img) ||
        !(q_scale > 0.f) || !(k_scale > 0.f) || !(v_scale > 0.f))
        return fail(LMI_EINVAL, "lmi_attn_prep_fp8: bad argument (16-byte aligned pointers, ld %% 8 == 0, ldq8 %% 16 == 0, positive scales)");
    if ((long)n_tiles * ATT8_IMG >= (1L << 32)) return fail(LMI_EINVAL, "lmi_attn_prep_fp8: %d key tiles per kv head exceed a 4 GiB image", n_tiles);
    if (n_seq == 0 || n_tiles == 0) return LMI_OK;
    AttnPrepArgs a;
    a.qkv = qkv; a.cu = cu_seqlens; a.tile_base = tile_base; a.q8 = (uint8_t*)q8; a.k_img = (uint8_t*)k_img; a.v_img = (uint8_t*)v_img;
    a.ld = ld; a.ldq8 = ldq8; a.n_seq = n_seq; a.n_tiles = n_tiles; a.n_heads = n_q_heads; a.n_kv_heads = n_kv_heads;
    a.q_scale = q_scale; a.k_scale = k_scale; a.v_scale = v_scale;
    const dim3 grid((unsigned)n_tiles, (unsigned)(2 * n_kv_heads + n_q_heads));
    if (dtype == LMI_F16) LMI_LAUNCH((attn_prep_fp8_kernel<f16_t>), grid, dim3(256), 0, stream, a);
    else if (dtype == LMI_BF16) LMI_LAUNCH((attn_prep_fp8_kernel<bf16_t>), grid, dim3(256), 0, stream, a);
    else return fail(LMI_EINVAL, "lmi_attn_prep_fp8: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_attn_prep_fp8");
}

int lmi_attn_fp8_fwd(const void* q8, int ldq8, const void* k_img, const void* v_img, void* out, int ldo, void* out_fp8, int ldo8, float out_fp8_scale,
                     const int* cu_seqlens, const int* tile_base, int n_seq, int n_tiles, int max_seqlen, int n_heads, int n_kv_heads, int head_dim,
                     float softmax_scale, float q_scale, float k_scale, float v_scale, int causal, int dtype, void* stream) {
    if (!q8 || !k_img || !v_img || (!out && !out_fp8) || !cu_seqlens || !tile_base) return fail(LMI_EINVAL, "lmi_attn_fp8_fwd: null pointer");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_attn_fp8_fwd: head_dim %d (only 128)", head_dim);
    if (n_seq < 0 || n_tiles < 0 || max_seqlen < 0 || n_heads <= 0 || n_kv_heads <= 0 || (n_heads % n_kv_heads) || (ldq8 & 15) || ldq8 < n_heads * 128 ||
        (out && ((ldo & 7) || ldo < n_heads * 128 || !aligned16(out))) || (out_fp8 && ((ldo8 & 7) || ldo8 < n_heads * 128 || ((uintptr_t)out_fp8 & 7))) ||
        !aligned16(q8) || !aligned16(k_img) || !aligned16(v_img) || !(q_scale > 0.f) || !(k_scale > 0.f) || !(v_scale > 0.f))
        return fail(LMI_EINVAL, "lmi_attn_fp8_fwd: bad argument");
    if ((long)n_tiles * ATT8_IMG >= (1L << 32)) return fail(LMI_EINVAL, "lmi_attn_fp8_fwd: %d key tiles per kv head exceed a 4 GiB image", n_tiles);
    if (n_seq == 0 || max_seqlen == 0 || n_tiles == 0) return LMI_OK;
    AttnFp8Args a;
    a.q8 = (const uint8_t*)q8; a.k_img = (const uint8_t*)k_img; a.v_img = (const uint8_t*)v_img; a.out = out_fp8 ? nullptr : out; a.out_fp8 = (uint8_t*)out_fp8;
    a.cu = cu_seqlens; a.tile_base = tile_base; a.ldq8 = ldq8; a.ldo = ldo; a.ldo8 = ldo8; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads;
    a.n_tiles = n_tiles; a.n_qblocks = (max_seqlen + ATT_BQ - 1) / ATT_BQ;
    a.c2 = softmax_scale * 1.4426950408889634f / (q_scale * k_scale);
    a.inv_v_scale = 1.0f / v_scale;
    a.out_fp8_scale = out_fp8_scale;
    LMI_DISPATCH_T(dtype, (attn_fp8_impl<f16_t>(a, n_seq, causal, stream)), (attn_fp8_impl<bf16_t>(a, n_seq, causal, stream)));
}

int lmi_attn_varlen_fwd_f32(const void* q, const void* k, const void* v, float* out_f32, int ldo32, const int* cu_seqlens_q,
                            const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, float scale, int causal, int window, int dtype, void* stream) {
    if (!out_f32) return fail(LMI_EINVAL, "lmi_attn_varlen_fwd_f32: null pointer");
    return attn_varlen_entry("lmi_attn_varlen_fwd_f32", q, k, v, nullptr, out_f32, ldo32, nullptr, 0, 0.f, cu_seqlens_q, cu_seqlens_k, n_seq, max_seqlen_q,
                             n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, 0, scale, causal, window, 1, dtype, stream);
}

int lmi_attn_varlen_fwd_lo4_rows(const void* q, const void* k, const void* v, void* out, void* out4, void* out4_scale, int ld_out4, int ld_out4s,
                                 const int* cu_seqlens_q, const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                                 int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, const void* row_sel, int dtype, void* stream) {
    if (!out || !out4 || !out4_scale) return fail(LMI_EINVAL, "lmi_attn_varlen_fwd_lo4: null pointer");
    return attn_varlen_entry("lmi_attn_varlen_fwd_lo4", q, k, v, out, nullptr, 0, nullptr, 0, 0.f, cu_seqlens_q, cu_seqlens_k, n_seq, max_seqlen_q, n_heads,
                             n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, causal, window, 1, dtype, stream, out4, out4_scale, ld_out4, ld_out4s,
                             (const uint8_t*)row_sel);
}

int lmi_attn_varlen_fwd_lo4(const void* q, const void* k, const void* v, void* out, void* out4, void* out4_scale, int ld_out4, int ld_out4s,
                            const int* cu_seqlens_q, const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, int dtype, void* stream) {
    return lmi_attn_varlen_fwd_lo4_rows(q, k, v, out, out4, out4_scale, ld_out4, ld_out4s, cu_seqlens_q, cu_seqlens_k, n_seq, max_seqlen_q, n_heads,
                                        n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, causal, window, nullptr, dtype, stream);
}

int lmi_split_hi_lo(const float* x, void* out, int M, int K, int ldx, int ldo, int dtype, void* stream) {
    if (!x || !out || M < 0 || K <= 0 || (K & 7) || (ldx & 3) || (ldo & 7) || ldo < 2 * K || !aligned16(x) || !aligned16(out))
        return fail(LMI_EINVAL, "lmi_split_hi_lo: bad argument (M=%d K=%d ldx=%d ldo=%d; K %% 8 == 0, ldo >= 2K)", M, K, ldx, ldo);
    if (M == 0) return LMI_OK;
    const int grid = grid_for((long)M * (K >> 3), 256);
    if (dtype == LMI_F16) LMI_LAUNCH((split_hi_lo_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, x, (f16_t*)out, M, K, ldx, ldo, (long)K);
    else if (dtype == LMI_BF16) LMI_LAUNCH((split_hi_lo_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, x, (bf16_t*)out, M, K, ldx, ldo, (long)K);
    else return fail(LMI_EINVAL, "lmi_split_hi_lo: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_split_hi_lo");
}

int lmi_split_rows_hl(const float* x, void* out, int M, int K, int ldx, int ldo, int dtype, void* stream) {
    if (!x || !out || M < 0 || K <= 0 || (K & 7) || (ldx & 3) || (ldo & 7) || ldo < K || !aligned16(x) || !aligned16(out))
        return fail(LMI_EINVAL, "lmi_split_rows_hl: bad argument (M=%d K=%d ldx=%d ldo=%d; K %% 8 == 0, ldo >= K)", M, K, ldx, ldo);
    if (M == 0) return LMI_OK;
    const int grid = grid_for((long)M * (K >> 3), 256);
    if (dtype == LMI_F16) LMI_LAUNCH((split_hi_lo_kernel<f16_t>), dim3(grid), dim3(256), 0, stream, x, (f16_t*)out, M, K, ldx, ldo, (long)M * ldo);
    else if (dtype == LMI_BF16) LMI_LAUNCH((split_hi_lo_kernel<bf16_t>), dim3(grid), dim3(256), 0, stream, x, (bf16_t*)out, M, K, ldx, ldo, (long)M * ldo);
    else return fail(LMI_EINVAL, "lmi_split_rows_hl: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_split_rows_hl");
}

// ---- caller-owned scratch of a prefill pass ("workspace: size from lmi_*_workspace_bytes"; the library allocates nothing) ----------
// One contiguous workspace per stage holds every activation buffer that lives between the launches of one pass; `offsets` (nullable)
// receives the byte offset of each buffer (256-byte aligned; the order is the LMI_WS_* enums of the header).
static int64_t ws_carve(const int64_t* sizes, int n, int64_t* offsets) {
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (offsets) offsets[i] = off;
        off += (sizes[i] + 255) / 256 * 256;
    }
    return off;
}
int64_t lmi_llm_prefill_workspace_bytes(int64_t rows, int hidden, int n_q_heads, int n_kv_heads, int head_dim, int ff, int dtype, int64_t* offsets) {
    if (rows < 0 || hidden <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0 || ff <= 0 || (dtype != LMI_F16 && dtype != LMI_BF16)) {
        fail(LMI_EINVAL, "lmi_llm_prefill_workspace_bytes: bad argument");
        return -1;
    }
    const int64_t es = 2, parts = (hidden + 63) / 64;
    const int64_t sizes[LMI_WS_LLM_COUNT] = {
        rows * hidden * es,                                          // LMI_WS_LLM_H    normalised rows / T(x * gamma) handed to q|k|v and gate/up
        rows * (int64_t)(n_q_heads + 2 * n_kv_heads) * head_dim * es,  // LMI_WS_LLM_QKV  packed q | k | v rows
        rows * (int64_t)n_q_heads * head_dim * es,                    // LMI_WS_LLM_ATT  attention output = o_proj operand
        rows * (int64_t)ff * es,                                      // LMI_WS_LLM_GU   SwiGLU product = down_proj operand
        rows * parts * 4,                                             // LMI_WS_LLM_SQ_A row-square partials feeding gate/up (folded RMSNorm)
        rows * parts * 4};                                            // LMI_WS_LLM_SQ_B ... feeding the next layer's q|k|v
    return ws_carve(sizes, LMI_WS_LLM_COUNT, offsets);
}
int64_t lmi_vit_workspace_bytes(int64_t rows, int hidden, int qkv_width, int ff_padded, int dtype, int64_t* offsets) {
    if (rows < 0 || hidden <= 0 || qkv_width <= 0 || ff_padded <= 0 || (dtype != LMI_F16 && dtype != LMI_BF16)) {
        fail(LMI_EINVAL, "lmi_vit_workspace_bytes: bad argument");
        return -1;
    }
    const int64_t es = 2;
    const int64_t sizes[LMI_WS_VIT_COUNT] = {rows * hidden * es,     // LMI_WS_VIT_H    LayerNorm output (q|k|v / fc1 operand; the post-LN features)
                                             rows * (int64_t)qkv_width * es,   // LMI_WS_VIT_QKV
                                             rows * hidden * es,     // LMI_WS_VIT_ATT  attention output = out_proj operand
                                             rows * (int64_t)ff_padded * es};  // LMI_WS_VIT_FF   GELU(fc1) = fc2 operand
    return ws_carve(sizes, LMI_WS_VIT_COUNT, offsets);
}

int64_t lmi_attn_decode_workspace_bytes(int q_rows, int n_heads, int head_dim, int max_seqlen_k) {
    if (q_rows < 0 || n_heads <= 0 || head_dim <= 0 || max_seqlen_k < 0) return -1;
    const int tiles = (max_seqlen_k + ATT_BKV - 1) / ATT_BKV;      // the most splits any launch shape takes: one per tile, at most 64
    const int n = tiles < 2 ? 2 : (tiles > 64 ? 64 : tiles);
    return (int64_t)n * q_rows * n_heads * (head_dim + 2) * 4;
}

static int attn_decode_entry(const char* who, const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* cu_seqlens_k,
                             const int* k_len, int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                             int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                             int dtype, void* stream, int hl = 0) {
    if (!q || !k || !v || !out || !cu_seqlens_q || !cu_seqlens_k || !workspace) return fail(LMI_EINVAL, "%s: null pointer", who);
    if (n_seq < 0 || max_seqlen_q < 0 || max_seqlen_k < 0 || q_rows < 0 || n_heads <= 0 || n_kv_heads <= 0 || (n_heads % n_kv_heads))
        return fail(LMI_EINVAL, "%s: bad sizes", who);
    if (head_dim != 128) return fail(LMI_EINVAL, "%s: head_dim %d (only 128)", who, head_dim);
    if (window < 0) return fail(LMI_EINVAL, "%s: window must be >= 0", who);
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) ||
        !aligned16(workspace))
        return fail(LMI_EINVAL, "%s: alignment", who);
    const int64_t need = lmi_attn_decode_workspace_bytes(q_rows, n_heads, head_dim, max_seqlen_k);
    if (workspace_bytes < need) return fail(LMI_EINVAL, "%s: workspace %lld < %lld bytes", who, (long long)workspace_bytes, (long long)need);
    if (n_seq == 0 || max_seqlen_q == 0 || q_rows == 0) return LMI_OK;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.cu_q = cu_seqlens_q; a.cu_k = cu_seqlens_k; a.k_len = k_len;
    a.out_fp8 = nullptr; a.ldo8 = 0; a.out_fp8_scale = 0.f; a.out_f32 = nullptr; a.ldo32 = 0;
    a.out4 = nullptr; a.out4_scale = nullptr; a.ld_out4 = a.ld_out4s = 0; a.row_sel = nullptr;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.n_heads = n_heads; a.n_kv_heads = n_kv_heads; a.scale = scale; a.window = window;
    if (((long)max_seqlen_k * ldk + head_dim) * 2 >= (1L << 32) || ((long)max_seqlen_k * ldv + head_dim) * 2 >= (1L << 32))
        return fail(LMI_EINVAL, "%s: one sequence's K / V rows span >= 4 GiB (max_seqlen_k %d, ldk %d, ldv %d)", who, max_seqlen_k, ldk, ldv);
    a.check_k_extent = 0;
    a.n_splits = decode_splits(max_seqlen_k, decode_grid_heads(n_heads, n_kv_heads, max_seqlen_q), &a.split_tiles);
    a.part_rows = q_rows;
    a.part_o = (float*)workspace;
    a.part_ml = a.part_o + (size_t)a.n_splits * q_rows * n_heads * head_dim;
    LMI_DISPATCH_T(dtype, (attn_decode_impl<f16_t>(a, n_seq, max_seqlen_q, q_rows, out, ldo, stream, hl ? q_rows : 0)),
                   (attn_decode_impl<bf16_t>(a, n_seq, max_seqlen_q, q_rows, out, ldo, stream, hl ? q_rows : 0)));
}

int lmi_attn_decode_fwd(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* cu_seqlens_k,
                        int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                        int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                        int dtype, void* stream) {
    return attn_decode_entry("lmi_attn_decode_fwd", q, k, v, out, cu_seqlens_q, cu_seqlens_k, nullptr, n_seq, max_seqlen_q, max_seqlen_k, q_rows,
                             n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, window, workspace, workspace_bytes, dtype, stream);
}

int lmi_attn_decode_pool(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* k_begin, const int* k_len,
                         int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                         int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                         int dtype, void* stream) {
    if (!k_len) return fail(LMI_EINVAL, "lmi_attn_decode_pool: null k_len");
    return attn_decode_entry("lmi_attn_decode_pool", q, k, v, out, cu_seqlens_q, k_begin, k_len, n_seq, max_seqlen_q, max_seqlen_k, q_rows,
                             n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, window, workspace, workspace_bytes, dtype, stream);
}

static int gemm_skinny_entry(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed,
                             const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, int ld_norm, const float* norm_gamma,
                             float* rowsq_out, int dtype, void* stream, int hl) {
    if (!W || !X || !out) return fail(LMI_EINVAL, "lmi_gemm_skinny: null pointer");
    if (hl && 2 * M > 16) return fail(LMI_EINVAL, "lmi_gemm_skinny_hl: the hi + lo row pairs need 2 M <= 16 (M = %d)", M);
    if (M < 0 || M > 16 || N <= 0 || K <= 0 || (K % 128) || epilogue < LMI_SKINNY_STORE || epilogue > LMI_SKINNY_STORE_F32 ||
        (N % (epilogue == LMI_SKINNY_SWIGLU ? 64 : 16)))
        return fail(LMI_EINVAL, "lmi_gemm_skinny: need M <= 16, K %% 128 == 0, N %% 16 == 0 (SwiGLU: N %% 64 == 0) (M=%d N=%d K=%d)", M, N, K);
    if ((ldw & 7) || (ldx & 7) || ldw < K || ldx < K || (packed && ldw != K) || !aligned16(W) || !aligned16(X) ||
        ((epilogue == LMI_SKINNY_RESIDUAL || epilogue == LMI_SKINNY_STORE_F32) ? !aligned16(out) && ((uintptr_t)out & 3) : ((uintptr_t)out & 1)))
        return fail(LMI_EINVAL, "lmi_gemm_skinny: rows must be 16-byte aligned (ldw, ldx multiples of 8 and >= K)");
    // the output row holds N columns (SwiGLU: N / 2 products): a smaller ldo would let the residual read-modify-write run into the next row
    if (ldo < (epilogue == LMI_SKINNY_SWIGLU ? N / 2 : N))
        return fail(LMI_EINVAL, "lmi_gemm_skinny: ldo %d < output row width %d", ldo, epilogue == LMI_SKINNY_SWIGLU ? N / 2 : N);
    SkinnyNorm nm;
    if (int rc = skinny_norm_args("lmi_gemm_skinny_ex", nm, M, N, epilogue, rowsq_in, rowsq_parts, norm_dim, norm_eps, norm_out, ld_norm, norm_gamma, rowsq_out))
        return rc;
    nm.hl = hl ? 1 : 0;
    if (M == 0) return LMI_OK;
    if (packed)
        LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 1>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)),
                       (skinny_impl<bf16_t, 1>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)));
    if (g_skinny_coalesce.load())
        LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 2>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)),
                       (skinny_impl<bf16_t, 2>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)));
    LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 0>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)),
                   (skinny_impl<bf16_t, 0>(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, stream, RopeEpi(), nm)));
}

int lmi_attn_decode_fwd_hl(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* cu_seqlens_k,
                           int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                           int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                           int dtype, void* stream) {
    return attn_decode_entry("lmi_attn_decode_fwd_hl", q, k, v, out, cu_seqlens_q, cu_seqlens_k, nullptr, n_seq, max_seqlen_q, max_seqlen_k, q_rows,
                             n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, window, workspace, workspace_bytes, dtype, stream, 1);
}

int lmi_attn_decode_pool_hl(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* k_begin, const int* k_len,
                            int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                            int dtype, void* stream) {
    if (!k_len) return fail(LMI_EINVAL, "lmi_attn_decode_pool_hl: null k_len");
    return attn_decode_entry("lmi_attn_decode_pool_hl", q, k, v, out, cu_seqlens_q, k_begin, k_len, n_seq, max_seqlen_q, max_seqlen_k, q_rows,
                             n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, window, workspace, workspace_bytes, dtype, stream, 1);
}

int lmi_gemm_skinny_ex(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed,
                       const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, int ld_norm, const float* norm_gamma,
                       float* rowsq_out, int dtype, void* stream) {
    return gemm_skinny_entry(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, packed, rowsq_in, rowsq_parts, norm_dim, norm_eps, norm_out, ld_norm, norm_gamma,
                             rowsq_out, dtype, stream, 0);
}

int lmi_gemm_skinny_hl(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed,
                       const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, int ld_norm, const float* norm_gamma,
                       float* rowsq_out, int dtype, void* stream) {
    return gemm_skinny_entry(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, packed, rowsq_in, rowsq_parts, norm_dim, norm_eps, norm_out, ld_norm, norm_gamma,
                             rowsq_out, dtype, stream, 1);
}

int lmi_gemm_skinny(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed, int dtype,
                    void* stream) {
    return lmi_gemm_skinny_ex(W, X, out, M, N, K, ldw, ldx, ldo, epilogue, packed, nullptr, 0, 0, 0.f, nullptr, 0, nullptr, nullptr, dtype, stream);
}

static int rope_qkv_skinny_entry(const void* Wqkv_rope, const void* X, void* qkv, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int ldw, int ldx,
                                 int ldo, int packed, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_all, const float* sin_all,
                                 void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream, int hl) {
    if (hl && 2 * M > 16) return fail(LMI_EINVAL, "lmi_rope_qkv_skinny_hl: the hi + lo row pairs need 2 M <= 16 (M = %d)", M);
    if (!Wqkv_rope || !X || !qkv || !cos_all || !sin_all || !k_cache || !v_cache || !pos_rows_dev) return fail(LMI_EINVAL, "lmi_rope_qkv_skinny: null pointer");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_rope_qkv_skinny: head_dim %d (only 128)", head_dim);
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    if (M < 0 || M > 16 || n_q_heads <= 0 || n_kv_heads <= 0 || K <= 0 || (K % 128) || (ldw & 7) || (ldx & 7) || ldw < K || ldx < K || (packed && ldw != K) ||
        (ld_cache & 7) || cache_stride <= 0 || ldo < N || !aligned16(Wqkv_rope) || !aligned16(X))
        return fail(LMI_EINVAL, "lmi_rope_qkv_skinny: bad argument (M <= 16, K %% 128 == 0, 16-byte aligned rows)");
    SkinnyNorm nm;
    if (int rc = skinny_norm_args("lmi_rope_qkv_skinny", nm, M, N, 4, rowsq_in, rowsq_parts, K, norm_eps, nullptr, 0, nullptr, nullptr)) return rc;
    nm.hl = hl ? 1 : 0;
    if (M == 0) return LMI_OK;
    RopeEpi rp;
    rp.cos_all = cos_all; rp.sin_all = sin_all; rp.pos = pos_rows_dev; rp.k_cache = k_cache; rp.v_cache = v_cache; rp.ld_cache = ld_cache;
    rp.cache_stride = (long)cache_stride; rp.rope_q = n_q_heads * head_dim; rp.rope_k = n_kv_heads * head_dim;
    if (packed)
        LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 1>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)),
                       (skinny_impl<bf16_t, 1>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)));
    if (g_skinny_coalesce.load())
        LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 2>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)),
                       (skinny_impl<bf16_t, 2>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)));
    LMI_DISPATCH_T(dtype, (skinny_impl<f16_t, 0>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)),
                   (skinny_impl<bf16_t, 0>(Wqkv_rope, X, qkv, M, N, K, ldw, ldx, ldo, 4, stream, rp, nm)));
}

int lmi_rope_qkv_skinny(const void* Wqkv_rope, const void* X, void* qkv, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int ldw, int ldx,
                        int ldo, int packed, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_all, const float* sin_all,
                        void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream) {
    return rope_qkv_skinny_entry(Wqkv_rope, X, qkv, M, n_q_heads, n_kv_heads, head_dim, K, ldw, ldx, ldo, packed, rowsq_in, rowsq_parts, norm_eps, cos_all,
                                 sin_all, k_cache, v_cache, ld_cache, cache_stride, pos_rows_dev, dtype, stream, 0);
}

int lmi_rope_qkv_skinny_hl(const void* Wqkv_rope, const void* X, void* qkv, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int ldw, int ldx,
                           int ldo, int packed, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_all, const float* sin_all,
                           void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream) {
    return rope_qkv_skinny_entry(Wqkv_rope, X, qkv, M, n_q_heads, n_kv_heads, head_dim, K, ldw, ldx, ldo, packed, rowsq_in, rowsq_parts, norm_eps, cos_all,
                                 sin_all, k_cache, v_cache, ld_cache, cache_stride, pos_rows_dev, dtype, stream, 1);
}

int lmi_rope_qk_rows(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_all, const float* sin_all,
                     void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream) {
    if (!qkv || !cos_all || !sin_all || !k_cache || !v_cache || !pos_rows_dev || S < 0 || (head_dim & 15) || (ld & 7) || (ld_cache & 7) ||
        cache_stride <= 0 || !aligned16(qkv) || !aligned16(k_cache) || !aligned16(v_cache))
        return fail(LMI_EINVAL, "lmi_rope_qk_rows: bad argument");
    if (S == 0) return LMI_OK;
    const long work = (long)S * ((n_q_heads + n_kv_heads) * (head_dim / 16) + n_kv_heads * head_dim / 8);
    const int grid = grid_for(work, 256);
    LMI_DISPATCH_T(dtype, (rope_rows_impl<f16_t>(qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, ld_cache, (long)cache_stride, pos_rows_dev, grid, stream)),
                   (rope_rows_impl<bf16_t>(qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, ld_cache, (long)cache_stride, pos_rows_dev, grid, stream)));
}

int lmi_rope_qk(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_table,
                const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int dtype, void* stream) {
    return rope_entry("lmi_rope_qk", qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_table, sin_table, k_cache, v_cache, ld_cache,
                      cache_pos0, nullptr, dtype, stream);
}

int lmi_rope_qk_at(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_all,
                   const float* sin_all, void* k_cache, void* v_cache, int ld_cache, const int* pos_dev, int dtype, void* stream) {
    if (!pos_dev) return fail(LMI_EINVAL, "lmi_rope_qk_at: null position pointer");
    return rope_entry("lmi_rope_qk_at", qkv, S, ld, n_q_heads, n_kv_heads, head_dim, cos_all, sin_all, k_cache, v_cache, ld_cache, 0,
                      pos_dev, dtype, stream);
}

int lmi_embed_merge(const int64_t* ids, const int64_t* src, const void* embed_table, const float* visual_tokens, float* out,
                    int S, int D, int ld_feats, int dtype, void* stream) {
    if (!ids || !src || !embed_table || !out || S < 0 || (D & 7) || (ld_feats & 3)) return fail(LMI_EINVAL, "lmi_embed_merge: bad argument");
    if (S == 0) return LMI_OK;
    LMI_DISPATCH_T(dtype, (merge_impl<f16_t>(ids, src, embed_table, visual_tokens, out, S, D, ld_feats, stream)),
                   (merge_impl<bf16_t>(ids, src, embed_table, visual_tokens, out, S, D, ld_feats, stream)));
}

int lmi_decode_advance(const float* logits, int B, int vocab, int ld_logits, const int64_t* suppress, int n_suppress, int64_t* tok, int* pos,
                       int* k_len, int* live, int* budget, const int64_t* eos, int n_eos, int64_t* hist, int* hist_pos, int hist_len, void* stream) {
    if (!logits || !tok || !pos || B < 0 || vocab <= 0 || ld_logits < vocab || n_suppress < 0 || (n_suppress && !suppress) || n_eos < 0 ||
        (n_eos && !eos) || (hist && (!hist_pos || hist_len <= 0)))
        return fail(LMI_EINVAL, "lmi_decode_advance: bad argument");
    if (B == 0) return LMI_OK;
    DecodeAdvanceArgs a;
    a.logits = logits; a.vocab = vocab; a.ld_logits = ld_logits; a.suppress = suppress; a.n_suppress = n_suppress; a.tok = tok; a.pos = pos;
    a.k_len = k_len; a.live = live; a.budget = budget; a.eos = eos; a.n_eos = n_eos; a.hist = hist; a.hist_pos = hist_pos; a.hist_len = hist_len;
    a.B = B;
    LMI_LAUNCH(decode_advance_kernel, dim3(B), dim3(1024), 0, stream, a);
    return check_launch("lmi_decode_advance");
}

int lmi_gemv(const void* W, const void* x, const float* bias, void* out, int N, int K, int ldw, int epilogue, int dtype,
             void* stream) {
    if (!W || !x || !out || N <= 0 || K <= 0 || (K & 7) || (ldw & 7) || K > 28 * 512 || (epilogue == 3 && (N & 63)))
        return fail(LMI_EINVAL, "lmi_gemv: bad argument (N=%d K=%d)", N, K);
    LMI_DISPATCH_T(dtype, (dispatch_gemv<f16_t, false>(W, x, nullptr, 0.f, bias, out, N, K, ldw, epilogue, stream)),
                   (dispatch_gemv<bf16_t, false>(W, x, nullptr, 0.f, bias, out, N, K, ldw, epilogue, stream)));
}

int lmi_lm_head_last(const void* W, const float* x, const int64_t* rows, const float* norm_weight, float eps, float* out, int n_rows,
                     int N, int K, int ldw, int ldx, int ldo, int dtype, void* stream) {
    if (!W || !x || !out || n_rows < 0 || N <= 0 || K <= 0 || (K & 7) || (ldw & 7) || (ldx & 3) || ldo < N || !aligned16(W) ||
        !aligned16(x) || (norm_weight && !aligned16(norm_weight)))
        return fail(LMI_EINVAL, "lmi_lm_head_last: bad argument (n_rows=%d N=%d K=%d; K %% 8 == 0, ldo >= N)", n_rows, N, K);
    const int lds = (K + 4) * 4;
    if (lds > 64 * 1024) return fail(LMI_EINVAL, "lmi_lm_head_last: K = %d does not fit the LDS row buffer (max 16380)", K);
    if (n_rows == 0) return LMI_OK;
    int rpw = 32;                                                  // vocabulary rows per workgroup: about 2048 workgroups per selected row
    while ((N + rpw - 1) / rpw > 2048) rpw *= 2;
    const dim3 grid((N + rpw - 1) / rpw, n_rows);
    if (dtype == LMI_F16)
        LMI_LAUNCH((lm_head_rows_kernel<f16_t, 8>), grid, dim3(256), lds, stream, (const f16_t*)W, x, (const long*)rows, norm_weight, eps, out,
                   N, K, ldw, ldx, ldo, rpw);
    else if (dtype == LMI_BF16)
        LMI_LAUNCH((lm_head_rows_kernel<bf16_t, 8>), grid, dim3(256), lds, stream, (const bf16_t*)W, x, (const long*)rows, norm_weight, eps,
                   out, N, K, ldw, ldx, ldo, rpw);
    else
        return fail(LMI_EINVAL, "lmi_lm_head_last: dtype must be LMI_F16 or LMI_BF16");
    return check_launch("lmi_lm_head_last");
}

int lmi_gemv_rmsnorm(const void* W, const float* x, const float* norm_weight, float eps, void* out, int N, int K, int ldw,
                     int epilogue, int dtype, void* stream) {
    if (!W || !x || !norm_weight || !out || N <= 0 || K <= 0 || (ldw & 7) || (epilogue == 3 && (N & 63)) || !aligned16(x) ||
        !aligned16(norm_weight))
        return fail(LMI_EINVAL, "lmi_gemv_rmsnorm: bad argument (N=%d K=%d)", N, K);
    LMI_DISPATCH_T(dtype, (dispatch_gemv<f16_t, true>(W, x, norm_weight, eps, nullptr, out, N, K, ldw, epilogue, stream)),
                   (dispatch_gemv<bf16_t, true>(W, x, norm_weight, eps, nullptr, out, N, K, ldw, epilogue, stream)));
}

int lmi_gemv_rmsnorm_rope(const void* Wqkv_rope, const float* x, const float* norm_weight, float eps, void* qkv, int n_q_heads, int n_kv_heads,
                          int head_dim, int K, int ldw, const float* cos_all, const float* sin_all, void* k_cache, void* v_cache, int ld_cache,
                          const int* pos_dev, int dtype, void* stream) {
    if (!Wqkv_rope || !x || !norm_weight || !qkv || !cos_all || !sin_all || !k_cache || !v_cache || !pos_dev)
        return fail(LMI_EINVAL, "lmi_gemv_rmsnorm_rope: null pointer");
    if (head_dim != 128) return fail(LMI_EINVAL, "lmi_gemv_rmsnorm_rope: head_dim %d (only 128)", head_dim);
    if (n_q_heads <= 0 || n_kv_heads <= 0 || K != 4096 || (ldw & 7) || ldw < K || ld_cache < n_kv_heads * head_dim || !aligned16(x) ||
        !aligned16(norm_weight) || !aligned16(Wqkv_rope))
        return fail(LMI_EINVAL, "lmi_gemv_rmsnorm_rope: bad argument (K = %d: only the 4096 hidden size; 16-byte aligned rows)", K);
    const int N = (n_q_heads + 2 * n_kv_heads) * head_dim;
    RopeEpi rp;
    rp.cos_all = cos_all; rp.sin_all = sin_all; rp.pos = pos_dev; rp.k_cache = k_cache; rp.v_cache = v_cache; rp.ld_cache = ld_cache;
    rp.cache_stride = 0; rp.rope_q = n_q_heads * head_dim; rp.rope_k = n_kv_heads * head_dim;
    LMI_DISPATCH_T(dtype, (launch_gemv<f16_t, GEMV_QKV_ROPE_T, true>(Wqkv_rope, x, norm_weight, eps, nullptr, qkv, N, K, ldw, stream, rp)),
                   (launch_gemv<bf16_t, GEMV_QKV_ROPE_T, true>(Wqkv_rope, x, norm_weight, eps, nullptr, qkv, N, K, ldw, stream, rp)));
}

}  // extern "C"

// ---- RCCL collectives over xGMI ------------------------------------------------------------------------
// librccl is bound at run time with dlopen: the library has no link-time dependency on it (single-GPU users never load it),
// and inside a PyTorch process the copy PyTorch already mapped is reused instead of a second one.  One communicator per
// process / GPU (one process per GPU, as the reference launches its evaluation, run_eval_llava_siglip_multiimg.sh:9-11).
#include <dlfcn.h>
namespace {
struct RcclUid { char internal[128]; };                       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
enum { RCCL_SUM = 0, RCCL_F16 = 6, RCCL_F32 = 7, RCCL_BF16 = 9 };   // ncclRedOp_t / ncclDataType_t values of rccl.h
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::atomic<int> g_rccl_state{0};                              // 0 = not tried, 1 = bound, -1 = unavailable

std::once_flag g_rccl_once;
void rccl_bind_once() {
#ifdef LMI_EMU
    g_rccl_state = -1;
#else
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;     // a copy already in the process (PyTorch's)
    if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { g_rccl_state = -1; return; }
    Rccl r;
    r.handle = h;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) ok = false; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.ReduceScatter = (decltype(r.ReduceScatter))sym("ncclReduceScatter");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { g_rccl_state = -1; return; }
    g_rccl = r;                                                    // written once, under the once_flag; readers see it after the state store
    g_rccl_state.store(1, std::memory_order_release);
#endif
}
int rccl_bind() {                                                  // thread-safe: two threads' first collectives cannot both write g_rccl
    std::call_once(g_rccl_once, rccl_bind_once);
    return g_rccl_state.load(std::memory_order_acquire);
}
int rccl_fail(const char* what, int rc) {
    return fail(LMI_ECOMM, "%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
}
int rccl_dtype(int dtype) { return dtype == LMI_F16 ? RCCL_F16 : dtype == LMI_BF16 ? RCCL_BF16 : dtype == LMI_F32 ? RCCL_F32 : -1; }
#define LMI_NEED_RCCL(who) do { if (rccl_bind() != 1) return fail(LMI_ECOMM, "%s: librccl could not be loaded (dlopen)", who); } while (0)
}  // namespace

extern "C" {

int lmi_comm_unique_id(void* id128) {
    if (!id128) return fail(LMI_EINVAL, "lmi_comm_unique_id: null pointer");
    LMI_NEED_RCCL("lmi_comm_unique_id");
    RcclUid id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc) return rccl_fail("lmi_comm_unique_id", rc);
    memcpy(id128, id.internal, 128);
    return LMI_OK;
}

int lmi_comm_init(int rank, int nranks, const void* id128, void** comm_out) {
    if (!id128 || !comm_out || nranks <= 0 || rank < 0 || rank >= nranks) return fail(LMI_EINVAL, "lmi_comm_init: bad argument");
    LMI_NEED_RCCL("lmi_comm_init");
    RcclUid id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);      // binds the CURRENT HIP device to this rank
    if (rc) return rccl_fail("lmi_comm_init", rc);
    int n = -1;
    if (g_rccl.CommCount(comm, &n) || n != nranks) { g_rccl.CommDestroy(comm); return fail(LMI_ECOMM, "lmi_comm_init: communicator reports %d ranks, expected %d", n, nranks); }
    *comm_out = comm;
    return LMI_OK;
}

int lmi_comm_destroy(void* comm) {
    if (!comm) return LMI_OK;
    LMI_NEED_RCCL("lmi_comm_destroy");
    const int rc = g_rccl.CommDestroy(comm);
    return rc ? rccl_fail("lmi_comm_destroy", rc) : LMI_OK;
}

int lmi_comm_size(void* comm) {
    if (!comm || rccl_bind() != 1) return -1;
    int n = -1;
    return g_rccl.CommCount(comm, &n) ? -1 : n;
}

int lmi_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream) {
    if (!comm || !send || !recv || count_per_rank < 0 || rccl_dtype(dtype) < 0) return fail(LMI_EINVAL, "lmi_allgather: bad argument");
    LMI_NEED_RCCL("lmi_allgather");
    const int rc = g_rccl.AllGather(send, recv, (size_t)count_per_rank, rccl_dtype(dtype), comm, stream);
    return rc ? rccl_fail("lmi_allgather", rc) : LMI_OK;
}

int lmi_allreduce(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream) {
    if (!comm || !send || !recv || count < 0 || rccl_dtype(dtype) < 0) return fail(LMI_EINVAL, "lmi_allreduce: bad argument");
    LMI_NEED_RCCL("lmi_allreduce");
    const int rc = g_rccl.AllReduce(send, recv, (size_t)count, rccl_dtype(dtype), RCCL_SUM, comm, stream);
    return rc ? rccl_fail("lmi_allreduce", rc) : LMI_OK;
}

int lmi_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype, void* stream) {
    if (!comm || !send || !recv || recv_count < 0 || rccl_dtype(dtype) < 0) return fail(LMI_EINVAL, "lmi_reduce_scatter: bad argument");
    LMI_NEED_RCCL("lmi_reduce_scatter");
    const int rc = g_rccl.ReduceScatter(send, recv, (size_t)recv_count, rccl_dtype(dtype), RCCL_SUM, comm, stream);
    return rc ? rccl_fail("lmi_reduce_scatter", rc) : LMI_OK;
}

int lmi_broadcast(void* comm, const void* send, void* recv, int64_t count, int dtype, int root, void* stream) {
    if (!comm || !send || !recv || count < 0 || rccl_dtype(dtype) < 0) return fail(LMI_EINVAL, "lmi_broadcast: bad argument");
    LMI_NEED_RCCL("lmi_broadcast");
    const int rc = g_rccl.Broadcast(send, recv, (size_t)count, rccl_dtype(dtype), root, comm, stream);
    return rc ? rccl_fail("lmi_broadcast", rc) : LMI_OK;
}

}  // extern "C"
