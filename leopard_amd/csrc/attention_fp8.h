// attention_fp8.h — causal self-attention of the fp8 schedule with BOTH products on the fp8 matrix pipe (BASELINE configs[4]:
// "fp8 MFMA ViT+LLM prefill"; VERDICT r03 item 9: QK^T / PV on v_mfma_scale_f32_32x32x64_f8f6f4).  Llama / Mistral shape only:
// head_dim 128, GQA, one sequence = queries and keys of the same rows (megatron_patch/model/llava/transformer.py:678-885).
//
// Two kernels:
//   attn_prep_fp8_kernel   q | k | v rows (16-bit, rotated) -> e4m3 operands in the layouts the attention wants: q8 rows as they are,
//                          K and V as ready-made 8-KiB LDS IMAGES per (kv head, 64-key tile) — K [64 keys][128 B] with the bank
//                          swizzle applied, V TRANSPOSED to [128 d][64 key slots] with the keys in the order the score accumulators
//                          hold them — so that a tile is one contiguous 8-KiB LDS-DMA copy and no transposing read is needed.
//   attn_fwd_fp8_kernel    the register-level design of attention.h (S^T = K . Q^T, O^T = V^T . P^T, everything of query row q in lane
//                          q & 31 and its half-wave partner), per 64-key tile 4 + 4 MFMAs of 32x32x64 instead of 16 + 16 of 32x32x16:
//                          half the matrix-pipe cycles, half the LDS bytes, half the DMA pieces.  P is rounded to e4m3 (values <= 2^8:
//                          the deferred softmax reference keeps them in range), row sums and O accumulate in fp32.
// Key order of a tile.  The score accumulator of lane (q, hi) holds, for 32-key block b and register r, key b*32 + (r&3) + 8*(r>>2) + 4*hi.
// The PV product contracts over 64 key SLOTS; lane (q, hi) supplies slots hi*32 + j, j = b*16 + r — its own P registers, unshuffled.  The V
// image therefore stores, in row d, slot hi*32 + b*16 + r = V[key(b, r, hi)][d].
#pragma once
#include "attention.h"

namespace lmi {

constexpr int ATT8_IMG = ATT_BKV * 128;              // bytes of one K or V tile image (64 keys x 128 e4m3)

// key (0..63) of slot s (0..63) of a tile
LMI_DEV int att8_slot_key(int s) {
    const int hi = s >> 5, j = s & 31, b = j >> 4, r = j & 15;
    return b * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
}
// byte offset of 16-byte chunk c (0..7) of key row r in the K image; of 16-byte slot chunk c (0..3) of d row d in the V image
LMI_DEV int att8_k_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
LMI_DEV int att8_v_off(int d, int c) { return d * 64 + ((c ^ ((d >> 2) & 3)) << 4); }

// four fp32 -> one dword of e4m3 (first value in the low byte).  The hardware conversion does not saturate (an overflow becomes the NaN
// code): pack4_fp8 is for values known to be in range (P <= 2^8), pack4_fp8_sat clamps to +-448 first, as to_fp8 does.
LMI_DEV unsigned pack4_fp8(float a, float b, float c, float d) {
#ifndef LMI_EMU
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
#else
    return (unsigned)to_fp8(a) | ((unsigned)to_fp8(b) << 8) | ((unsigned)to_fp8(c) << 16) | ((unsigned)to_fp8(d) << 24);
#endif
}

LMI_DEV float sat448(float x) { return fminf(fmaxf(x, -448.0f), 448.0f); }
LMI_DEV unsigned pack4_fp8_sat(float a, float b, float c, float d) { return pack4_fp8(sat448(a), sat448(b), sat448(c), sat448(d)); }

struct AttnPrepArgs {
    const void* qkv;          // T [rows, ld]: q heads | k heads | v heads (rotated)
    const int* cu;            // [n_seq + 1]
    const int* tile_base;     // [n_seq + 1]: first 64-key tile of sequence s in the images; tile_base[n_seq] = n_tiles
    uint8_t* q8;              // [rows, ldq8]
    uint8_t* k_img;           // [n_kv_heads][n_tiles][ATT8_IMG]
    uint8_t* v_img;
    int ld, ldq8, n_seq, n_tiles, n_heads, n_kv_heads;
    float q_scale, k_scale, v_scale;      // operand = e4m3(value * scale), saturating
};

// grid = n_tiles x (2 n_kv_heads + n_heads): y < n_kv_heads: the K image of (head y, tile x); y < 2 n_kv_heads: the V image; else the q8 rows of
// query head y - 2 n_kv_heads for the 64 rows of tile x.  256 threads.
template <typename T>
__global__ void __launch_bounds__(256) attn_prep_fp8_kernel(AttnPrepArgs p) {
    typedef typename vec_of<T>::x8 T8;
    __shared__ __attribute__((aligned(16))) uint8_t vt[64][128 + 16];
    const int tid = threadIdx.x, tile = blockIdx.x, y = blockIdx.y;
    int seq = 0;
    while (seq + 1 < p.n_seq && tile >= p.tile_base[seq + 1]) ++seq;
    const int row_beg = p.cu[seq], len = p.cu[seq + 1] - row_beg;
    const int r0 = (tile - p.tile_base[seq]) * ATT_BKV;            // first row of this tile inside its sequence
    const int KV = p.n_kv_heads, H = p.n_heads;
    const int r = tid >> 2, d0 = (tid & 3) * 32;                   // this thread: row r of the tile, 32 elements from d0
    const bool live = r0 + r < len;
    const long row = row_beg + r0 + (live ? r : 0);
    int col;
    float scale;
    if (y < KV) { col = (H + y) * 128; scale = p.k_scale; }
    else if (y < 2 * KV) { col = (H + KV + (y - KV)) * 128; scale = p.v_scale; }
    else { col = (y - 2 * KV) * 128; scale = p.q_scale; }
    u32x4 out[2];                                                  // 32 e4m3 bytes
    {
        const T* src = (const T*)p.qkv + row * p.ld + col + d0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const T8 a = *(const T8*)(src + h * 16), b = *(const T8*)(src + h * 16 + 8);
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                w[e] = live ? pack4_fp8_sat((float)a[4 * e] * scale, (float)a[4 * e + 1] * scale, (float)a[4 * e + 2] * scale, (float)a[4 * e + 3] * scale) : 0u;
                w[2 + e] = live ? pack4_fp8_sat((float)b[4 * e] * scale, (float)b[4 * e + 1] * scale, (float)b[4 * e + 2] * scale, (float)b[4 * e + 3] * scale) : 0u;
            }
            out[h] = u32x4{w[0], w[1], w[2], w[3]};
        }
    }
    if (y >= 2 * KV) {                                             // q8: plain rows
        if (live) {
            uint8_t* dst = p.q8 + row * p.ldq8 + (y - 2 * KV) * 128 + d0;
            *(u32x4*)dst = out[0];
            *(u32x4*)(dst + 16) = out[1];
        }
        return;
    }
    if (y < KV) {                                                  // K image: row-major with the read swizzle applied
        uint8_t* img = p.k_img + ((long)y * p.n_tiles + tile) * ATT8_IMG;
        *(u32x4*)(img + att8_k_off(r, (d0 >> 4))) = out[0];
        *(u32x4*)(img + att8_k_off(r, (d0 >> 4) + 1)) = out[1];
        return;
    }
    // V image: transpose through LDS, keys in slot order
    *(u32x4*)&vt[r][d0] = out[0];
    *(u32x4*)&vt[r][d0 + 16] = out[1];
    __syncthreads();
    uint8_t* img = p.v_img + ((long)(y - KV) * p.n_tiles + tile) * ATT8_IMG;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * 256;                           // 128 d rows x 4 slot chunks
        const int d = item >> 2, c = item & 3;
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (unsigned)vt[att8_slot_key(c * 16 + e * 4 + b)][d] << (8 * b);
            w[e] = v;
        }
        *(u32x4*)(img + att8_v_off(d, c)) = u32x4{w[0], w[1], w[2], w[3]};
    }
}

struct AttnFp8Args {
    const uint8_t* q8;        // [rows, ldq8] e4m3, head h at + h * 128
    const uint8_t* k_img;     // [n_kv_heads][n_tiles][ATT8_IMG]
    const uint8_t* v_img;
    void* out;                // T [rows, ldo] (or null: out_fp8)
    uint8_t* out_fp8;         // e4m3(O * out_fp8_scale) [rows, ldo8]: the o_proj operand of the fp8 schedule
    const int* cu;            // [n_seq + 1]
    const int* tile_base;     // [n_seq + 1]
    int ldq8, ldo, ldo8, n_heads, n_kv_heads, n_tiles, n_qblocks;
    float c2;                 // softmax scale * log2(e) / (q_scale * k_scale): applied to the raw e4m3 products
    float inv_v_scale;        // 1 / v_scale
    float out_fp8_scale;
};

template <typename T, bool CAUSAL>
__global__ void __launch_bounds__(ATT_THREADS, 2) attn_fwd_fp8_kernel(AttnFp8Args p) {
    constexpr int D = 128, NW = ATT_THREADS / 64, NDB = 4, BQ = ATT_BQ;
    LMI_DYN_SMEM(smem);                                            // 2 slots x (K image, V image) = 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int fr = lane & 31, fh = lane >> 5;
    // heads fastest (attention.h: heavy causal blocks first, one kv head's stream per XCD)
    const int bid = (int)blockIdx.x;
    const int h_idx = bid % p.n_heads, rest = bid / p.n_heads;
    const int qb = p.n_qblocks - 1 - rest % p.n_qblocks, seq = rest / p.n_qblocks;
    const int kvh = h_idx % p.n_kv_heads;
    const int head = kvh * (p.n_heads / p.n_kv_heads) + h_idx / p.n_kv_heads;
    const int q_beg = p.cu[seq], len = p.cu[seq + 1] - q_beg;
    const int q0 = qb * BQ;
    if (q0 >= len) return;
    const int kv_end = CAUSAL ? imin(len, q0 + BQ) : len;
    const int n_tiles = (kv_end + ATT_BKV - 1) / ATT_BKV;
    const int wave_q_lo = q0 + wave * 32, wave_q_hi = wave_q_lo + 31;
    int my_tiles = CAUSAL ? imax(0, imin(n_tiles, wave_q_hi / ATT_BKV + 1)) : n_tiles;
    if (wave_q_lo >= len) my_tiles = 0;                            // no real row in this wave: feed and sync only
    const int my_q = wave_q_lo + fr;

    // Q fragments: lane (q, hi) holds Q[q][64 kk + 32 hi .. + 31] for the two 64-deep steps kk
    v8i qf[2];
    {
        const uint8_t* q_row = p.q8 + (long)(q_beg + imin(my_q, len - 1)) * p.ldq8 + head * D;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 lo = *(const u32x4*)(q_row + kk * 64 + fh * 32), hi = *(const u32x4*)(q_row + kk * 64 + fh * 32 + 16);
            qf[kk] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
    }
    // LDS-DMA: a tile image is 8 contiguous 1-KiB pieces; wave w copies pieces w and w + 4 of K and of V
    const int t0 = p.tile_base[seq];
    const BufRsrc k_buf = make_buf(p.k_img + (long)kvh * p.n_tiles * ATT8_IMG, (unsigned)p.n_tiles * ATT8_IMG);
    const BufRsrc v_buf = make_buf(p.v_img + (long)kvh * p.n_tiles * ATT8_IMG, (unsigned)p.n_tiles * ATT8_IMG);
    auto issue_tile = [&](int t, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + NW * i;
            char* dst = smem + slot * 2 * ATT8_IMG + piece * 1024;
            glds16_buf(k_buf, (unsigned)(piece * 1024 + lane * 16), (unsigned)(t0 + t) * ATT8_IMG, dst);
            glds16_buf(v_buf, (unsigned)(piece * 1024 + lane * 16), (unsigned)(t0 + t) * ATT8_IMG, dst + ATT8_IMG);
        }
    };
    // fragment read offsets (lane-only)
    int k_off[2][2][2], v_off[NDB][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 2; ++e) k_off[b][kk][e] = att8_k_off(b * 32 + fr, kk * 4 + fh * 2 + e);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int e = 0; e < 2; ++e) v_off[db][e] = att8_v_off(db * 32 + fr, fh * 2 + e);

    f32x16 o_acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = p.c2;

    if (n_tiles > 0) issue_tile(0, 0);
    int t = 0;
    for (; t < my_tiles; ++t) {
        wait_vmcnt_barrier<0>();                                   // tile t landed; every wave is done with tile t - 1
        if (t + 1 < n_tiles) issue_tile(t + 1, (t + 1) & 1);
        const int kv0 = t * ATT_BKV;
        const char* k_lds = smem + (t & 1) * 2 * ATT8_IMG;
        const char* v_lds = k_lds + ATT8_IMG;
        // ---- S^T = K . Q^T (raw e4m3 products: the scales ride in c2) ----------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const u32x4 lo = *(const u32x4*)(k_lds + k_off[b][kk][0]), hi = *(const u32x4*)(k_lds + k_off[b][kk][1]);
                const v8i kf = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                s[b] = mfma32_fp8(kf, qf[kk], s[b], 0x7f7f7f7f);
            }
        }
        const bool need_mask = (kv0 + ATT_BKV > len) || (CAUSAL && (kv0 + ATT_BKV - 1 > wave_q_lo));
        if (need_mask) {
            const int lim = CAUSAL ? imin(len - 1, my_q) : len - 1;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key > lim) s[b][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][r]);
        mx = xhalf_max(mx);
        // deferred reference (attention.h): P <= 2^ATT_DEFER_LOG2 = 256 <= 448, in range of e4m3
        const float m_cand = fmaxf(m_run, mx);
        if (wave_any((m_cand - m_run) * c2 > ATT_DEFER_LOG2)) {
            const float alpha = fast_exp2((m_run - ((m_cand == -INFINITY) ? 0.f : m_cand)) * c2);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
            m_run = m_cand;
        }
        const float mc = ((m_run == -INFINITY) ? 0.f : m_run) * c2;
        float psum = 0.f;
        unsigned pw[8];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[e] = fast_exp2(s[b][4 * g + e] * c2 - mc);
                    psum += pv[e];
                }
                pw[b * 4 + g] = pack4_fp8(pv[0], pv[1], pv[2], pv[3]);
            }
        l_run += psum;
        const v8i pf = v8i{(int)pw[0], (int)pw[1], (int)pw[2], (int)pw[3], (int)pw[4], (int)pw[5], (int)pw[6], (int)pw[7]};
        // ---- O^T += V^T . P^T ---------------------------------------------------------------------------------------------
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const u32x4 lo = *(const u32x4*)(v_lds + v_off[db][0]), hi = *(const u32x4*)(v_lds + v_off[db][1]);
            const v8i vf = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            o_acc[db] = mfma32_fp8(vf, pf, o_acc[db], 0x7f7f7f7f);
        }
    }
    for (; t < n_tiles; ++t) {                                     // keep feeding / syncing for the waves below the diagonal
        wait_vmcnt_barrier<0>();
        if (t + 1 < n_tiles) issue_tile(t + 1, (t + 1) & 1);
    }

    // ---- finish: O / l / v_scale; the two half-wave lanes of a row trade 4-element groups (attention.h) ----------------------
    const float l_tot = xhalf_sum(l_run);
    const float inv = (l_tot > 0.f ? 1.0f / l_tot : 0.f) * p.inv_v_scale;
    if (p.out_fp8) {
        const float sc = inv * p.out_fp8_scale;
        uint8_t* o8 = p.out_fp8 + (long)(q_beg + imin(my_q, len - 1)) * p.ldo8 + head * D;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                unsigned a = pack4_fp8_sat(o_acc[db][8 * qp] * sc, o_acc[db][8 * qp + 1] * sc, o_acc[db][8 * qp + 2] * sc, o_acc[db][8 * qp + 3] * sc);
                unsigned b = pack4_fp8_sat(o_acc[db][8 * qp + 4] * sc, o_acc[db][8 * qp + 5] * sc, o_acc[db][8 * qp + 6] * sc, o_acc[db][8 * qp + 7] * sc);
                swap_hi_lo(a, b);
                const int d = db * 32 + 16 * qp + 8 * fh;
                if (my_q < len) *(u32x2*)(o8 + d) = u32x2{a, b};
            }
        return;
    }
    T* o_row = (T*)p.out + (long)(q_beg + imin(my_q, len - 1)) * p.ldo + head * D;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            unsigned a[2], b[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                a[w] = pack2<T>(o_acc[db][8 * qp + 2 * w] * inv, o_acc[db][8 * qp + 2 * w + 1] * inv);
                b[w] = pack2<T>(o_acc[db][8 * qp + 4 + 2 * w] * inv, o_acc[db][8 * qp + 4 + 2 * w + 1] * inv);
                swap_hi_lo(a[w], b[w]);
            }
            const int d = db * 32 + 16 * qp + 8 * fh;
            if (my_q < len) *(u32x4*)(o_row + d) = u32x4{a[0], a[1], b[0], b[1]};
        }
}

}  // namespace lmi
