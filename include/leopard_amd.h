/* leopard_amd.h — C ABI of libleopard_amd.so: the MI355X (gfx950) kernels behind Leopard's multi-image
 * prefill path.
 *
 * The reference has NO native/FFI layer on this path: it is Python over `transformers` + PyTorch
 * (SURVEY.md 0.1).  Each entry point below therefore replaces a *Python call site* of the reference; the
 * citation gives the file:line of that call site (EVAL = evaluations/models/llava_multiimg_siglip_anyres.py)
 * and, where the arithmetic lives in third-party code, the in-tree Megatron analogue that states the same math.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - raw device pointers + explicit sizes / leading dimensions (in ELEMENTS) + dtype enum + hipStream_t
 *     (passed as void*); no torch types.  All pointers must be 16-byte aligned, leading dimensions multiples of 8.
 *   - stream-ordered, no implicit synchronisation, no allocation: the caller owns every buffer.  Entry points may be called
 *     concurrently from several host threads (one stream / device each); the only process-global state is lmi_set_option's
 *     experiment knobs, which are not meant to be changed while launches are in flight.
 *   - returns 0 on success, a negative LMI_E* code otherwise; lmi_last_error() gives a thread-local message.
 *   - "T" below is the 16-bit compute type selected by `dtype` (LMI_F16 or LMI_BF16); accumulation, softmax
 *     statistics, normalisation statistics and the residual stream are fp32.
 */
#ifndef LEOPARD_AMD_H
#define LEOPARD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMI_OK 0
#define LMI_EINVAL (-1)   /* bad argument (shape / alignment / enum) */
#define LMI_ELAUNCH (-2)  /* HIP launch error */
#define LMI_ECOMM (-3)    /* RCCL unavailable or a collective failed */

enum { LMI_F16 = 0, LMI_BF16 = 1, LMI_F32 = 2, LMI_FP8 = 3 /* OCP e4m3fn bytes; fp8 entry points only */ };

/* epilogues of lmi_gemm */
enum {
    LMI_EPI_STORE = 0,    /* out[T]   = act(acc + bias)                                   */
    LMI_EPI_RESIDUAL = 1, /* out[f32] += acc + bias            (residual stream update)   */
    LMI_EPI_STORE_F32 = 2,/* out[f32] = acc + bias + addmat[row % add_period]             */
    LMI_EPI_SWIGLU = 3,   /* out[T][:, N/2] = silu(gate) * up, W rows interleaved [32 gate | 32 up] */
    LMI_EPI_QKV_ROPE = 4, /* q | k | v projection + RoPE + KV-cache append (lmi_rmsnorm_rope only) */
    LMI_EPI_SWIGLU_F32 = 5/* LMI_EPI_SWIGLU with an fp32 destination [., N/2] (split-operand precision mode) */
};
enum { LMI_ACT_NONE = 0, LMI_ACT_GELU_TANH = 1, LMI_ACT_GELU_ERF = 2 };
enum { LMI_A_PLAIN = 0, LMI_A_PIXEL_SHUFFLE = 1 };

const char* lmi_last_error(void);
int lmi_abi_version(void);

/* Tuning knobs (process-global; experiments and A/B runs only, the defaults are the measured best):
 *   "gemm.config"   -1 = choose the GEMM geometry / schedule per shape (default); 0..9 = force one of those listed in
 *                   csrc/capi.hip (tools/bench_kernels.py)
 *   "gemm.wide" / "gemm.short_k" / "gemm.narrow_n" / "gemm.small"   geometry 0..7 per shape class (csrc/capi.hip
 *                   choose_gemm_cfg: N >= 2048 with K > 1536 / K <= 1536; N < 2048 with K >= 2048 / K < 2048)
 *   "gemm.group_m"  row-tiles per group of the XCD-aware tile order (default 4)
 *   "gemm.order"    0 = each XCD owns a contiguous slab of the tile order (default), 1 = round-robin 32-tile patches
 *   "attn.dma"      1 = LDS-DMA attention kernel (default), 0 = register-staged cross-check kernel
 *   "attn.lds_pad"  extra dynamic LDS per attention workgroup in bytes (lowers residency; default 0)
 *   "attn.gqa_pack" 1 = decode blocks take the 4 query heads of one kv head (default), 0 = one block per query head
 *   "attn.decode_split_tiles"  64-key tiles per split-KV decode workgroup: 0 = chosen from the launch shape (default), 1 / 2 / 4 / 8
 *   "attn.stream_kv"  1 = non-temporal K / V tile loads in GQA-packed decode blocks (default), 0 = default cache policy
 *   "gemv.plan"     1 = decode GEMV grid of a whole number of equal workgroups per CU when the shape allows (default), 0 = ~1000 workgroups
 *   "gemm.mid_m" / "gemm.auto_small"  M-complete 384 x 128 tiles for 256 < M <= 384 / small-shape geometries (default 1)
 *   "attn.rows64"   0 = the production attention kernel for every launch (default); 1 / 2 = long head_dim-128 self-attention prefills take the
 *                   software-pipelined kernels of csrc/attention64.h (64 rows per wave / 32 rows per wave, two waves per SIMD) — verified, slower
 *   "attn.rows64_min"  shortest max_seqlen_q that takes them (default 1024)
 *   "skinny.coalesce"  lmi_gemm_skinny* on nn.Linear-layout weights: 1 = coalescing lane order + ds_bpermute (default), 0 = MFMA lane order
 * Unknown keys and out-of-range values return LMI_EINVAL. */
int lmi_set_option(const char* key, int value);

/* Diagnostics (tools/overlap_probe.py): 16-byte grid-stride copy on exactly n_workgroups workgroups of 256 threads, no LDS — a
 * stand-in for a collective's transport kernel when measuring what runs beside the GEMMs.  Not used by the product path. */
int lmi_debug_copy(const void* src, void* dst, int64_t bytes, int n_workgroups, void* stream);

/* Deterministic synthetic parameters (no checkpoints exist offline): element i = f(seed, i, kind); bit-identical
 * to leopard_amd/synth.py.  out_dtype in {LMI_F16, LMI_BF16, LMI_F32}. */
int lmi_fill_synthetic(void* out, int64_t n, uint32_t seed, int kind, int out_dtype, void* stream);

/* a5 — replaces SiglipImageProcessor.preprocess + the patch gather of the SigLIP patch conv
 * (EVAL:403-405; third-party modeling_siglip patch_embedding; analogue
 * megatron_patch/model/idefics2/idefics_vision_tower.py:57-64).
 * in: u8 tiles [n_tiles, S, S, 3] (from_u8=1; applies x/255 then (x-0.5)/0.5) or normalised fp32 pixel_values
 * [n_tiles, 3, S, S] (from_u8=0).  out: T [n_tiles*(S/P)^2, ldo], column c*P*P+ky*P+kx, zero padded to ldo. */
int lmi_preprocess_tiles(const void* in, int from_u8, void* out, int n_tiles, int image_size, int patch,
                         int ldo, int dtype, void* stream);

/* a3 / a5 (next row f3: the tiler on the GPU) — one pass of Pillow's antialiased 8-bit RGB resampling, the arithmetic
 * behind `image.resize(...)` in resize_and_pad_image (EVAL:102-140) and behind SiglipImageProcessor's bicubic resize of
 * the thumbnail (EVAL:403-404).  Bit-identical to libImaging for the taps it is given: bounds[o] = {first source index,
 * tap count}, taps[o][ksize] = 22-bit fixed-point weights (leopard_amd/tiler.py pil_resample_coeffs).
 * axis 0: src [out_rows, in, 3] -> dst [out_rows, out_cols, 3] (along rows); axis 1: src [in, out_cols, 3] ->
 * dst [out_rows, out_cols, 3] (down columns).  Pitches in bytes, so dst may be a window of a larger canvas (the paste). */
int lmi_resample_u8(const void* src, void* dst, int axis, int out_rows, int out_cols, int src_pitch, int dst_pitch,
                    const int* bounds, const int* taps, int ksize, void* stream);

/* Same for rectangular images (Leopard-Idefics2's NaViT tower, idefics2_multiimg.py:91-93; third-party
 * Idefics2VisionEmbeddings; analogue idefics_vision_tower.py:118-150): n_images of height x width (u8 HWC or normalised
 * fp32 CHW); the valid patch conv drops remainder pixels -> (height/patch)*(width/patch) rows per image. */
int lmi_preprocess_images(const void* in, int from_u8, void* out, int n_images, int height, int width, int patch, int ldo,
                          int dtype, void* stream);

/* LayerNorm (SigLIP layer_norm1/2, post_layernorm; analogue idefics_vision_tower.py:77-81,176) and
 * RMSNorm (Llama input/post_attention/final norm; megatron/legacy/model/rms_norm.py:26-31):
 * x fp32 [M, ldx] -> out T [M, ldo]; w, b fp32 [D].  dtype LMI_F32 writes the normalised rows in fp32 (Idefics2 perceiver
 * output, which is merged into the fp32 stream without a 16-bit rounding). */
int lmi_layernorm(const float* x, const float* w, const float* b, void* out, int M, int D, int ldx, int ldo,
                  float eps, int dtype, void* stream);
int lmi_rmsnorm(const float* x, const float* w, void* out, int M, int D, int ldx, int ldo, float eps, int dtype,
                void* stream);

/* Residual update + RMSNorm on the rows a rank owns (sequence-parallel tensor parallelism; the exchange pattern of
 * Megatron-LM-240603/megatron/core/tensor_parallel/mappings.py:107-145): x[M, ldx] fp32 += delta[M, ldd] (the reduce-scattered
 * partial products, delta_dtype = LMI_F32 or dtype), then out[M, ldo] T = w * (x * rsqrt(mean(x^2) + eps)).  out null: add only. */
int lmi_add_rmsnorm(float* x, const void* delta, int delta_dtype, const float* w, void* out, int M, int D, int ldx, int ldd, int ldo,
                    float eps, int dtype, void* stream);

/* out = epilogue(A[M,K] . W[N,K]^T): every nn.Linear / conv-as-GEMM on the path —
 * SigLIP patch-embed, q/k/v/out_proj, fc1 (+gelu_tanh), fc2; projector linear_1 (+gelu_erf, A gathered through
 * the 2x2 pixel shuffle of EVAL:165-176) and linear_2 (EVAL:187-192); Llama qkv / o_proj / gate+up (+SwiGLU,
 * megatron_patch/model/llava/transformer.py:136-139) / down_proj; all-position lm_head (EVAL:333).
 * Requires N % 128 == 0, K % 64 == 0.  A, W are T; bias/addmat fp32 (nullable); addmat row = add_rows[m] when add_rows is
 * given (NaViT bucketised position ids), else m % add_period; row_map (nullable) scatters
 * output row m to row row_map[m].  For LMI_A_PIXEL_SHUFFLE, A is the ViT output [tiles*G*G, K/4] and M counts
 * shuffled rows (tiles*(G/2)^2).
 *
 * W layouts (lmi_gemm, lmi_gemm_bias_act, lmi_gemm_ex, lmi_rmsnorm_rope).  ldw > 0: the nn.Linear layout, row n at W + n * ldw.
 * ldw = LMI_LDW_PACKED(K): W is stored in the operand order the decode kernels stream (lmi_gemm_skinny, packed = 1): 1-KiB blocks
 * [16-row group][k-step of 128][32-k chunk][lane 16 g + i][8 elements] = W[16 r + i][128 s + 32 c + 8 g + j]; needs K % 128 == 0,
 * N % 16 == 0, plain A.  That order permutes the 16-byte pieces of the row-major matrix inside (16 rows x 64 k) tiles, and the GEMM's
 * LDS-DMA lanes pick their source piece: the LDS image, the MFMA order and every output bit are those of the row-major call.  ONE
 * copy of the LLM weights then serves prefill and decode (leopard_amd.weights.skinny_pack / LeopardEngine.pack_llm_weights). */
#define LMI_LDW_PACKED(K) (-(K))
int lmi_gemm(const void* A, const void* W, void* out, const float* bias, const float* addmat, const int* add_rows,
             const int* row_map, int M, int N, int K, int lda, int ldw, int ldo, int add_period, int epilogue, int act, int a_mode,
             int ps_grid, int dtype, void* stream);

/* SURVEY.md 8(b) names the linear "lmi_gemm_bias_act": the same GEMM behind the short argument list of that row —
 * act in {LMI_ACT_NONE, LMI_ACT_GELU_TANH, LMI_ACT_GELU_ERF, LMI_ACT_SWIGLU}, residual != 0: out[f32] += result (else out[T] = result),
 * ps_grid > 0: A gathered through the 2x2 pixel shuffle of a ps_grid x ps_grid token grid (EVAL:165-176). */
enum { LMI_ACT_SWIGLU = 3 /* lmi_gemm_bias_act only: W rows interleaved [32 gate | 32 up], out width N/2 */ };
int lmi_gemm_bias_act(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int act,
                      int residual, int ps_grid, int dtype, void* stream);

/* SURVEY.md 8(b) "lmi_patch_embed" — SiglipVisionEmbeddings (patch convolution 3 -> N with kernel = stride = patch, bias, + position
 * embedding; inside self.vision_tower(...), EVAL:268) fused with SiglipImageProcessor's rescale / normalise (EVAL:403-405) as ONE
 * im2col + MFMA GEMM: each k-tile stages a slice of the patches' pixel rows from the image into LDS (normalised with the
 * processor's exact arithmetic, rounded to T) — no im2col matrix in HBM.  pixels: u8 [n_tiles, S, S, 3] (from_u8 = 1; the GPU
 * tiler's output) or fp32 [n_tiles, 3, S, S] pixel_values (from_u8 = 0): both give bit-identical results.  W: T [N, ldw] with
 * K in IMAGE order, k = ky * RP + kx * 3 + c, RP = roundup(3 * patch, 8), zero in the pad positions, ldw >= roundup(patch * RP, 64)
 * (leopard_amd/weights.py: patch_w_fused); pos_emb fp32 [(S/patch)^2, N]; out fp32 [n_tiles * (S/patch)^2, ldo].  N % 128 == 0. */
int lmi_patch_embed(const void* pixels, int from_u8, const void* W, const float* bias, const float* pos_emb, float* out, int n_tiles,
                    int image_size, int patch, int N, int ldw, int ldo, int dtype, void* stream);

/* SURVEY.md 8(b) "lmi_kv_append" — the cache update of the decode branch (EVAL:291-320) for rows that are already rotated:
 * k_cache / v_cache rows [cache_pos0, cache_pos0 + S) = k / v rows [0, S) (T, `width` = n_kv_heads * head_dim elements).  Used to
 * move a packed batch's K/V from the pooled prefill cache into per-sample caches (LeopardEngine.generate_batch). */
int lmi_kv_append(const void* k, const void* v, void* k_cache, void* v_cache, int S, int width, int ld_src, int ld_cache, int cache_pos0,
                  int dtype, void* stream);

/* RMSNorm folded into the GEMMs around it (north_star "fused RMSNorm + RoPE"; RMSNorm = megatron/legacy/model/rms_norm.py:26-31,
 * call sites megatron_patch/model/llava/transformer.py:1208-1340).  lmi_gemm restricted to plain A, plus:
 *   producer (epilogue LMI_EPI_RESIDUAL, norm_out != null): after x += acc + bias it also writes
 *       norm_out[m, n]    = T(x[m, n] * norm_gamma[n])            (the next RMSNorm's gain applied, its row scale still missing)
 *       rowsq_out[m, n/64] = sum of x[m, 64*(n/64) .. +63]^2       (N/64 partials per row; no atomics: bit-reproducible)
 *   consumer (any epilogue, rowsq_in != null): accumulator row m is multiplied by
 *       rstd[m] = rsqrt(sum_j rowsq_in[m, j] / norm_dim + norm_eps)  before bias / activation / SwiGLU,
 *   which completes the norm:  (x * gamma) . W^T * rstd == (gamma * x * rstd) . W^T.  One 16-bit rounding of the operand, as
 *   with lmi_rmsnorm, but no norm launch and no separate pass over the fp32 stream. */
int lmi_gemm_ex(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, const float* norm_gamma, float* rowsq_out,
                int ld_norm, int dtype, void* stream);

/* SURVEY.md 8(b) "lmi_rmsnorm_rope" — the q | k | v projection of a Llama / Mistral layer with the RMSNorm's row scale, the
 * rotary embedding and the KV-cache append all in the GEMM's epilogue (RoPE: rotary_pos_embedding.py:197-239 rotate-half;
 * call site megatron_patch/model/llava/transformer.py:838-856):
 *     qkv[m] = [ RoPE(q), RoPE(k), v ]   with  (q | k | v) = rstd[m] * (A[m] . Wqkv^T),
 * rstd from rowsq_in as in lmi_gemm_ex (rowsq_in null: A is already normalised, e.g. by lmi_rmsnorm for the first layer).
 * The rotation acts on the fp32 accumulators: q and k are rounded to T once, not twice.
 * Wqkv: [(n_q + 2 n_kv) * 128, K]; the 128 rows of every q and k head must be stored in the order
 * d = 0..31, 64..95, 32..63, 96..127 (leopard_amd.weights.rope_permute_rows), so that the 64 output columns a wave owns hold 32
 * first-half elements and their rotate-half partners; the epilogue restores the natural order on store.  cos/sin: fp32
 * [M, 64] per packed row.  k_cache / v_cache (nullable): rotated K and V also go to cache rows cache_pos0 + m.  head_dim = 128. */
int lmi_rmsnorm_rope(const void* A, const void* Wqkv, void* qkv, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_table,
                     const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads,
                     int head_dim, int K, int lda, int ldw, int ldo, int dtype, void* stream);

/* fp8 MFMA linears (BASELINE config 5: "fp8 MFMA ViT+LLM prefill").  Operands are OCP fp8 e4m3fn bytes (gfx950's format), [M, K]
 * and [N, K] row-major, K % 128 == 0; the product runs on v_mfma_scale_f32_32x32x64_f8f6f4 (the fp8 path above the 16-bit MFMA
 * rate on gfx950) with fp32 accumulation and comes out multiplied by 2^scale_exp — the inverse of the power-of-two scales the two
 * operands were quantised with.  Epilogues as lmi_gemm (STORE [+GELU-tanh], RESIDUAL, STORE_F32, SWIGLU); out_dtype (LMI_F16 /
 * LMI_BF16) is the type of the 16-bit outputs.  lmi_quantize_fp8 is the hand-over of an activation (fp32 / 16-bit [M, D]) to such
 * an operand: out = fp8(x * scale), round to nearest even, saturating at +-448. */
int lmi_quantize_fp8(const void* x, int x_dtype, void* out, int M, int D, int ldx, int ldo, float scale, void* stream);
/* out_dtype LMI_FP8 (STORE [+GELU-tanh] and SWIGLU only): the results are written as fp8(value * out_scale), ready to be the A operand
 * of the next fp8 GEMM (SigLIP fc1 -> fc2, Llama gate/up -> down) without a separate quantisation pass. */
int lmi_gemm_fp8(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                 int scale_exp, int out_dtype, float out_scale, void* stream);
/* fp8 schedule: q | k | v projection on e4m3 operands with RoPE and the KV-cache append in the epilogue (the fp8 counterpart of
 * lmi_rmsnorm_rope; A8 is the already normalised operand lmi_norm_fp8 wrote, Wqkv8 the fp8 weight in weights.rope_permute_rows order).
 * qkv / caches are 16-bit (out_dtype LMI_F16 | LMI_BF16); accumulators are multiplied by 2^scale_exp first. */
int lmi_rope_qkv_fp8(const void* A8, const void* Wqkv8, void* qkv, int scale_exp, const float* cos_table, const float* sin_table, void* k_cache,
                     void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int lda, int ldw, int ldo,
                     int out_dtype, void* stream);

/* fp8 schedule: the attention ARITHMETIC of the Llama / Mistral layers on the fp8 matrix pipe too (BASELINE configs[4] "fp8 MFMA ViT+LLM
 * prefill": QK^T and PV on v_mfma_scale_f32_32x32x64_f8f6f4; the layers' self-attention, megatron_patch/model/llava/transformer.py:678-885).
 * Causal or full self-attention over packed sequences, head_dim 128, GQA.  Two launches per layer:
 *   lmi_attn_prep_fp8  q | k | v rows (16-bit, rotated; [rows, ld]: q heads | k heads | v heads) -> q8 rows [rows, ldq8] = e4m3(q * q_scale) and,
 *                      per kv head and 64-key tile, an 8-KiB K image and an 8-KiB transposed V image (e4m3(k * k_scale), e4m3(v * v_scale);
 *                      k_img / v_img: n_kv_heads * n_tiles * 8192 bytes each).  tile_base (device, [n_seq + 1]): tile_base[s] = the first tile of
 *                      sequence s, tile_base[s + 1] - tile_base[s] = ceil(len_s / 64), tile_base[n_seq] = n_tiles.
 *   lmi_attn_fp8_fwd   softmax(q k^T * softmax_scale) v from those operands; P is rounded to e4m3, sums and O stay fp32.  Output either T rows
 *                      (out, ldo) or, out_fp8 != null, e4m3(O * out_fp8_scale) — the o_proj operand of the fp8 schedule. */
int lmi_attn_prep_fp8(const void* qkv, int ld, const int* cu_seqlens, const int* tile_base, int n_seq, int n_tiles, int n_q_heads, int n_kv_heads,
                      int head_dim, float q_scale, float k_scale, float v_scale, void* q8, int ldq8, void* k_img, void* v_img, int dtype,
                      void* stream);
int lmi_attn_fp8_fwd(const void* q8, int ldq8, const void* k_img, const void* v_img, void* out, int ldo, void* out_fp8, int ldo8, float out_fp8_scale,
                     const int* cu_seqlens, const int* tile_base, int n_seq, int n_tiles, int max_seqlen, int n_heads, int n_kv_heads, int head_dim,
                     float softmax_scale, float q_scale, float k_scale, float v_scale, int causal, int dtype, void* stream);

/* LayerNorm (b != null) / RMSNorm (b == null) of the fp32 stream written straight as an fp8 GEMM operand: out = fp8(norm(x) * out_scale). */
int lmi_norm_fp8(const float* x, const float* w, const float* b, void* out, int M, int D, int ldx, int ldo, float eps, float out_scale,
                 void* stream);

/* Variable-length FlashAttention-2 forward over packed sequences (SigLIP: non-causal, head_dim 72, one
 * sequence per tile; Llama / Mistral: causal GQA, head_dim 128, optional sliding window; Idefics2 perceiver: head_dim 96,
 * len_q != len_k) — replaces the attention inside self.vision_tower(...)
 * and self.language_model(...) (EVAL:268,322; analogue transformer.py:456-512 flash_attn_varlen_func).
 * q/k/v/out: T, head h of row r at base + r*ld + h*head_dim.  cu_seqlens: int32 [n_seq+1] on device.
 * Causal alignment is bottom-right (key j visible to query i iff j <= i + len_k - len_q); window > 0 additionally hides
 * keys with i + len_k - len_q - j >= window (Mistral sliding window), 0 = unlimited.
 * use_tr=1 uses ds_read_b64_tr_b16 for the V operand (production); 0 uses plain LDS gathers (cross-check).
 * Limit: the K (and V) rows of ONE sequence must span less than 4 GiB (32-bit buffer offsets); the kernel traps beyond. */
int lmi_attn_varlen_fwd(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q,
                        const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads,
                        int head_dim, int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window,
                        int use_tr, int dtype, void* stream);

/* The same attention with the output written as the NEXT GEMM's fp8 operand: out_fp8[row, h * head_dim + d] = e4m3(O * out_scale)
 * (uint8, row stride ldo8), straight from the fp32 accumulators — the fp8 schedule's o_proj operand without a 16-bit round trip and a
 * conversion launch (BASELINE configs[4]).  LDS-DMA kernel only (head_dim 72 / 96 / 128). */
int lmi_attn_varlen_fwd_fp8(const void* q, const void* k, const void* v, void* out_fp8, int ldo8, float out_scale, const int* cu_seqlens_q,
                            const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, float scale, int causal, int window, int dtype, void* stream);

/* Split-operand precision mode (LeopardEngine.split_operands).  The HIP path's distance from the fp32 reference is one
 * rounding to the 16-bit type per hand-over of an activation to an MFMA operand.  lmi_split_hi_lo writes an fp32 activation [M, K] as the
 * 16-bit pair [M, 2K] = [T(x) | T(x - T(x))]; a GEMM over it against the weight laid out twice, [W | W], computes hi.W + lo.W — the
 * product of the unrounded activation — at twice the K.  The producers hand over fp32: the norms (LMI_F32 outputs), the attention
 * (lmi_attn_varlen_fwd_f32: normalised output in fp32, LDS-DMA kernel), fc1 (LMI_EPI_STORE_F32 + GELU) and gate/up (LMI_EPI_SWIGLU_F32). */
int lmi_split_hi_lo(const float* x, void* out, int M, int K, int ldx, int ldo, int dtype, void* stream);
int lmi_attn_varlen_fwd_f32(const void* q, const void* k, const void* v, float* out_f32, int ldo32, const int* cu_seqlens_q,
                            const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, float scale, int causal, int window, int dtype, void* stream);

/* Low-bit correction phase (LeopardEngine.precision = "lo4") — the cheap way to north_star's "logits within 1e-3".
 * The reference runs fp32 (evaluations/models/llava_multiimg_siglip_anyres.py:373,322-333); a 16-bit-operand MFMA path differs from it by
 * ONE rounding of every activation handed to a matrix operand.  That residual x - T(x) only needs a few bits: producers hand it over as
 * an MX fp4 image (e2m1; one E8M0 power-of-two scale per 32 consecutive k), the weight has an fp4 image with one E8M0 scale per row, and
 * after its 16-bit k-loop the GEMM runs K4 / 256 more k-tiles of v_mfma_scale_f32_32x32x64_f8f6f4 on the two images INTO THE SAME
 * ACCUMULATORS: + 25 % matrix-pipe time instead of + 100 % for the hi + lo 16-bit pair, ~80 % of the rounding removed.
 * Images: row-major, element k of a row in nibble k & 1 of byte k >> 1; K4 = K rounded up to a multiple of 256 elements (or wider, when
 * the two images share a padded k order: lmi_attn_varlen_fwd_lo4), zero codes and zero scale bytes in the padding; lda4 / ldw4 / ld_out4 in BYTES; a4_scale [M, lds4] with lds4 >= K4 / 32; w4_scale [N].
 * lmi_lo4: inputs (a4, a4_scale, w4, w4_scale: all four) consumed by this launch, outputs (out4, out4_scale: both or neither) = the image of
 * the residual of this launch's own 16-bit result — `out` for LMI_EPI_STORE (+ activation) and LMI_EPI_SWIGLU, `norm_out` for the
 * RESIDUAL producer mode — for the next GEMM.
 * Row selection (round 6): the last-position logits of a sequence are dominated by the hand-over roundings of THAT row's own path through the
 * layers — the other rows' roundings reach it only through the softmax average over ~S keys — so the correction need only cover the rows whose
 * logits are read.  row_sel [M] bytes: != 0 <=> row m's operands carry a residual image (producers write the images of selected rows only: the
 * image buffers must be zero-filled once before the pass, an unselected row then reads as zero codes and its results are bit for bit the fast
 * schedule's wherever it sits in a tile); unit_sel [ceil(M / 64)] bytes: OR of row_sel over rows [64 u, 64 u + 64) — a tile none of whose units
 * is set skips the fp4 k-tiles.  Both null (or both given): null = every row.
 * sel_ranges (optional, HOST memory, read during the call): n_sel_ranges pairs [begin, end) of rows, ascending and disjoint, covering the selected
 * rows — a hint for the TILE ORDER only: the row tiles that hold selected rows are dispatched first and spread evenly over the XCDs (workgroups
 * are dispatched in order and the XCDs advance in lock-step: a few 1.25 x longer tiles inside one XCD's slab would slow the whole launch). */
typedef struct lmi_lo4 {
    const void* a4; const void* a4_scale; const void* w4; const void* w4_scale;
    int lda4, ldw4, lds4, k4;
    void* out4; void* out4_scale;
    int ld_out4, ld_out4s;
    const void* row_sel; const void* unit_sel;
    const int* sel_ranges; int n_sel_ranges;
} lmi_lo4;
/* lmi_gemm_ex / lmi_rmsnorm_rope with the correction phase (plain A, row-major or packed W for the 16-bit pass, K >= 128). */
int lmi_gemm_lo4(const void* A, const void* W, void* out, const float* bias, int M, int N, int K, int lda, int ldw, int ldo, int epilogue, int act,
                 const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, const float* norm_gamma, float* rowsq_out,
                 int ld_norm, const lmi_lo4* lo, int dtype, void* stream);
int lmi_rmsnorm_rope_lo4(const void* A, const void* Wqkv, void* qkv, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_table,
                         const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int M, int n_q_heads, int n_kv_heads,
                         int head_dim, int K, int lda, int ldw, int ldo, const lmi_lo4* lo, int dtype, void* stream);
/* lmi_attn_varlen_fwd that also writes the residual image of its 16-bit output rows (LDS-DMA kernel; head_dim 72 / 96 / 128).  The image
 * has its own k order so that no 32-element block straddles two heads: head h occupies blocks [h * NB, (h + 1) * NB), NB = ceil(head_dim / 32)
 * (head_dim 128: the natural order; 72 / 96: every head padded to 96 with zero codes), i.e. k4 = n_heads * NB * 32 rounded up to 256 by the
 * caller's buffer; the consuming projection's weight image must use the same order.  out4 [total_q, ld_out4 bytes], out4_scale [total_q, ld_out4s]. */
int lmi_attn_varlen_fwd_lo4(const void* q, const void* k, const void* v, void* out, void* out4, void* out4_scale, int ld_out4, int ld_out4s,
                            const int* cu_seqlens_q, const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, int dtype, void* stream);
/* the same with a row selection (lmi_lo4.row_sel: [total_q] bytes, null = every row): unselected rows get their 16-bit row only */
int lmi_attn_varlen_fwd_lo4_rows(const void* q, const void* k, const void* v, void* out, void* out4, void* out4_scale, int ld_out4, int ld_out4s,
                                 const int* cu_seqlens_q, const int* cu_seqlens_k, int n_seq, int max_seqlen_q, int n_heads, int n_kv_heads, int head_dim,
                                 int ldq, int ldk, int ldv, int ldo, float scale, int causal, int window, const void* row_sel, int dtype, void* stream);
/* fp32 activation [M, K] (K % 32 == 0) -> hi = T(x) [M, ldh] + the fp4 image of x - T(x) [M, ld4 bytes] + its block scales [M, lds]
 * (attention outputs: lmi_attn_varlen_fwd_f32 hands over fp32). */
int lmi_split_lo4(const float* x, void* hi, void* lo4, void* scales, int M, int K, int K4, int ldx, int ldh, int ld4, int lds, int dtype, void* stream);
/* LayerNorm (b != null) / RMSNorm (b == null) writing T(y) and the fp4 image of y - T(y) in one launch (D % 32 == 0, D <= 4096). */
int lmi_norm_lo4(const float* x, const float* w, const float* b, void* out, void* out4, void* scales, int M, int D, int K4, int ldx, int ldo,
                 int ld4, int lds, float eps, int dtype, void* stream);
/* the same with a row selection (lmi_lo4.row_sel: [M] bytes, null = every row): unselected rows get T(y) only */
int lmi_norm_lo4_rows(const float* x, const float* w, const float* b, void* out, void* out4, void* scales, int M, int D, int K4, int ldx, int ldo,
                      int ld4, int lds, float eps, const void* row_sel, int dtype, void* stream);
/* lmi_add_rmsnorm (tensor parallel: residual add of the reduce-scattered partial products + RMSNorm on the rank's rows) writing the Lo4 pair. */
int lmi_add_rmsnorm_lo4(float* x, const void* delta, int delta_dtype, const float* w, void* out, void* out4, void* scales, int M, int D, int K4,
                        int ldx, int ldd, int ldo, int ld4, int lds, float eps, int dtype, void* stream);
/* Weight image, once at load: W T [N, K] row-major (ldw elements) -> fp4 [N, ld4 bytes] + one E8M0 scale per row [N]. */
int lmi_quantize_w4(const void* W, void* w4, void* scales, int N, int K, int K4, int ldw, int ld4, int dtype, void* stream);

/* Caller-owned scratch of a prefill pass (SURVEY.md 8b "workspace: size from lmi_*_workspace_bytes"; the library allocates nothing).
 * One contiguous, 256-byte aligned workspace per stage holds every activation buffer that lives between the launches of one pass — the
 * Llama / Mistral layer stack over `rows` packed sequence rows (reference: LlamaForCausalLM.forward, EVAL:322-333), the SigLIP layer
 * stack over `rows` = n_vit_inputs x tokens (EVAL:268-273).  Returns the total bytes (-1: bad argument, lmi_last_error); `offsets`
 * (nullable) receives the byte offset of buffer LMI_WS_* inside the workspace.  The fp32 residual streams, the KV cache and the logits
 * are results, not scratch: they stay separate caller buffers. */
enum { LMI_WS_LLM_H = 0, LMI_WS_LLM_QKV = 1, LMI_WS_LLM_ATT = 2, LMI_WS_LLM_GU = 3, LMI_WS_LLM_SQ_A = 4, LMI_WS_LLM_SQ_B = 5, LMI_WS_LLM_COUNT = 6 };
enum { LMI_WS_VIT_H = 0, LMI_WS_VIT_QKV = 1, LMI_WS_VIT_ATT = 2, LMI_WS_VIT_FF = 3, LMI_WS_VIT_COUNT = 4 };
int64_t lmi_llm_prefill_workspace_bytes(int64_t rows, int hidden, int n_q_heads, int n_kv_heads, int head_dim, int ff, int dtype, int64_t* offsets);
int64_t lmi_vit_workspace_bytes(int64_t rows, int hidden, int qkv_width, int ff_padded, int dtype, int64_t* offsets);

/* a12 (next row f2: the decode loop) — the same attention for a few query rows against a long KV cache: the key range is
 * split over workgroups (512 keys each, at most 64 splits), partial (O, max, sum) go to `workspace` and are merged.
 * Replaces the attention inside the decode branch of the reference forward (EVAL:291-333).  Causal (bottom-right), GQA,
 * head_dim 128.  q_rows = rows of q / out (>= cu_seqlens_q[n_seq]); max_seqlen_k is a host upper bound that fixes the
 * launch geometry (pass the cache capacity to keep it constant under a captured HIP graph); cu_seqlens on device.
 * workspace: lmi_attn_decode_workspace_bytes(q_rows, n_heads, head_dim, max_seqlen_k) bytes, 16-byte aligned. */
int64_t lmi_attn_decode_workspace_bytes(int q_rows, int n_heads, int head_dim, int max_seqlen_k);
int lmi_attn_decode_fwd(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* cu_seqlens_k,
                        int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                        int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                        int dtype, void* stream);

/* The same for a BATCH of decode sequences whose KV caches share one pooled buffer (several samples per GPU;
 * the reference loop EVAL:448-454 is batch 1): sequence s owns cache rows [k_begin[s], k_begin[s] + k_len[s]); k_begin is static
 * (slot * capacity), k_len advances on the device, so the launch can sit in a captured HIP graph. */
int lmi_attn_decode_pool(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* k_begin, const int* k_len,
                         int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                         int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                         int dtype, void* stream);

/* The tail of a greedy decode step (EVAL:448-452: argmax, stop at eos / max_new_tokens) for B sequences, in device memory only (it sits
 * inside the captured step): tok[b] = argmax of logits row b over [0, vocab) (lowest index on ties; ids in `suppress` excluded), recorded
 * in hist[hist_pos[b] % hist_len][b] (hist_pos[b] += 1); budget[b] -= live[b]; a sequence whose token is one of `eos` (entries < 0 unused)
 * or whose budget reached 0 gets live[b] = 0; pos[b] += live[b], k_len[b] += live[b].  live / budget / hist / k_len / suppress nullable
 * (null live = every sequence runs on; the batch-1 step passes its one position and key count). */
int lmi_decode_advance(const float* logits, int B, int vocab, int ld_logits, const int64_t* suppress, int n_suppress, int64_t* tok, int* pos,
                       int* k_len, int* live, int* budget, const int64_t* eos, int n_eos, int64_t* hist, int* hist_pos, int hist_len, void* stream);

/* RoPE (rotate-half; cos/sin fp32 [S, head_dim/2] built from position_ids and the llama3-scaled inverse
 * frequencies, rotary_pos_embedding.py:48-83,197-239) applied in place to the q and k heads of packed qkv rows
 * [S, ld]; when k_cache/v_cache are non-null also appends rotated K and V to the cache rows cache_pos0.. */
int lmi_rope_qk(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_table,
                const float* sin_table, void* k_cache, void* v_cache, int ld_cache, int cache_pos0, int dtype,
                void* stream);

/* Same with the position of row 0 read from device memory (*pos_dev): cos_all/sin_all are tables for every position of the
 * cache ([capacity, head_dim/2]) and the K/V rows go to cache row *pos_dev + s.  Lets a decode step be captured in a HIP graph
 * and replayed while a device counter advances (EVAL:291-320: position_ids = mask.sum - 1). */
int lmi_rope_qk_at(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_all,
                   const float* sin_all, void* k_cache, void* v_cache, int ld_cache, const int* pos_dev, int dtype, void* stream);

/* Batched decode: row s is sequence s's next token at position pos_rows_dev[s]; its rotated K and its V go to row
 * s * cache_stride + pos_rows_dev[s] of the pooled caches (EVAL:291-320 per sequence). */
int lmi_rope_qk_rows(void* qkv, int S, int ld, int n_q_heads, int n_kv_heads, int head_dim, const float* cos_all, const float* sin_all,
                     void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream);

/* a10 — get_input_embeddings()(input_ids) + _merge_input_ids_with_image_features (EVAL:263,284-287; analogue
 * megatron_patch/model/llava/vlm_model.py:526-533): out fp32 [S, D]; src[s] >= 0 -> embed_table[ids[src[s]]],
 * src[s] < 0 -> visual_tokens[-src[s]-1] (fp32 [., ld_feats]).  ids, src: int64 on device. */
int lmi_embed_merge(const int64_t* ids, const int64_t* src, const void* embed_table, const float* visual_tokens,
                    float* out, int S, int D, int ld_feats, int dtype, void* stream);

/* M = 1 weight-streaming GEMV: last-token lm_head (EVAL:333 restricted to the position generate() consumes)
 * and the decode step (EVAL:291-320).  epilogue: 0 store fp32, 1 store T, 2 fp32 +=, 3 SwiGLU (N/2 outputs). */
int lmi_gemv(const void* W, const void* x, const float* bias, void* out, int N, int K, int ldw, int epilogue,
             int dtype, void* stream);

/* M <= 16 weight-streaming GEMM: the projections of a batched decode step (the weight stream of one token serves M tokens).
 * out[M, .] = epilogue(X[M, K] . W[N, K]^T), X and W 16-bit, K % 128 == 0.  epilogue: LMI_SKINNY_STORE (T out [M, N]),
 * LMI_SKINNY_RESIDUAL (fp32 out += ), LMI_SKINNY_SWIGLU (W rows interleaved [32 gate | 32 up]; T out [M, N/2]),
 * LMI_SKINNY_STORE_F32.  packed = 1: W is the same matrix pre-arranged in the MFMA operand order (1-KiB blocks [16-row group][k-step of 128]
 * [32-k chunk][lane][8 elements]; leopard_amd.weights.skinny_pack), ldw == K: every wave load is one coalesced 1-KiB request instead
 * of 64 scattered 16-byte pieces. */
#define LMI_SKINNY_STORE 0
#define LMI_SKINNY_RESIDUAL 1
#define LMI_SKINNY_SWIGLU 2
#define LMI_SKINNY_STORE_F32 3
int lmi_gemm_skinny(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed, int dtype,
                    void* stream);

/* lmi_gemm_skinny with the RMSNorms of the batched decode step folded into its projections — the M <= 16 counterpart of lmi_gemm_ex:
 *   producer (LMI_SKINNY_RESIDUAL, norm_out != null): after out += acc it also writes norm_out[m, n] = T(out[m, n] * norm_gamma[n]) (T [M, ld_norm])
 *       and rowsq_out[m, n / 16] = the sum of out[m, 16 (n / 16) .. + 15]^2 (fp32 [M, N / 16]; one partial per workgroup: bit-reproducible);
 *   consumer (any epilogue, rowsq_in != null, fp32 [M, rowsq_parts]): accumulator row m is multiplied by
 *       rstd[m] = rsqrt(sum_j rowsq_in[m, j] / norm_dim + norm_eps) before SwiGLU / store.
 * The decode step then needs no norm launch after its first layer's. */
int lmi_gemm_skinny_ex(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed,
                       const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, int ld_norm, const float* norm_gamma,
                       float* rowsq_out, int dtype, void* stream);

/* Batched decode: the q | k | v projection as lmi_gemm_skinny with RoPE and the KV append in its epilogue (W rows in
 * weights.rope_permute_rows order, optionally packed): row m rotates at position pos_rows_dev[m] and appends K / V to row
 * m * cache_stride + pos_rows_dev[m] of the pooled caches — lmi_gemm_skinny + lmi_rope_qk_rows in one launch.  rowsq_in != null
 * (fp32 [M, rowsq_parts], written by the lmi_gemm_skinny_ex producer over the K-wide residual row): X is the un-normalised T(x * gamma)
 * and every row is scaled by rstd before the rotation (consumer side of the folded RMSNorm). */
int lmi_rope_qkv_skinny(const void* Wqkv_rope, const void* X, void* qkv, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int ldw, int ldx,
                        int ldo, int packed, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_all, const float* sin_all,
                        void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream);

/* Decode precision mode (round 6; LeopardEngine.precision = "lo4" / "split" also covers the decode branch, EVAL:291-320): every operand of a
 * decode-step projection is handed over as a PAIR of 16-bit rows — T(x) and T(x - T(x)), rows m and m + M of a [2 M, K] buffer (2 M <= 16) —
 * and both products land in the same fp32 sums: the operand is seen to ~22 bits at no extra weight traffic (the step is bound by the weight
 * stream; the lo rows are two of the <= 16 batch rows).  The _hl entry points take X with 2 M rows and write every 16-bit output that is itself a
 * projection operand (STORE / SWIGLU results, the producer's norm_out) as such a pair again (out / norm_out need 2 M rows); q | k | v rows and the
 * fp32 stream are single.  lmi_split_rows_hl: fp32 [M, K] -> T [2 M, K] (hi rows, then lo rows); lmi_attn_decode_*_hl: the attention output rows
 * as pairs (out [2 q_rows, ldo]). */
int lmi_gemm_skinny_hl(const void* W, const void* X, void* out, int M, int N, int K, int ldw, int ldx, int ldo, int epilogue, int packed,
                       const float* rowsq_in, int rowsq_parts, int norm_dim, float norm_eps, void* norm_out, int ld_norm, const float* norm_gamma,
                       float* rowsq_out, int dtype, void* stream);
int lmi_rope_qkv_skinny_hl(const void* Wqkv_rope, const void* X, void* qkv, int M, int n_q_heads, int n_kv_heads, int head_dim, int K, int ldw, int ldx,
                           int ldo, int packed, const float* rowsq_in, int rowsq_parts, float norm_eps, const float* cos_all, const float* sin_all,
                           void* k_cache, void* v_cache, int ld_cache, int64_t cache_stride, const int* pos_rows_dev, int dtype, void* stream);
int lmi_split_rows_hl(const float* x, void* out, int M, int K, int ldx, int ldo, int dtype, void* stream);
int lmi_attn_decode_fwd_hl(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* cu_seqlens_k,
                           int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                           int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                           int dtype, void* stream);
int lmi_attn_decode_pool_hl(const void* q, const void* k, const void* v, void* out, const int* cu_seqlens_q, const int* k_begin, const int* k_len,
                            int n_seq, int max_seqlen_q, int max_seqlen_k, int q_rows, int n_heads, int n_kv_heads, int head_dim,
                            int ldq, int ldk, int ldv, int ldo, float scale, int window, void* workspace, int64_t workspace_bytes,
                            int dtype, void* stream);

/* Same with the RMSNorm of the decode step folded in: x is the fp32 residual row [K], norm_weight fp32 [K], and the row
 * fed to the product is T(norm_weight * (x * rsqrt(mean(x^2) + eps))) — the arithmetic of lmi_rmsnorm, without its launch.
 * K = 4096 (the hidden size of Llama-3.1-8B / Mistral-7B). */
int lmi_gemv_rmsnorm(const void* W, const float* x, const float* norm_weight, float eps, void* out, int N, int K, int ldw,
                     int epilogue, int dtype, void* stream);

/* The decode step's q | k | v projection in ONE launch: lmi_gemv_rmsnorm (store epilogue) + lmi_rope_qk_at — RMSNorm folded into
 * the load of the fp32 residual row, RoPE at the device position *pos_dev applied to the fp32 sums, q | k | v written in natural
 * order, K / V appended to cache row *pos_dev.  Wqkv_rope: the q and k rows in weights.rope_permute_rows order, v rows natural
 * (the weight of lmi_rmsnorm_rope, modeling_llama.py:254-277 restated for one row).  K = 4096, head_dim = 128. */
int lmi_gemv_rmsnorm_rope(const void* Wqkv_rope, const float* x, const float* norm_weight, float eps, void* qkv, int n_q_heads, int n_kv_heads,
                          int head_dim, int K, int ldw, const float* cos_all, const float* sin_all, void* k_cache, void* v_cache, int ld_cache,
                          const int* pos_dev, int dtype, void* stream);

/* Last-position lm_head (EVAL:333 restricted to the rows generate() consumes; SURVEY.md 8(b) "lmi_lm_head_last"): for each
 * selected row r of the fp32 residual stream x [., ldx] (row index rows[r], int64 on device; null = row r),
 * out[r, 0..N) = W[N,K] . (norm_weight * (x_row * rsqrt(mean(x_row^2) + eps))) with the final RMSNorm
 * (megatron/legacy/model/rms_norm.py:26-31) applied in the launch (norm_weight null = no normalisation).  The normalised
 * row is kept in fp32 — it is never rounded to T — so these logits carry no activation rounding at all; the kernel is
 * bound by the weight stream (2 bytes per weight).  out: fp32 [n_rows, ldo], ldo >= N; K % 8 == 0, K <= 16380. */
int lmi_lm_head_last(const void* W, const float* x, const int64_t* rows, const float* norm_weight, float eps, float* out, int n_rows,
                     int N, int K, int ldw, int ldx, int ldo, int dtype, void* stream);

/* ---- multi-GPU: RCCL collectives over xGMI (SURVEY.md 8(b) "lmi_allgather / lmi_allreduce wrappers over RCCL communicators
 * ... created by lmi_comm_init(rank, nranks, unique_id) and freed by lmi_comm_destroy", 8(e)).  The reference's evaluation is one
 * process per GPU with no collective (run_eval_llava_siglip_multiimg.sh:9-11); its training side states the exchange pattern
 * these serve: all-gather / reduce-scatter around sequence-parallel norms
 * (Megatron-LM-240603/megatron/core/tensor_parallel/mappings.py:107-145, layers.py:387-454).
 * One process per GPU: lmi_comm_init binds the CURRENT HIP device to `rank`.  Rank 0 calls lmi_comm_unique_id and hands the 128
 * bytes to the other ranks out of band (the host-side rendezvous: torch.distributed's store in leopard_amd/dist.py).  The
 * communicator is the only thing this library ever allocates; collectives are stream-ordered like every other entry point
 * (launch them on a side stream to overlap them with GEMMs).  librccl is bound with dlopen at the first call; when it cannot be
 * loaded, or a collective fails, the call returns LMI_ECOMM — there is no fallback transport.
 * counts are in elements of `dtype` (LMI_F16 / LMI_BF16 / LMI_F32); all reductions are sums. */
int lmi_comm_unique_id(void* id128);
int lmi_comm_init(int rank, int nranks, const void* id128, void** comm_out);
int lmi_comm_destroy(void* comm);
int lmi_comm_size(void* comm);                                   /* ranks RCCL sees in `comm` (negative on error) */
/* recv[r*count_per_rank ...] = send of rank r */
int lmi_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream);
int lmi_allreduce(void* comm, const void* send, void* recv, int64_t count, int dtype, void* stream);
/* recv[0..recv_count) = sum over ranks of send[rank*recv_count ...] */
int lmi_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype, void* stream);
int lmi_broadcast(void* comm, const void* send, void* recv, int64_t count, int dtype, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LEOPARD_AMD_H */
