/* coda_hip.h - C ABI of libcoda_hip.so: hand-written CDNA4 (gfx950, MI355X) kernels for the
 * masked-diffusion denoising sampler of CoDA (CoDALanguageModel.diffusion_generate).
 *
 * The reference is 100 % Python/PyTorch and has no FFI for this path; every
 * entry point below replaces a sequence of PyTorch ops of the reference, cited as file:line
 * relative to the reference checkout.  Conventions:
 *   - plain pointers to DEVICE memory, sizes and a hipStream_t passed as void*; no torch types;
 *   - every call only enqueues work on `stream` and returns (no sync, no allocation);
 *   - return value: 0 = ok, <0 = error (CODA_ERR_*); nothing is launched on error;
 *   - bf16 tensors are raw uint16 storage, row-major, leading dimensions in ELEMENTS;
 *   - no ownership transfer: all buffers (including workspaces) belong to the caller;
 *   - no tunable global state: a call's result and kernel choice depend on its arguments only.  The library keeps,
 *     per device, the CU count and which kernels already had their dynamic-LDS limit raised (read-once facts), and a
 *     thread-local last-error string.  Calls must be made with the stream's device current (hipSetDevice).
 */
#ifndef CODA_HIP_H
#define CODA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CODA_ABI_VERSION 2

#define CODA_OK 0
#define CODA_ERR_SHAPE (-1)  /* unsupported / inconsistent dimensions */
#define CODA_ERR_ALIGN (-2)  /* pointer or leading dimension not aligned as required */
#define CODA_ERR_LAUNCH (-3) /* HIP launch failure */
#define CODA_ERR_ARG (-4)    /* null pointer / bad enum */

/* epilogues of coda_gemm_bf16 */
#define CODA_EPI_PLAIN 0    /* C = bf16(X W^T) */
#define CODA_EPI_RESIDUAL 1 /* C = bf16(bf16(X W^T) + R)                    modeling_coda.py:247,253 */
#define CODA_EPI_SWIGLU 2   /* C = bf16(bf16(silu(gate)) * up), W = gate|up interleaved per 16 rows; :59 */
#define CODA_EPI_F32 3      /* C = fp32(X W^T) */

/* confidence algorithms (generation_utils.py:485-506) */
#define CODA_ALG_MASKGIT_PLUS 0
#define CODA_ALG_TOPK_MARGIN 1
#define CODA_ALG_ENTROPY 2

int coda_version(void);

/* Last HIP error string seen by the library on this thread (for diagnostics). */
const char* coda_last_error(void);

/* embed_tokens lookup.  Replaces modeling_coda.py:318.
 * ids int64 [T]; table bf16 [V,H]; out bf16 [T,H].  H % 8 == 0.  ids outside [0,V) -> CODA_ERR not
 * detectable on device: rows are clamped to [0,V-1]. */
int coda_embed_gather(const long long* ids, const void* table, void* out, int T, int V, int H, void* stream);

/* CoDARMSNorm.  Replaces modeling_coda.py:36-41 (fp32 x*rsqrt(mean(x^2)+eps) -> bf16 -> * weight).
 * x, out bf16 [T,H] (ld in elements), weight bf16 [H].  H % 8 == 0. */
int coda_rmsnorm(const void* x, long long ldx, const void* weight, void* out, long long ldo, int T, int H, float eps,
                 void* stream);

/* Linear layers.  Replaces nn.Linear at modeling_coda.py:58-60,123-125,155,411 and the ops fused
 * behind them (see CODA_EPI_*).  X bf16 [M,K] (optionally row-gathered: row i is X[x_rows[i]]),
 * W bf16 [N,K] (nn.Linear layout), C [M,N] (SWIGLU: [M,N/2]).  K % 64 == 0, N % 4 == 0. */
int coda_gemm_bf16(const void* X, long long ldx, const int* x_rows, const void* W, long long ldw, void* C,
                   long long ldc, const void* R, long long ldr, int M, int N, int K, int epilogue, void* stream);

/* coda_gemm_bf16 with the tile engine and store width chosen by the CALLER (A/B benches and the per-engine exactness
 * tests; the product path calls coda_gemm_bf16 = engine 0, flags 0).  engine: 0 = automatic choice by shape,
 * 1 = 128x128, 2 = 256x256 one tile per workgroup, 3 = persistent 256x256, 4 = persistent 256x128, 5 / 6 = persistent
 * 192x256 with the 3-stage ring on the weights / the activations, 7 = persistent 256x256 with a 3-stage activation
 * ring, 8 / 9 = four-wave persistent 256x256 / 192x256 (one wave per SIMD, wave tile 128 / 96 tokens x 128 features).  flags: CODA_GEMM_NARROW_STORES = interior tiles store 8-byte pieces per lane instead of 16-byte pieces after
 * a v_permlane16_swap exchange.  Results are identical up to fp32 summation order. */
#define CODA_GEMM_NARROW_STORES 1
int coda_gemm_bf16_ex(const void* X, long long ldx, const int* x_rows, const void* W, long long ldw, void* C,
                      long long ldc, const void* R, long long ldr, int M, int N, int K, int epilogue, int engine,
                      int flags, void* stream);

/* coda_gemm_bf16 for SMALL M (the B = 1 serving shape, M = 384; the compact last-layer tail): same result contract
 * (CODA_EPI_PLAIN / RESIDUAL / SWIGLU; bf16 rounding points as above), computed as `ksplit` K ranges whose fp32 partial
 * tiles go to `workspace` (ksplit * M * N * 4 bytes) and are summed in a fixed order by a second kernel (deterministic).
 * Why: at M = 384 a projection has 48-96 tiles, so 48-96 of 256 CUs do all the loading; split-K engages the others.
 * ksplit <= 1 forwards to coda_gemm_bf16.  K % ksplit == 0, (K / ksplit) % 64 == 0, N % 32 == 0.  The caller picks ksplit
 * (coda_amd/engine.py::splitk_factor): the library keeps no heuristics state. */
int coda_gemm_bf16_splitk(const void* X, long long ldx, const void* W, long long ldw, void* C, long long ldc,
                          const void* R, long long ldr, int M, int N, int K, int epilogue, int ksplit, void* workspace,
                          long long workspace_bytes, void* stream);

/* coda_gemm_bf16_splitk(CODA_EPI_RESIDUAL) followed by coda_rmsnorm on its output, in ONE reduce launch: C = bf16(bf16(X W^T)
 * + R) (modeling_coda.py:247,253) and normed = CoDARMSNorm(C) (:36-41: the post-attention norm, the next layer's input norm
 * or the final norm).  Same bits as the two calls (same summation orders).  B = 1 serving shape: one launch fewer per
 * projection.  N % 8 == 0, N <= 4096; other constraints as coda_gemm_bf16_splitk; ksplit >= 2. */
int coda_gemm_bf16_splitk_norm(const void* X, long long ldx, const void* W, long long ldw, void* C, long long ldc,
                               const void* R, long long ldr, int M, int N, int K, int ksplit, void* workspace,
                               long long workspace_bytes, const void* norm_weight, void* normed, long long ld_normed, float eps,
                               void* stream);

/* coda_gemm_bf16_splitk(CODA_EPI_PLAIN) of the fused q|k|v projection followed by coda_qknorm_rope, in ONE reduce launch
 * (modeling_coda.py:123-148, modeling_utils.py:70-101).  W bf16 [(n_q + 2 n_kv) * 128, K] in natural row order.  Same bits as
 * the two calls.  ksplit >= 2, K % ksplit == 0, (K / ksplit) % 64 == 0; workspace ksplit * T * N * 4 bytes. */
int coda_qkv_proj_rope_splitk(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                              const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T, int L, int n_q,
                              int n_kv, int K, float eps, int ksplit, void* workspace, long long workspace_bytes, void* stream);

/* q_norm / k_norm + rotary embedding, in place on the fused QKV activation.
 * Replaces modeling_coda.py:128-148 and modeling_utils.py:70-101.
 * qkv bf16 [T, (n_q + 2 n_kv) * 128]: q heads, then k heads, then v heads (v untouched).
 * cos/sin bf16 [L,128] (already cast to bf16 as the reference does, modeling_coda.py:197);
 * position of token t is t % L.  head_dim must be 128. */
int coda_qknorm_rope(void* qkv, long long ld, const void* q_weight, const void* k_weight, const void* cos,
                     const void* sin, int T, int L, int n_q, int n_kv, float eps, void* stream);

/* QKV projection with q_norm / k_norm + rotary fused into the GEMM epilogue: one launch replacing
 * coda_gemm_bf16(PLAIN) + coda_qknorm_rope (modeling_coda.py:123-148, modeling_utils.py:70-101), same results.
 * W bf16 [(n_q+2 n_kv)*128, K]: q|k|v rows, with the 128 rows of every q and k head stored in the order
 * [0..31, 64..95, 32..63, 96..127] (rotary partners j, j+64 then share an MFMA lane); v rows unpermuted.
 * Requires T >= 256, n_q and n_kv even; returns CODA_ERR_SHAPE otherwise (use the two-call path). */
int coda_qkv_proj_rope(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                       const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T, int L, int n_q,
                       int n_kv, int K, float eps, void* stream);

/* The same fused launch on the four-wave engine (one wave per SIMD, a wave's 128 features = one head): W in its NATURAL
 * row order (q | k | v heads, no permutation), same results.  Requires T >= 256, n_q and n_kv even. */
int coda_qkv_proj_rope_w4(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                          const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T, int L, int n_q,
                          int n_kv, int K, float eps, void* stream);

/* Candidate-row compaction and the projection over a device-side row count: the two halves of
 * `logits[mask_index]` (generation_utils.py:462) when logits must be materialised (top-p, top-k).
 * coda_gather_rows: dst[i] = src[rows[i]] for i < min(count[0], max_rows) (count NULL: all), H bf16 per row.
 * coda_gemm_bf16_counted: C[m, :] = bf16(X[m] W^T) for m < min(m_count[0], M_max); rows beyond are not
 * touched.  M_max only bounds the launch. */
int coda_gather_rows(const void* src, long long ld_src, const int* rows, const int* count, int max_rows, void* dst,
                     long long ld_dst, int H, void* stream);
int coda_gemm_bf16_counted(const void* X, long long ldx, const int* m_count, const void* W, long long ldw, void* C,
                           long long ldc, int M_max, int N, int K, void* stream);

/* out[c] = clamp(count[0] - c * chunk, 0, chunk) for c < n_chunks: the device-side row counts of the chunks when a long
 * candidate list goes through coda_gather_rows / coda_gemm_bf16_counted / coda_confidence_from_logits chunk by chunk
 * (`logits[mask_index]` of generation_utils.py:457 with a bounded logits buffer; no counterpart line in the reference). */
int coda_chunk_counts(const int* count, int chunk, int n_chunks, int* out, void* stream);

/* Inverse of coda_gather_rows: dst[rows[i]] = src[i] for i < n (rows outside [0, dst_rows) are skipped), H bf16 per row.
 * Used by the last decoder layer's compact tail: only the hidden rows whose logits the sampler consumes
 * (generation_utils.py:457,462: positions p-1 of the candidate positions p) go through o_proj / the MLP / the final
 * norm of the LAST layer (modeling_coda.py:247-253,353); their final-normed rows are written back to their token rows. */
int coda_scatter_rows(const void* src, long long ld_src, const int* rows, int n, void* dst, long long ld_dst,
                      int dst_rows, int H, void* stream);

/* Non-causal, mask-free GQA flash attention.  Replaces attention.py:12-90 (repeat_kv + SDPA with
 * is_causal=False and no mask).  Reads q/k/v from the fused QKV activation
 * (layout as above), writes out bf16 [T, n_q*128] (token-major, heads concatenated: what o_proj
 * consumes, modeling_coda.py:153-154).  T = B*L, head_dim 128, n_q % n_kv == 0. */
int coda_flash_attn_noncausal(const void* qkv, long long ld_qkv, void* out, long long ld_out, int B, int L, int n_q,
                              int n_kv, float scale, void* stream);
/* Same, with the work split chosen by the caller (A/B benches, bit-identity tests): pair_mode -1 = automatic,
 * 0 = one query head per workgroup, 1 = the two query heads of a GQA pair share one K/V stream; qsplit 0 = automatic,
 * > 0 = workgroups per (batch, head).  Results do not depend on either. */
int coda_flash_attn_noncausal_ex(const void* qkv, long long ld_qkv, void* out, long long ld_out, int B, int L, int n_q,
                                 int n_kv, float scale, int pair_mode, int qsplit, void* stream);

/* Diagnostic: 1 if `logits / temperature` (generation_utils.py:71) runs on the verified three-operation quotient for bf16
 * logits at this temperature, 0 if the kernels take the IEEE division (coda_amd/csrc/coda_common.h::coda_div_by: the fast
 * form is licensed per temperature by an exhaustive host check over all bf16 inputs; fp32 logits always divide).  Either
 * way the quotient is the correctly rounded one; host-only, touches no GPU. */
int coda_temperature_fast_division(float temperature);

/* Candidate list for one denoise step.  Replaces `mask_index = x == mask_token_id`,
 * the shift `cat(logits[:, :1], logits[:, :-1])` and the boolean gather `logits[mask_index]`
 * (generation_utils.py:455-462) without touching logits: for every masked position p of row b
 * inside [win_lo, win_hi), in row-major order, emits src_row = b*L + max(p-1,0) (the hidden row
 * whose lm-head output predicts p) and dst = b*L + p.  count[0] receives the number of
 * candidates.  src_rows/dst_pos must hold B*(win_hi-win_lo) ints. */
int coda_build_candidates(const long long* x, int B, int L, int win_lo, int win_hi, long long mask_id, int* src_rows,
                          int* dst_pos, int* count, void* stream);

/* Fused lm-head + online softmax + confidence.  Replaces modeling_coda.py:411 (lm_head) and
 * generation_utils.py:61-100 (sample_tokens) for top_p = top_k = None; never materialises logits.
 * hidden bf16 [*, H] final-normed hidden states; src_rows/count from coda_build_candidates;
 * W bf16 [V,H].  For candidate i: logits = bf16(hidden[src_rows[i]] W^T) (/temperature if > 0),
 * x0 = argmax (first index) or, if temperature > 0, argmax(logit - log(noise)) with
 * noise fp32 [max_rows, V] ~ Exp(1) (the exponential race of Categorical.sample());
 * noise == NULL with temperature > 0: Exp(1) variates come from an in-kernel counter hash of
 * (seed, dst_pos[i], vocab index) - coda_hash_uniform in coda_amd/csrc/coda_common.h, restated on the host by
 * tests/parity.py::hash_uniform_host;
 * confidence per `alg`.  Outputs are written at conf[dst_pos[i]], x0[dst_pos[i]].
 * workspace: coda_lmhead_workspace_bytes(max_rows, V) bytes.  max_rows bounds the grid; rows
 * beyond count[0] do no work.  H % 64 == 0, V % 4 == 0. */
long long coda_lmhead_workspace_bytes(int max_rows, int V);
int coda_lmhead_confidence(const void* hidden, long long ld_hidden, const int* src_rows, const int* dst_pos,
                           const int* count, int max_rows, const void* W, long long ldw, int V, int H, int alg,
                           float temperature, const float* noise, unsigned long long seed, int round_conf_bf16,
                           float* conf, long long* x0, void* workspace, void* stream);

/* coda_lmhead_confidence for callers that cannot allocate and cannot hand over a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE (which is PyTorch and
 * allocates freely): it exists for those callers only.  Same arguments, kernels and results as coda_lmhead_confidence,
 * minus `workspace`: the partial statistics go to a 4 MiB array that the library holds in each GPU's memory for the
 * life of the process.  Calls no hipMalloc / hipFree or allocating routine.  Limits: max_rows <=
 * CODA_LMHEAD_NOALLOC_MAX_ROWS (CODA_ERR_SHAPE beyond, with a message in coda_last_error()); every call on one GPU
 * shares that GPU's array, so calls that OVERLAP on one GPU (two streams, two threads) race - serialise them.  The first
 * call on a device looks the array's address up (hipGetSymbolAddress, no allocation): make it outside graph capture. */
#define CODA_LMHEAD_NOALLOC_MAX_ROWS 131072
int coda_lmhead_confidence_noalloc(const void* hidden, long long ld_hidden, const int* src_rows, const int* dst_pos,
                                   const int* count, int max_rows, const void* W, long long ldw, int V, int H, int alg,
                                   float temperature, const float* noise, unsigned long long seed, int round_conf_bf16,
                                   float* conf, long long* x0, void* stream);

/* Unfused confidence from materialised logits (hooks, top-p, top-k).  Replaces
 * generation_utils.py:39-100.  logits bf16 or fp32 [N,V] (is_bf16), one row per candidate i < count[0]
 * (count may be NULL: all N rows).  top_p <= 0 or >= 1 disables top-p; top_k <= 0 disables top-k.
 * Outputs at conf[dst_pos[i]], x0[dst_pos[i]] (dst_pos NULL: at i). */
int coda_confidence_from_logits(const void* logits, long long ld, int is_bf16, int N, int V, const int* count,
                                const int* dst_pos, int alg, float temperature, float top_p, int top_k,
                                const float* noise, unsigned long long seed, int round_conf_bf16, float* conf,
                                long long* x0, void* stream);

/* Schedule + selection + scatter.  Replaces generation_utils.py:507-538.
 * k = trunc((count/B) * ratio) in fp32 (ratio = 1 - s/t computed by the host exactly as the
 * reference does; take_all != 0 on the last step of a schedule: k = trunc(count/B)).
 * Per row b, among masked positions in [win_lo, win_hi): alg_temp == 0 -> the k largest conf
 * (ties: lowest index); alg_temp > 0 -> the k largest softmax(conf/alg_temp)/noise with
 * noise fp32 [B,L] ~ Exp(1) (== torch.multinomial without replacement); noise == NULL: the Exp(1) variate of
 * position (b, p) is -log(coda_hash_uniform(seed, b*L + p, 0xFFFFFFFF)), generated in the kernel (no RNG launch,
 * no noise tensor).  Writes x[b,p] = x0[b,p] in place for the selected p.  k_out (nullable) receives k. */
int coda_select_unmask(long long* x, const float* conf, const long long* x0, const int* count, int B, int L,
                       int win_lo, int win_hi, long long mask_id, float ratio, int take_all, float alg_temp,
                       const float* noise, unsigned long long seed, int* k_out, void* stream);

/* `confidence_threshold` selection for ONE sequence.  Replaces the serving sampler's
 * serving/dream/model/generation_utils.py:442-465: among the masked positions of x[0, win_lo:win_hi) take
 * the k largest conf (ties: lowest index); the largest is always revealed, each of the others only if
 * !(conf < threshold).  Writes x[p] = x0[p] for the revealed p; info[0] = number revealed,
 * info[1] = number of the k held back (the reference's left_tokens_last_step increment). */
int coda_select_threshold(long long* x, const float* conf, const long long* x0, int L, int win_lo, int win_hi,
                          long long mask_id, int k, float threshold, int* info, void* stream);

/* `origin` algorithm transfer (generation_utils.py:466-483): every candidate i < count[0] with
 * u[i] < p_transfer takes x[dst_pos[i]] = x0[dst_pos[i]].  u fp32 [count] uniform(0,1); u == NULL: the uniform of
 * candidate i is coda_hash_uniform(seed, dst_pos[i], 0xFFFFFFFE), generated in the kernel. */
int coda_origin_transfer(long long* x, const long long* x0, const int* dst_pos, const int* count, int max_rows,
                         const float* u, unsigned long long seed, float p_transfer, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_HIP_H */
