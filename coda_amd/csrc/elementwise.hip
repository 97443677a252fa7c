// HBM-bound row kernels of the decoder forward: embedding gather, RMSNorm, q/k-norm + RoPE.
// All accesses are 16 B per lane (8 bf16), rows are handled by one wave (or one quarter wave)
// so reductions are shuffle-only.
#include "coda_abi_internal.h"

// ------------------------------------------------------------------ embedding gather
// modeling_coda.py:318
__global__ __launch_bounds__(256) void embed_gather_kernel(const long long* __restrict__ ids, const u32x4_t* __restrict__ table,
                                                           u32x4_t* __restrict__ out, int T, int V, int chunks) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    long long id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const u32x4_t* src = table + (size_t)id * chunks;
    u32x4_t* dst = out + (size_t)row * chunks;
    for (int c = lane; c < chunks; c += 64) dst[c] = src[c];
}

extern "C" int coda_embed_gather(const long long* ids, const void* table, void* out, int T, int V, int H, void* stream) {
    if (T <= 0 || V <= 0 || H <= 0 || H % 8) return CODA_ERR_SHAPE;
    if (!ids || !table || !out) return CODA_ERR_ARG;
    if (((uintptr_t)table | (uintptr_t)out) & 15) return CODA_ERR_ALIGN;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids,
                       (const u32x4_t*)table, (u32x4_t*)out, T, V, H / 8);
    CODA_CHECK_LAUNCH("embed_gather");
    return CODA_OK;
}

// ------------------------------------------------------------------ candidate-row compaction
// generation_utils.py:462 (`logits[mask_index]`): the hidden rows of the count[0] candidates, packed
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, long long ld_src_b, const int* __restrict__ rows,
                                                          const int* __restrict__ count, int max_rows, char* __restrict__ dst,
                                                          long long ld_dst_b, int chunks) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = count ? min(count[0], max_rows) : max_rows;
    if (i >= n) return;
    const u32x4_t* s = (const u32x4_t*)(src + (size_t)rows[i] * ld_src_b);
    u32x4_t* d = (u32x4_t*)(dst + (size_t)i * ld_dst_b);
    for (int c = lane; c < chunks; c += 64) d[c] = s[c];
}

extern "C" int coda_gather_rows(const void* src, long long ld_src, const int* rows, const int* count, int max_rows, void* dst,
                                long long ld_dst, int H, void* stream) {
    if (max_rows <= 0 || H <= 0 || H % 8 || ld_src < H || ld_dst < H) return CODA_ERR_SHAPE;
    if (!src || !rows || !dst) return CODA_ERR_ARG;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) || (ld_src * 2) % 16 || (ld_dst * 2) % 16) return CODA_ERR_ALIGN;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const char*)src, ld_src * 2,
                       rows, count, max_rows, (char*)dst, ld_dst * 2, H / 8);
    CODA_CHECK_LAUNCH("gather_rows");
    return CODA_OK;
}

// Row counts of the chunks of a counted launch: out[c] = clamp(count[0] - c * chunk, 0, chunk).  A long candidate list is
// projected and reduced chunk by chunk (bounded logits buffer); every chunk's kernels read their own count here, so the
// host still never learns the count.
__global__ void chunk_counts_kernel(const int* __restrict__ count, int chunk, int n_chunks, int* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_chunks) out[c] = max(0, min(chunk, count[0] - c * chunk));
}

extern "C" int coda_chunk_counts(const int* count, int chunk, int n_chunks, int* out, void* stream) {
    if (chunk <= 0 || n_chunks <= 0) return CODA_ERR_SHAPE;
    if (!count || !out) return CODA_ERR_ARG;
    hipLaunchKernelGGL(chunk_counts_kernel, dim3((n_chunks + 63) / 64), dim3(64), 0, (hipStream_t)stream, count, chunk, n_chunks, out);
    CODA_CHECK_LAUNCH("chunk_counts");
    return CODA_OK;
}

// inverse of the compaction: dst[rows[i]] = src[i] (the final-normed hidden rows of the last layer's compact tail go
// back to their token rows, where coda_build_candidates' src_rows point)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const char* __restrict__ src, long long ld_src_b, const int* __restrict__ rows,
                                                           int n, char* __restrict__ dst, long long ld_dst_b, int dst_rows, int chunks) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int r = rows[i];
    if (r < 0 || r >= dst_rows) return;
    const u32x4_t* s = (const u32x4_t*)(src + (size_t)i * ld_src_b);
    u32x4_t* d = (u32x4_t*)(dst + (size_t)r * ld_dst_b);
    for (int c = lane; c < chunks; c += 64) d[c] = s[c];
}

extern "C" int coda_scatter_rows(const void* src, long long ld_src, const int* rows, int n, void* dst, long long ld_dst,
                                 int dst_rows, int H, void* stream) {
    if (n <= 0 || dst_rows <= 0 || H <= 0 || H % 8 || ld_src < H || ld_dst < H) return CODA_ERR_SHAPE;
    if (!src || !rows || !dst) return CODA_ERR_ARG;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) || (ld_src * 2) % 16 || (ld_dst * 2) % 16) return CODA_ERR_ALIGN;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const char*)src, ld_src * 2,
                       rows, n, (char*)dst, ld_dst * 2, dst_rows, H / 8);
    CODA_CHECK_LAUNCH("scatter_rows");
    return CODA_OK;
}

// ------------------------------------------------------------------ RMSNorm
// modeling_coda.py:36-41: y = w * bf16(x_f32 * rsqrt(mean(x_f32^2) + eps)), product rounded to bf16.

template <int MAXC>   // MAXC: chunks (of 512 elements) kept in registers per lane
__global__ __launch_bounds__(256) void rmsnorm_kernel(const char* __restrict__ x, long long ldx_b, const u32x4_t* __restrict__ w,
                                                      char* __restrict__ out, long long ldo_b, int T, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const u32x4_t* src = (const u32x4_t*)(x + (size_t)row * ldx_b);
    u32x4_t* dst = (u32x4_t*)(out + (size_t)row * ldo_b);
    const int chunks = H / 8;
    u32x4_t keep[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            keep[i] = src[c];
            float f[8];
            unpack8(keep[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
    }
    for (int c = lane + 64 * MAXC; c < chunks; c += 64) {      // rows longer than the register window
        float f[8];
        unpack8(src[c], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)H + eps);
    auto emit = [&](int c, const u32x4_t xv) {
        float f[8], g[8];
        unpack8(xv, f);
        unpack8(w[c], g);
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = pack_bf16x2(g[2 * i] * round_bf16(f[2 * i] * inv), g[2 * i + 1] * round_bf16(f[2 * i + 1] * inv));
        dst[c] = o;
    };
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) emit(c, keep[i]);
    }
    for (int c = lane + 64 * MAXC; c < chunks; c += 64) emit(c, src[c]);
}

extern "C" int coda_rmsnorm(const void* x, long long ldx, const void* weight, void* out, long long ldo, int T, int H,
                            float eps, void* stream) {
    if (T <= 0 || H <= 0 || H % 8) return CODA_ERR_SHAPE;
    if (!x || !weight || !out) return CODA_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight) & 15 || (ldx * 2) % 16 || (ldo * 2) % 16) return CODA_ERR_ALIGN;
    hipLaunchKernelGGL(rmsnorm_kernel<4>, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx * 2,
                       (const u32x4_t*)weight, (char*)out, ldo * 2, T, H, eps);
    CODA_CHECK_LAUNCH("rmsnorm");
    return CODA_OK;
}

// ------------------------------------------------------------------ q/k RMSNorm over head_dim + rotary embedding
// modeling_coda.py:128-148, modeling_utils.py:70-101.  One (token, head) row of 128 bf16 = 256 B is
// handled by 16 lanes (16 B each); the rotate_half partner of element j is element j^64, held by
// lane^8 of the same 16-lane group.  Rounding points follow the reference's bf16 op sequence:
//   n = w * bf16(x * rsqrt(mean x^2 + eps))            -> bf16
//   y = bf16(n * cos) + bf16(rot(n) * sin)             -> bf16
__global__ __launch_bounds__(256) void qknorm_rope_kernel(char* __restrict__ qkv, long long ld_b, const u32x4_t* __restrict__ qw,
                                                          const u32x4_t* __restrict__ kw, const u32x4_t* __restrict__ cs,
                                                          const u32x4_t* __restrict__ sn, int T, int L, int n_q, int n_heads,
                                                          float eps) {
    const int sub = threadIdx.x & 15;                       // 16-B chunk within the head row
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long total = (long long)T * n_heads;
    const bool live = row < total;
    const long long rr = live ? row : total - 1;            // keep all lanes in the shuffles
    const int t = (int)(rr / n_heads), h = (int)(rr % n_heads);
    u32x4_t* p = (u32x4_t*)(qkv + (size_t)t * ld_b + (size_t)h * 256) + sub;
    float f[8];
    unpack8(*p, f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
    float w[8], c[8], s[8], n[8], other[8];
    unpack8((h < n_q ? qw : kw)[sub], w);
    const int pos = t % L;
    unpack8(cs[(size_t)pos * 16 + sub], c);
    unpack8(sn[(size_t)pos * 16 + sub], s);
#pragma unroll
    for (int j = 0; j < 8; ++j) n[j] = round_bf16(w[j] * round_bf16(f[j] * inv));
#pragma unroll
    for (int j = 0; j < 8; ++j) other[j] = __shfl_xor(n[j], 8, 64);
    const float sign = sub < 8 ? -1.0f : 1.0f;              // rotate_half: first half takes -x2, second half x1
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = round_bf16(n[2 * i] * c[2 * i]) + round_bf16(sign * other[2 * i] * s[2 * i]);
        const float a1 = round_bf16(n[2 * i + 1] * c[2 * i + 1]) + round_bf16(sign * other[2 * i + 1] * s[2 * i + 1]);
        o[i] = pack_bf16x2(a0, a1);
    }
    if (live) *p = o;
}

extern "C" int coda_qknorm_rope(void* qkv, long long ld, const void* q_weight, const void* k_weight, const void* cos,
                                const void* sin, int T, int L, int n_q, int n_kv, float eps, void* stream) {
    if (T <= 0 || L <= 0 || n_q <= 0 || n_kv <= 0 || T % L) return CODA_ERR_SHAPE;
    if (!qkv || !q_weight || !k_weight || !cos || !sin) return CODA_ERR_ARG;
    if (((uintptr_t)qkv | (uintptr_t)q_weight | (uintptr_t)k_weight | (uintptr_t)cos | (uintptr_t)sin) & 15 || (ld * 2) % 16)
        return CODA_ERR_ALIGN;
    if (ld < (long long)(n_q + 2 * n_kv) * 128) return CODA_ERR_SHAPE;
    const int n_heads = n_q + n_kv;                         // q heads then k heads are contiguous in the row
    const long long rows = (long long)T * n_heads;
    hipLaunchKernelGGL(qknorm_rope_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (char*)qkv,
                       ld * 2, (const u32x4_t*)q_weight, (const u32x4_t*)k_weight, (const u32x4_t*)cos, (const u32x4_t*)sin, T, L,
                       n_q, n_heads, eps);
    CODA_CHECK_LAUNCH("qknorm_rope");
    return CODA_OK;
}
