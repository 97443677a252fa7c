// Confidence / token choice from MATERIALISED logits: the path taken when a logits hook, top-p or
// top-k is active (the fused lm-head kernel cannot be used).  Replaces generation_utils.py:39-100:
// temperature, top_p_logits (sort + cumsum), top_k_logits, softmax, argmax / exponential race,
// top1-top2 via full sort, entropy - without any sort: thresholds are found by radix select over
// order-preserving keys (8 bits per pass; mass histograms for top-p, count histograms for top-k).
// One workgroup per row; the row (<= 600 KB) is re-read from L2 by each pass (HBM-bound overall).
#include "coda_abi_internal.h"

namespace {

struct ConfArgs {
    const char* logits; long long ld_b; int is_bf16;
    int N, V;
    const int* count; const int* dst_pos;
    int alg; float temperature, top_p; int top_k;
    const float* noise; unsigned int seed_lo, seed_hi;
    int round_conf; float* conf; long long* x0;
};

__device__ __forceinline__ unsigned int fkey(float f) {       // larger float <-> larger key
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float hash_uniform2(unsigned int seed_lo, unsigned int seed_hi, unsigned int rowkey, unsigned int v) {
    unsigned int h = v * 0x9E3779B1u ^ rowkey * 0x85EBCA77u ^ seed_lo;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += seed_hi;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

template <typename T>
__device__ __forceinline__ T block_reduce(T v, T* red, int tid, T (*op)(T, T)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = op(op(red[0], red[1]), op(red[2], red[3]));
    return v;
}
__device__ __forceinline__ float fmax_op(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float fadd_op(float a, float b) { return a + b; }
__device__ __forceinline__ int iadd_op(int a, int b) { return a + b; }
__device__ __forceinline__ int imin_op(int a, int b) { return min(a, b); }

__global__ __launch_bounds__(256) void confidence_kernel(ConfArgs a) {
    __shared__ float hist_f[256];
    __shared__ int hist_i[256];
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    __shared__ unsigned int sh_prefix;
    __shared__ float sh_acc;
    __shared__ int sh_k;
    __shared__ int sh_wave_tot[4];
    __shared__ int sh_base, sh_cut;

    const int row = blockIdx.x;
    const int n_rows = a.count ? min(a.count[0], a.N) : a.N;
    if (row >= n_rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V;
    const bool sampled = a.temperature > 0.f;
    const char* rp = a.logits + (size_t)row * a.ld_b;

    auto logit = [&](int v) -> float {
        float x;
        if (a.is_bf16) {
            x = bf16_bits_to_f32(((const unsigned short*)rp)[v]);
            if (sampled) x = round_bf16(__fdiv_rn(x, a.temperature));
        } else {
            x = ((const float*)rp)[v];
            if (sampled) x = __fdiv_rn(x, a.temperature);
        }
        return x;
    };
    const int kbits = a.is_bf16 ? 16 : 32;   // significant key bits (bf16 values have zero low halves)

    // ---- softmax statistics of the unfiltered row (needed by top-p)
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, logit(v));
    mx = block_reduce<float>(mx, red_f, tid, fmax_op);

    const bool use_p = a.top_p > 0.f && a.top_p < 1.f;
    const bool use_k = a.top_k > 0 && a.top_k < V;
    unsigned int kp = 0u;        // top-p threshold key (keep key > kp, ties by index up to p_cut)
    int p_cut = 0x7fffffff;      // among key == kp keep indices <= p_cut
    unsigned int kk = 0u;        // top-k threshold key (keep key >= kk)

    if (use_p) {
        float z = 0.f;
        for (int v = tid; v < V; v += 256) z += __expf(logit(v) - mx);
        z = block_reduce<float>(z, red_f, tid, fadd_op);
        const float inv_z = 1.0f / z;
        // radix select: smallest key K whose strictly-above probability mass A(K) <= top_p
        unsigned int prefix = 0u;
        float acc = 0.f;             // mass of keys above the current prefix range
        for (int shift = 24; shift >= 32 - kbits; shift -= 8) {
            hist_f[tid] = 0.f;
            hist_i[tid] = 0;
            __syncthreads();
            const unsigned int hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int v = tid; v < V; v += 256) {
                const float x = logit(v);
                const unsigned int k = fkey(x);
                if ((k & hi_mask) == prefix) {
                    atomicAdd(&hist_f[(k >> shift) & 255], __expf(x - mx) * inv_z);
                    atomicAdd(&hist_i[(k >> shift) & 255], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                float run = acc;
                int pick = -1;
                float pick_acc = acc;
                for (int bin = 255; bin >= 0; --bin) {
                    if (hist_i[bin] == 0) continue;
                    if (run <= a.top_p || pick < 0) { pick = bin; pick_acc = run; }   // pick < 0: the top key is always kept
                    else break;
                    run += hist_f[bin];
                }
                sh_prefix = prefix | ((unsigned int)pick << shift);
                sh_acc = pick_acc;
            }
            __syncthreads();
            prefix = sh_prefix;
            acc = sh_acc;
            __syncthreads();
        }
        kp = prefix;
        // ties at the threshold: the reference keeps them one by one while the running mass <= top_p
        int ties = 0;
        float pt = 0.f;
        for (int v = tid; v < V; v += 256) {
            const float x = logit(v);
            if (fkey(x) == kp) { ++ties; pt = __expf(x - mx) * inv_z; }
        }
        ties = block_reduce<int>(ties, red_i, tid, iadd_op);
        pt = block_reduce<float>(pt, red_f, tid, fmax_op);
        int keep_ties = ties;
        if (pt > 0.f) {
            const float room = (a.top_p - acc) / pt;
            if (room < (float)ties) keep_ties = max(1, min(ties, (int)floorf(room) + 1));
        }
        if (keep_ties < ties) {
            // index of the keep_ties-th tie in index order (ordered scan)
            if (tid == 0) { sh_base = 0; sh_cut = 0x7fffffff; }
            __syncthreads();
            for (int start = 0; start < V; start += 256) {
                const int v = start + tid;
                const bool flag = v < V && fkey(logit(v)) == kp;
                const unsigned long long bal = __ballot(flag);
                const int pre = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) sh_wave_tot[wave] = __popcll(bal);
                __syncthreads();
                int off = sh_base;
                for (int w = 0; w < wave; ++w) off += sh_wave_tot[w];
                if (flag && off + pre + 1 == keep_ties) sh_cut = v;
                __syncthreads();
                if (tid == 0) sh_base += sh_wave_tot[0] + sh_wave_tot[1] + sh_wave_tot[2] + sh_wave_tot[3];
                __syncthreads();
                if (sh_base >= keep_ties) break;
            }
            p_cut = sh_cut;
        }
    }

    if (use_k) {
        // radix select of the top_k-th largest key (duplicates counted); ties with it are all kept
        unsigned int prefix = 0u;
        int want = a.top_k;
        for (int shift = 24; shift >= 32 - kbits; shift -= 8) {
            hist_i[tid] = 0;
            __syncthreads();
            const unsigned int hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int v = tid; v < V; v += 256) {
                const unsigned int k = fkey(logit(v));
                if ((k & hi_mask) == prefix) atomicAdd(&hist_i[(k >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int w = want, pick = 0;
                for (int bin = 255; bin >= 0; --bin) {
                    if (hist_i[bin] >= w) { pick = bin; break; }
                    w -= hist_i[bin];
                }
                sh_prefix = prefix | ((unsigned int)pick << shift);
                sh_k = w;
            }
            __syncthreads();
            prefix = sh_prefix;
            want = sh_k;
            __syncthreads();
        }
        kk = prefix;
        // top_k_logits runs on the top-p filtered row: if top-p kept fewer than top_k tokens the k-th
        // largest is a removed entry (finfo.min) and top-k removes nothing more.
        if (use_p && kk < kp) kk = 0u;
        // (kk == kp with an index cut is covered by the keep predicate below)
    }

    auto kept = [&](int v, unsigned int k) -> bool {
        if (use_p && (k < kp || (k == kp && v > p_cut))) return false;
        if (use_k && k < kk) return false;
        return true;
    };

    // ---- final statistics over the kept set (the maximum is always kept)
    float z = 0.f, s = 0.f, m2 = -INFINITY;
    int arg = 0x7fffffff, n_max = 0;
    float bkey = -INFINITY, bl = 0.f;
    int barg = 0x7fffffff;
    const unsigned int rowkey = a.dst_pos ? (unsigned int)a.dst_pos[row] : (unsigned int)row;
    for (int v = tid; v < V; v += 256) {
        const float x = logit(v);
        if (!kept(v, fkey(x))) continue;
        const float d = x - mx, e = __expf(d);
        z += e;
        s += e * d;
        if (x == mx) { ++n_max; arg = min(arg, v); }
        else m2 = fmaxf(m2, x);
        if (sampled) {
            const float q = a.noise ? a.noise[(size_t)row * V + v] : -__logf(hash_uniform2(a.seed_lo, a.seed_hi, rowkey, (unsigned)v));
            const float key = x - __logf(q);
            if (key > bkey || (key == bkey && v < barg)) { bkey = key; bl = x; barg = v; }
        }
    }
    z = block_reduce<float>(z, red_f, tid, fadd_op);
    s = block_reduce<float>(s, red_f, tid, fadd_op);
    m2 = block_reduce<float>(m2, red_f, tid, fmax_op);
    n_max = block_reduce<int>(n_max, red_i, tid, iadd_op);
    arg = block_reduce<int>(arg, red_i, tid, imin_op);
    if (n_max > 1) m2 = mx;
    if (sampled) {
        const float best = block_reduce<float>(bkey, red_f, tid, fmax_op);
        int cand = bkey == best ? barg : 0x7fffffff;
        cand = block_reduce<int>(cand, red_i, tid, imin_op);
        if (!(bkey == best && barg == cand)) bl = -INFINITY;
        bl = block_reduce<float>(bl, red_f, tid, fmax_op);
        barg = cand;
    }
    if (tid == 0) {
        float c;
        if (a.alg == CODA_ALG_TOPK_MARGIN) c = (1.0f - __expf(m2 - mx)) / z;
        else if (a.alg == CODA_ALG_ENTROPY) c = s / z - __logf(z);
        else c = sampled ? __expf(bl - mx) / z : 1.0f / z;
        if (a.round_conf) c = round_bf16(c);
        const int at = a.dst_pos ? a.dst_pos[row] : row;
        a.conf[at] = c;
        a.x0[at] = sampled ? barg : arg;
    }
}

}  // namespace

extern "C" int coda_confidence_from_logits(const void* logits, long long ld, int is_bf16, int N, int V, const int* count,
                                           const int* dst_pos, int alg, float temperature, float top_p, int top_k,
                                           const float* noise, unsigned long long seed, int round_conf_bf16, float* conf,
                                           long long* x0, void* stream) {
    if (N <= 0 || V <= 0 || ld < V) return CODA_ERR_SHAPE;
    if (alg < 0 || alg > 2) return CODA_ERR_ARG;
    if (!logits || !conf || !x0) return CODA_ERR_ARG;
    ConfArgs a;
    a.logits = (const char*)logits; a.ld_b = ld * (is_bf16 ? 2 : 4); a.is_bf16 = is_bf16;
    a.N = N; a.V = V; a.count = count; a.dst_pos = dst_pos;
    a.alg = alg; a.temperature = temperature; a.top_p = top_p; a.top_k = top_k;
    a.noise = noise; a.seed_lo = (unsigned int)seed; a.seed_hi = (unsigned int)(seed >> 32);
    a.round_conf = round_conf_bf16; a.conf = conf; a.x0 = x0;
    hipLaunchKernelGGL(confidence_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
    CODA_CHECK_LAUNCH("confidence_from_logits");
    return CODA_OK;
}
