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
    int alg; float temperature, inv_temperature, top_p; int top_k;
    const float* noise; unsigned int seed_lo, seed_hi;
    int round_conf; float* conf; long long* x0;
    int only_marked;     // generic kernel as the fallback of the histogram kernel: only rows whose x0 slot holds -1
};

__device__ __forceinline__ unsigned int fkey(float f) {       // larger float <-> larger key
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
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
    // Radix histograms: NSUB private copies, one per group of 64 / NSUB lanes.  bf16 logits share a handful of
    // sign|exponent bytes, so on one copy nearly all 64 lanes of an LDS atomic hit the same bin and serialise.
    constexpr int NSUB = 4;
    __shared__ unsigned long long hist_m[NSUB * 256];
    __shared__ int hist_i[NSUB * 256];
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    __shared__ unsigned int sh_prefix;
    __shared__ unsigned long long sh_acc, sh_lim, sh_tie_mass;
    __shared__ int sh_seg[256];
    __shared__ int sh_k;
    __shared__ int sh_cut;
    __shared__ int sh_single;

    const int row = blockIdx.x;
    const int n_rows = a.count ? min(a.count[0], a.N) : a.N;
    if (row >= n_rows) return;
    if (a.only_marked && a.x0[a.dst_pos ? a.dst_pos[row] : row] != -1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int sub = (lane & (NSUB - 1)) * 256;        // this lane's histogram copy
    const int V = a.V;
    const bool sampled = a.temperature > 0.f;
    const char* rp = a.logits + (size_t)row * a.ld_b;

    auto logit = [&](int v) -> float {
        float x;
        if (a.is_bf16) {
            x = bf16_bits_to_f32(((const unsigned short*)rp)[v]);
            if (sampled) x = round_bf16(coda_div_by(x, a.temperature, a.inv_temperature));
        } else {
            x = ((const float*)rp)[v];
            if (sampled) x = coda_div_by(x, a.temperature, a.inv_temperature);
        }
        return x;
    };
    const int kbits = a.is_bf16 ? 16 : 32;   // significant key bits (bf16 values have zero low halves)
    // Keys are compared on their significant bits only: fkey() complements negative floats, so a negative bf16 value
    // carries 0xFFFF in its low half while the radix prefix (kp / kk) has zeros there.
    const unsigned int kmask = 0xffffffffu << (32 - kbits);
    auto key_of = [&](float x) -> unsigned int { return fkey(x) & kmask; };
    // Every pass streams the row through fn(v, x).  16-byte loads (8 bf16 / 4 fp32 per lane) when the row is 16-byte
    // aligned: with 2-byte loads the passes were bound by the NUMBER of load instructions (one per element and lane),
    // 8x more than needed.  Reductions below do not depend on the order in which a thread meets its elements.
    const bool vec_ok = (((uintptr_t)rp) & 15) == 0;
    const int per = a.is_bf16 ? 8 : 4;
    const int nvec = vec_ok ? V / per : 0;
    auto xform = [&](float x) -> float {
        if (!sampled) return x;
        x = coda_div_by(x, a.temperature, a.inv_temperature);
        return a.is_bf16 ? round_bf16(x) : x;
    };
    auto for_each = [&](auto&& fn) {
        if (a.is_bf16) {
            // four loads in flight per lane: one 16-byte load per iteration leaves ~25 KB in flight per CU, well under
            // what HBM latency x bandwidth asks for
            int cidx = tid;
            for (; cidx + 3 * 256 < nvec; cidx += 4 * 256) {
                u32x4_t raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = *(const u32x4_t*)(rp + (size_t)(cidx + u * 256) * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        fn((cidx + u * 256) * 8 + 2 * j, xform(bf16_lo(raw[u][j])));
                        fn((cidx + u * 256) * 8 + 2 * j + 1, xform(bf16_hi(raw[u][j])));
                    }
            }
            for (; cidx < nvec; cidx += 256) {
                const u32x4_t raw = *(const u32x4_t*)(rp + (size_t)cidx * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    fn(cidx * 8 + 2 * j, xform(bf16_lo(raw[j])));
                    fn(cidx * 8 + 2 * j + 1, xform(bf16_hi(raw[j])));
                }
            }
        } else {
            for (int cidx = tid; cidx < nvec; cidx += 256) {
                const f32x4_t raw = *(const f32x4_t*)(rp + (size_t)cidx * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) fn(cidx * 4 + j, xform(raw[j]));
            }
        }
        for (int v = nvec * per + tid; v < V; v += 256) fn(v, logit(v));
    };

    const bool use_p = a.top_p > 0.f && a.top_p < 1.f;
    const bool use_k = a.top_k > 0 && a.top_k < V;

    if (!use_p && !use_k && !sampled) {
        // ---- greedy, nothing filtered (the materialised lm-head route of long candidate lists): ONE pass over the row.
        // Every thread keeps an online softmax of its own entries - maximum, second maximum (duplicates counted), first
        // index of the maximum, sum e^(x-m) and sum e^(x-m)(x-m) relative to ITS running maximum, rescaled on the rare
        // step up - and the 256 partial states are merged once.  The row is read from memory once instead of twice.
        float m = -INFINITY, m2 = -INFINITY, z = 0.f, s = 0.f;
        int arg = 0x7fffffff;
        for_each([&](int v, float x) {
            if (x > m) {
                if (z > 0.f) {
                    const float d = m - x, sc = __expf(d);
                    s = sc * (s + d * z);
                    z *= sc;
                }
                z += 1.0f;
                m2 = m;
                m = x;
                arg = v;
            } else if (x > -INFINITY) {                  // (-inf entries - banned by a logits hook - carry nothing)
                const float d = x - m, e = __expf(d);
                z += e;
                s += e * d;
                m2 = fmaxf(m2, x);
                if (x == m) arg = min(arg, v);
            }
        });
        const float mx = block_reduce<float>(m, red_f, tid, fmax_op);
        const bool top = m == mx;
        float zt = 0.f, st = 0.f;
        if (z > 0.f) {
            const float d = m - mx, sc = __expf(d);
            zt = z * sc;
            st = sc * (s + d * z);
        }
        zt = block_reduce<float>(zt, red_f, tid, fadd_op);
        st = block_reduce<float>(st, red_f, tid, fadd_op);
        const int holders = block_reduce<int>(top ? 1 : 0, red_i, tid, iadd_op);
        float r2 = block_reduce<float>(top ? m2 : m, red_f, tid, fmax_op);
        if (holders > 1) r2 = mx;
        const int rarg = block_reduce<int>(top ? arg : 0x7fffffff, red_i, tid, imin_op);
        if (tid == 0) {
            float c;
            if (a.alg == CODA_ALG_TOPK_MARGIN) c = (1.0f - __expf(r2 - mx)) / zt;
            else if (a.alg == CODA_ALG_ENTROPY) c = st / zt - __logf(zt);
            else c = 1.0f / zt;
            if (a.round_conf) c = round_bf16(c);
            const int at = a.dst_pos ? a.dst_pos[row] : row;
            a.conf[at] = c;
            a.x0[at] = rarg;
        }
        return;
    }

    // ---- softmax statistics of the unfiltered row (needed by top-p)
    float mx = -INFINITY;
    for_each([&](int, float x) { mx = fmaxf(mx, x); });
    mx = block_reduce<float>(mx, red_f, tid, fmax_op);
    unsigned int kp = 0u;        // top-p threshold key (keep key > kp, ties by index up to p_cut)
    int p_cut = 0x7fffffff;      // among key == kp keep indices <= p_cut
    unsigned int kk = 0u;        // top-k threshold key (keep key >= kk)

    if (use_p) {
        // Probability mass is accumulated in 64-bit FIXED POINT (e^(x - max) * 2^40, integer LDS atomics): float LDS
        // atomics on the few hot bins were 70 % of this kernel's time, integer ones are an order cheaper, exact and
        // order-independent (the float histogram rounded differently from run to run).  152k entries of at most 2^40
        // sum below 2^58; one entry is resolved to 2^-40 of the row maximum's mass.  The first radix pass sees every
        // entry, so its bins also give the normaliser: thresholds are compared against top_p * (total mass).
        constexpr float FIX = 1099511627776.0f;               // 2^40
        auto mass = [&](float x) -> unsigned long long { return (unsigned long long)(__expf(x - mx) * FIX); };
        // radix select: smallest key K whose strictly-above probability mass A(K) <= top_p
        unsigned int prefix = 0u;
        unsigned long long acc = 0ull;   // mass of keys above the current prefix range
        unsigned long long lim = 0ull;   // top_p * total mass
        for (int shift = 24; shift >= 32 - kbits; shift -= 8) {
#pragma unroll
            for (int c = 0; c < NSUB; ++c) {
                hist_m[c * 256 + tid] = 0ull;
                hist_i[c * 256 + tid] = 0;
            }
            __syncthreads();
            const unsigned int hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            int n_top = 0;                           // first level only: entries equal to the row maximum
            for_each([&](int, float x) {
                if (shift == 24) n_top += x == mx ? 1 : 0;
                // Entries whose fixed-point mass is zero (x more than ~28 below the row maximum: e^-28 * 2^40 < 1) cannot
                // move the threshold: the running mass crosses top_p * total at an entry with non-zero mass, every
                // zero-mass entry sorts below it, and the tie count at the last level is taken at the threshold key
                // itself.  On peaked rows (low temperature, trained models) that is nearly the whole vocabulary, and the
                // two LDS atomics per entry are what this pass costs.
                const unsigned long long ms = mass(x);
                if (ms == 0ull) return;
                const unsigned int k = key_of(x);
                if ((k & hi_mask) == prefix) {
                    atomicAdd(&hist_m[sub + ((k >> shift) & 255)], ms);
                    atomicAdd(&hist_i[sub + ((k >> shift) & 255)], 1);
                }
            });
            __syncthreads();
            if (shift == 24) n_top = block_reduce<int>(n_top, red_i, tid, iadd_op);
            {
                unsigned long long hm = 0ull;
                int hi = 0;
#pragma unroll
                for (int c = 0; c < NSUB; ++c) {
                    hm += hist_m[c * 256 + tid];
                    hi += hist_i[c * 256 + tid];
                }
                __syncthreads();
                hist_m[tid] = hm;
                hist_i[tid] = hi;
            }
            __syncthreads();
            if (tid == 0) {
                if (shift == 24) {
                    unsigned long long tot = 0ull;
                    for (int bin = 0; bin < 256; ++bin) tot += hist_m[bin];
                    sh_lim = (unsigned long long)((double)a.top_p * (double)tot);
                    // A single row maximum whose own mass already exceeds top_p * total is the whole nucleus (the second key
                    // would be kept only if the mass above it were <= the limit): the same fixed-point comparison the
                    // levels below would end on, known one pass over the row earlier - most rows of a trained model late
                    // in the denoising, or at the lm-eval scripts' temperature of 0.1.
                    sh_single = n_top == 1 && mass(mx) > sh_lim;
                }
                const unsigned long long lm = sh_lim;
                unsigned long long run = acc, pick_acc = acc;
                int pick = -1;
                for (int bin = 255; bin >= 0; --bin) {
                    if (hist_i[bin] == 0) continue;
                    if (run <= lm || pick < 0) { pick = bin; pick_acc = run; }   // pick < 0: the top key is always kept
                    else break;
                    run += hist_m[bin];
                }
                sh_prefix = prefix | ((unsigned int)pick << shift);
                sh_acc = pick_acc;
                sh_k = hist_i[pick];                 // at the last level: the entries equal to the threshold key ...
                sh_tie_mass = hist_m[pick];          // ... and their total mass
            }
            __syncthreads();
            prefix = sh_prefix;
            acc = sh_acc;
            lim = sh_lim;
            const bool single = sh_single != 0;
            __syncthreads();
            if (single) {                            // (wave-uniform: read from LDS after the barrier)
                prefix = key_of(mx);
                if (tid == 0) { sh_k = 1; sh_tie_mass = mass(mx); }
                __syncthreads();
                break;
            }
        }
        kp = prefix;
        // ties at the threshold: the reference keeps them one by one while the running mass <= top_p
        const int ties = sh_k;
        const unsigned long long pt = ties > 0 ? sh_tie_mass / (unsigned long long)ties : 0ull;   // equal keys: equal masses
        int keep_ties = ties;
        if (pt > 0ull && lim >= acc) {
            const unsigned long long room = (lim - acc) / pt;
            if (room < (unsigned long long)ties) keep_ties = max(1, min(ties, (int)room + 1));
        } else if (pt > 0ull) {
            keep_ties = 1;
        }
        if (keep_ties < ties) {
            // index of the keep_ties-th tie in index order: every thread counts the ties of its own contiguous segment,
            // a scan over the 256 counts finds the segment that holds it, and that thread walks its segment again
            const int seg = ((V + 255) / 256 + 7) & ~7;        // multiple of 8: 16-byte aligned segments of bf16
            const int v0 = min(V, tid * seg), v1 = min(V, v0 + seg);
            const bool seg_vec = vec_ok && a.is_bf16 && (V & 7) == 0;
            // walk(fn): fn(v, is_tie) over the segment in index order; fn returns true to stop
            auto walk = [&](auto&& fn) {
                if (seg_vec) {
                    for (int v = v0; v < v1; v += 8) {
                        const u32x4_t raw = *(const u32x4_t*)(rp + (size_t)v * 2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (fn(v + 2 * j, key_of(xform(bf16_lo(raw[j]))) == kp)) return;
                            if (fn(v + 2 * j + 1, key_of(xform(bf16_hi(raw[j]))) == kp)) return;
                        }
                    }
                } else {
                    for (int v = v0; v < v1; ++v)
                        if (fn(v, key_of(logit(v)) == kp)) return;
                }
            };
            int mine = 0;
            walk([&](int, bool tie) { mine += tie ? 1 : 0; return false; });
            sh_seg[tid] = mine;
            if (tid == 0) sh_cut = 0x7fffffff;
            __syncthreads();
            int before = 0;
            for (int w = 0; w < tid; ++w) before += sh_seg[w];
            if (before < keep_ties && before + mine >= keep_ties) {
                int need = keep_ties - before;
                walk([&](int v, bool tie) {
                    if (tie && --need == 0) { sh_cut = v; return true; }
                    return false;
                });
            }
            __syncthreads();
            p_cut = sh_cut;
        }
    }

    if (use_k) {
        // radix select of the top_k-th largest key (duplicates counted); ties with it are all kept
        unsigned int prefix = 0u;
        int want = a.top_k;
        for (int shift = 24; shift >= 32 - kbits; shift -= 8) {
#pragma unroll
            for (int c = 0; c < NSUB; ++c) hist_i[c * 256 + tid] = 0;
            __syncthreads();
            const unsigned int hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for_each([&](int, float x) {
                const unsigned int k = key_of(x);
                if ((k & hi_mask) == prefix) atomicAdd(&hist_i[sub + ((k >> shift) & 255)], 1);
            });
            __syncthreads();
            {
                int hi = 0;
#pragma unroll
                for (int c = 0; c < NSUB; ++c) hi += hist_i[c * 256 + tid];
                __syncthreads();
                hist_i[tid] = hi;
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
    for_each([&](int v, float x) {
        if (!kept(v, key_of(x))) return;
        const float d = x - mx, e = __expf(d);
        z += e;
        s += e * d;
        if (x == mx) { ++n_max; arg = min(arg, v); }
        else m2 = fmaxf(m2, x);
        // In-kernel noise is bounded: u in [2^-24, 1 - 2^-24] gives -log q in [-2.82, 16.64], so the row maximum's key is at
        // least mx - 2.82 and an entry more than 19.5 below the maximum can never win the race: no hash, no logs for it
        // (on peaked rows that is nearly the whole vocabulary, and the race is what a sampled pass costs).
        if (sampled && (a.noise || x >= mx - CODA_RACE_WINDOW)) {
            const float q = a.noise ? a.noise[(size_t)row * V + v] : -__logf(coda_hash_uniform(a.seed_lo, a.seed_hi, rowkey, (unsigned)v));
            const float key = x - __logf(q);
            if (key > bkey || (key == bkey && v < barg)) { bkey = key; bl = x; barg = v; }
        }
    });
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

// =====================================================================================================================
// bf16 logits with top-p / top-k (with or without temperature): ONE counting pass instead of a maximum pass + two radix passes.
//
// A bf16 row has at most 65 536 distinct values, and equal values carry equal probability mass.  So a COUNT histogram over
// the exact 16-bit order key of every entry (65 536 x 16-bit counters = 128 KB of LDS, one 32-bit LDS atomic per entry, no
// hot bins: a flat row spreads over thousands of keys) holds everything the filters and the statistics need:
//   * the row maximum, the normaliser, the top-p threshold (fixed-point mass of a key = count x (u64)(e^(x' - max) 2^40),
//     the very numbers the generic kernel accumulates entry by entry), the top-k threshold, Z, sum e^d d, top-2, tie counts -
//     all from a compacted (key, count) list of a few thousand entries, no further pass over the row;
//   * the temperature: x' = bf16(x / T) is applied to the DISTINCT VALUES (a few thousand IEEE divisions per row), not to
//     151 936 entries per pass, and never through the three-operation quotient.
// What still walks the row: the index cut inside a tie group at the top-p boundary (one pass of key compares), and - with
// temperature > 0 - the exponential race over the kept entries inside the race window (hash + two logs per entry).
// Order among entries whose quotients collide after the division: by original logit, then by index (the reference's sort is
// unstable: any order among equal keys is its behaviour; equal keys have equal mass, so the kept COUNT is the same).
// Rows the counters cannot hold (a value repeated >= 32 768 times - canned test logits - or more distinct values than the
// list has room for) are marked (x0 = -1) and redone by the generic kernel in a second, predicated launch.
constexpr int HT = 1024;                 // threads per workgroup (one row per workgroup, one workgroup per CU: LDS)
constexpr int HWAVES = HT / 64;
constexpr int HLIST = 6144;              // distinct values kept per row
constexpr int HIST_WORDS = 32768;
constexpr int HIST_LDS_BYTES = HIST_WORDS * 4 + HLIST * 4;

__device__ __forceinline__ unsigned int key16_of_bits(unsigned int b) {            // b: bf16 bits in the low half
    const unsigned int m = (unsigned int)((int)(b << 16) >> 31);                   // all ones for negative values
    return (b ^ (0x8000u | (m & 0x7fffu))) & 0xffffu;                              // larger float <-> larger key
}
__device__ __forceinline__ float value_of_key16(unsigned int k) {
    const unsigned int b = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
    return __uint_as_float(b << 16);
}

template <typename T, typename Op>
__device__ __forceinline__ T hreduce(T v, T* red, int tid, Op op) {               // all HT threads; red: HWAVES slots
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, (T)__shfl_xor(v, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < HWAVES; ++w) r = op(r, red[w]);
    return r;
}
// sum of v over the threads with a HIGHER tid (exclusive suffix), and the total
template <typename T>
__device__ __forceinline__ T hsuffix(T v, T* red, int tid, T& total) {
    const int lane = tid & 63, wave = tid >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = (T)__shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    __syncthreads();
    if (lane == 0) red[wave] = inc;
    __syncthreads();
    T above = 0, all = 0;
#pragma unroll
    for (int w = 0; w < HWAVES; ++w) {
        all += red[w];
        if (w > wave) above += red[w];
    }
    total = all;
    return inc - v + above;
}
// sum of v over the threads with a LOWER tid (exclusive prefix), and the total
template <typename T>
__device__ __forceinline__ T hprefix(T v, T* red, int tid, T& total) {
    const int lane = tid & 63, wave = tid >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = (T)__shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    T below = 0, all = 0;
#pragma unroll
    for (int w = 0; w < HWAVES; ++w) {
        all += red[w];
        if (w < wave) below += red[w];
    }
    total = all;
    return inc - v + below;
}

__global__ __launch_bounds__(HT) void confidence_hist_kernel(ConfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned int hlds[];
    unsigned int* hist = hlds;                       // [HIST_WORDS]: two 16-bit counters per word
    unsigned int* list = hlds + HIST_WORDS;          // [HLIST]: key << 16 | count, ascending key
    __shared__ float red_f[HWAVES];
    __shared__ int red_i[HWAVES];
    __shared__ unsigned long long red_u[HWAVES];
    __shared__ unsigned long long sh_acc, sh_tie_mass;
    __shared__ int sh_ties, sh_cut;

    const int row = blockIdx.x;
    const int n_rows = a.count ? min(a.count[0], a.N) : a.N;
    if (row >= n_rows) return;
    const int tid = threadIdx.x;
    const int V = a.V;
    const bool sampled = a.temperature > 0.f;
    const bool use_p = a.top_p > 0.f && a.top_p < 1.f;
    const bool use_k = a.top_k > 0 && a.top_k < V;
    const char* rp = a.logits + (size_t)row * a.ld_b;
    const int at = a.dst_pos ? a.dst_pos[row] : row;
    auto fmaxl = [](float x, float y) { return fmaxf(x, y); };
    auto imaxl = [](int x, int y) { return max(x, y); };
    auto iminl = [](int x, int y) { return min(x, y); };
    auto iaddl = [](int x, int y) { return x + y; };
    auto faddl = [](float x, float y) { return x + y; };

    // ---- zero the counters
    {
        u32x4_t* h4 = (u32x4_t*)hist;
#pragma unroll
        for (int i = 0; i < HIST_WORDS / 4 / HT; ++i) h4[tid + HT * i] = (u32x4_t){0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // ---- pass A: count every entry by its exact key; greedy runs also carry (largest key, its first index)
    const int nvec = V / 8;                           // the host guarantees V % 8 == 0 and 16-byte aligned rows
    unsigned int best = 0u;
    int bidx = 0x7fffffff;
    auto count2 = [&](unsigned int raw, int v) {
        const unsigned int k0 = key16_of_bits(raw & 0xffffu), k1 = key16_of_bits(raw >> 16);
        atomicAdd(&hist[k0 >> 1], 1u << ((k0 & 1u) * 16));
        atomicAdd(&hist[k1 >> 1], 1u << ((k1 & 1u) * 16));
        if (!sampled) {                               // entries are met in increasing index order: strict > keeps the first
            if (k0 > best) { best = k0; bidx = v; }
            if (k1 > best) { best = k1; bidx = v + 1; }
        }
    };
    {
        int c = tid;
        for (; c + 3 * HT < nvec; c += 4 * HT) {
            u32x4_t raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *(const u32x4_t*)(rp + (size_t)(c + u * HT) * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) count2(raw[u][j], (c + u * HT) * 8 + 2 * j);
        }
        for (; c < nvec; c += HT) {
            const u32x4_t raw = *(const u32x4_t*)(rp + (size_t)c * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) count2(raw[j], c * 8 + 2 * j);
        }
    }
    __syncthreads();

    // ---- compact the non-empty counters into an ascending (key, count) list; a thread owns 64 consecutive keys
    int mine = 0, total_cnt = 0;
    bool big = false;
    const u32x4_t* h4 = (const u32x4_t*)(hist + tid * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32x4_t q = h4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int lo = q[j] & 0xffffu, hi = q[j] >> 16;
            mine += (lo != 0u) + (hi != 0u);
            total_cnt += (int)(lo + hi);
            big |= ((lo | hi) & 0x8000u) != 0u;
        }
    }
    int distinct = 0;
    const int off = hprefix<int>(mine, red_i, tid, distinct);
    const int seen = hreduce<int>(total_cnt, red_i, tid, iaddl);
    const int anybig = hreduce<int>(big ? 1 : 0, red_i, tid, imaxl);
    if (anybig || seen != V || distinct > HLIST) {   // a counter may have wrapped, or the list is too short: generic kernel
        if (tid == 0) a.x0[at] = -1;
        return;
    }
    if (mine) {
        int o = off;
        for (int i = 0; i < 8; ++i) {
            const u32x4_t q = h4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int lo = q[j] & 0xffffu, hi = q[j] >> 16;
                if (lo) list[o++] = ((unsigned int)(tid * 64 + 8 * i + 2 * j) << 16) | lo;
                if (hi) list[o++] = ((unsigned int)(tid * 64 + 8 * i + 2 * j + 1) << 16) | hi;
            }
        }
    }
    __syncthreads();
    const int D = distinct;
    const int per = (D + HT - 1) / HT;               // <= 6 list entries per thread, contiguous, ascending
    const int e0 = min(D, tid * per), e1 = min(D, e0 + per);
    auto xform = [&](float x) -> float { return sampled ? round_bf16(__fdiv_rn(x, a.temperature)) : x; };
    const unsigned int kmax = list[D - 1] >> 16;
    const float mx = xform(value_of_key16(kmax));

    // ---- per-entry fixed-point mass (count x the generic kernel's per-entry value), suffix sums in key order
    constexpr float FIX = 1099511627776.0f;           // 2^40
    unsigned long long my_mass = 0ull;
    int my_cnt = 0;
    for (int e = e0; e < e1; ++e) {
        const unsigned int it = list[e];
        const float x = xform(value_of_key16(it >> 16));
        my_mass += (unsigned long long)(__expf(x - mx) * FIX) * (unsigned long long)(it & 0xffffu);
        my_cnt += (int)(it & 0xffffu);
    }
    unsigned int kp = 0u, kk = 0u;
    int p_cut = 0x7fffffff, keep_ties = 0x7fffffff;
    if (use_p) {
        unsigned long long tot = 0ull;
        const unsigned long long above_t = hsuffix<unsigned long long>(my_mass, red_u, tid, tot);
        const unsigned long long lim = (unsigned long long)((double)a.top_p * (double)tot);
        // lowest key whose strictly-above mass is <= top_p * total, among keys that carry mass (the top key always is)
        int cand = 0x7fffffff;
        {
            unsigned long long run = above_t;
            for (int e = e1 - 1; e >= e0; --e) {
                const unsigned int it = list[e];
                const float x = xform(value_of_key16(it >> 16));
                const unsigned long long pe = (unsigned long long)(__expf(x - mx) * FIX);
                if (pe != 0ull && run <= lim) cand = (int)(it >> 16);
                run += pe * (unsigned long long)(it & 0xffffu);
            }
        }
        kp = (unsigned int)hreduce<int>(cand, red_i, tid, iminl);
        {
            unsigned long long run = above_t;
            for (int e = e1 - 1; e >= e0; --e) {
                const unsigned int it = list[e];
                const float x = xform(value_of_key16(it >> 16));
                const unsigned long long pe = (unsigned long long)(__expf(x - mx) * FIX);
                if ((it >> 16) == kp) { sh_acc = run; sh_ties = (int)(it & 0xffffu); sh_tie_mass = pe * (unsigned long long)(it & 0xffffu); }
                run += pe * (unsigned long long)(it & 0xffffu);
            }
        }
        __syncthreads();
        const unsigned long long acc = sh_acc;
        const int ties = sh_ties;
        const unsigned long long pt = ties > 0 ? sh_tie_mass / (unsigned long long)ties : 0ull;
        keep_ties = ties;
        if (pt > 0ull && lim >= acc) {
            const unsigned long long room = (lim - acc) / pt;
            if (room < (unsigned long long)ties) keep_ties = max(1, min(ties, (int)room + 1));
        } else if (pt > 0ull) {
            keep_ties = 1;
        }
        if (keep_ties < ties) {
            // index of the keep_ties-th entry equal to the threshold key, in index order.  Every WAVE owns a contiguous
            // range of 16-byte pieces and reads it with coalesced loads (lane l takes piece base + l): first the ties per
            // wave are counted, then the one wave that holds the entry asked for walks its range again and finds piece, lane
            // and element with a ballot-free prefix over the lanes.
            const int lane = tid & 63, wave = tid >> 6;
            const int per_wave = (nvec + HWAVES - 1) / HWAVES;
            const int c0 = min(nvec, wave * per_wave), c1 = min(nvec, c0 + per_wave);
            auto ties_in = [&](const u32x4_t raw) -> int {
                int n = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    n += (key16_of_bits(raw[j] & 0xffffu) == kp) + (key16_of_bits(raw[j] >> 16) == kp);
                return n;
            };
            int n = 0;
            for (int c = c0 + lane; c < c1; c += 64) n += ties_in(*(const u32x4_t*)(rp + (size_t)c * 16));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);      // ties of this wave's range
            __syncthreads();
            if (lane == 0) red_i[wave] = n;
            if (tid == 0) sh_cut = 0x7fffffff;
            __syncthreads();
            int before = 0;
            for (int w2 = 0; w2 < wave; ++w2) before += red_i[w2];
            if (before < keep_ties && before + n >= keep_ties) {             // wave-uniform: this wave holds it
                int need = keep_ties - before;                               // 1-based within this wave's range
                for (int base = c0; base < c1 && need > 0; base += 64) {
                    const int c = base + lane;
                    u32x4_t raw = {0u, 0u, 0u, 0u};
                    int m = 0;
                    if (c < c1) { raw = *(const u32x4_t*)(rp + (size_t)c * 16); m = ties_in(raw); }
                    int inc = m;                                             // inclusive prefix over the lanes
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int t = __shfl_up(inc, o, 64);
                        if (lane >= o) inc += t;
                    }
                    const int round_total = __shfl(inc, 63, 64);
                    if (round_total >= need) {
                        if (inc - m < need && inc >= need) {                 // exactly one lane
                            int left = need - (inc - m);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (key16_of_bits(raw[j] & 0xffffu) == kp && --left == 0) sh_cut = c * 8 + 2 * j;
                                if (key16_of_bits(raw[j] >> 16) == kp && --left == 0) sh_cut = c * 8 + 2 * j + 1;
                            }
                        }
                        need = 0;
                    } else {
                        need -= round_total;
                    }
                }
            }
            __syncthreads();
            p_cut = sh_cut;
        }
    }
    if (use_k) {
        // the top_k-th largest entry (duplicates counted) sits in the key whose strictly-above count is < top_k <= that + its own
        int tot = 0;
        const int above_t = hsuffix<int>(my_cnt, red_i, tid, tot);
        int cand = -1;
        int run = above_t;
        for (int e = e1 - 1; e >= e0; --e) {
            const unsigned int it = list[e];
            const int c = (int)(it & 0xffffu);
            if (run < a.top_k && a.top_k <= run + c) cand = (int)(it >> 16);
            run += c;
        }
        kk = (unsigned int)hreduce<int>(cand, red_i, tid, imaxl);
        if (sampled) {
            // top_k_logits keeps every entry EQUAL to the k-th largest (`logits < kth` is what it removes), and it compares
            // the quotients: a lower original value whose quotient rounds to the same bf16 is kept too (T > 1 makes such
            // collisions common)
            const float xk = xform(value_of_key16(kk));
            int low = (int)kk;
            for (int e = e0; e < e1; ++e) {
                const unsigned int k = list[e] >> 16;
                if (k < kk && xform(value_of_key16(k)) == xk) low = min(low, (int)k);
            }
            kk = (unsigned int)hreduce<int>(low, red_i, tid, iminl);
        }
        // top_k_logits runs on the top-p filtered row: if top-p kept fewer than top_k tokens the k-th largest is a removed
        // entry (finfo.min) and top-k removes nothing more
        if (use_p && kk < kp) kk = 0u;
    }

    // ---- statistics of the kept set, from the list
    float z = 0.f, sv = 0.f, m2 = -INFINITY;
    int n_max = 0;
    unsigned int kw = 0xffffu;                        // lowest key inside the race window
    for (int e = e0; e < e1; ++e) {
        const unsigned int it = list[e], k = it >> 16;
        int n = (int)(it & 0xffffu);
        if (use_p) n = k > kp ? n : (k == kp ? min(n, keep_ties) : 0);
        if (use_k && k < kk) n = 0;
        const float x = xform(value_of_key16(k));
        if (x >= mx - CODA_RACE_WINDOW) kw = min(kw, k);
        if (n == 0) continue;
        const float d = x - mx, ex = __expf(d);
        if (ex > 0.f) {
            z += (float)n * ex;
            sv += (float)n * (ex * d);
        }
        if (x == mx) n_max += n;
        else m2 = fmaxf(m2, x);
    }
    z = hreduce<float>(z, red_f, tid, faddl);
    sv = hreduce<float>(sv, red_f, tid, faddl);
    m2 = hreduce<float>(m2, red_f, tid, fmaxl);
    n_max = hreduce<int>(n_max, red_i, tid, iaddl);
    if (n_max > 1) m2 = mx;

    int arg = 0x7fffffff;
    float bl = 0.f;
    if (!sampled) {
        const unsigned int top = (unsigned int)hreduce<int>((int)best, red_i, tid, imaxl);
        arg = hreduce<int>(best == top ? bidx : 0x7fffffff, red_i, tid, iminl);
    } else {
        // ---- pass B: the exponential race over the kept entries inside the race window (in-kernel noise is bounded: an
        // entry more than CODA_RACE_WINDOW below the maximum cannot win; injected noise is unbounded: every kept entry runs)
        kw = (unsigned int)hreduce<int>((int)kw, red_i, tid, iminl);
        const unsigned int klow = a.noise ? 0u : kw;
        // the counters are dead: their 128 KB become the table key -> bf16(x / T), filled for the keys the row holds, so
        // the race divides nothing (151 936 entries, a few thousand distinct values)
        unsigned short* xtab = (unsigned short*)hist;
        for (int e = e0; e < e1; ++e) {
            const unsigned int k = list[e] >> 16;
            xtab[k] = (unsigned short)(__float_as_uint(xform(value_of_key16(k))) >> 16);
        }
        __syncthreads();
        const unsigned int rowkey = (unsigned int)at;
        float bkey = -INFINITY;
        int barg = 0x7fffffff;
        auto race = [&](unsigned int b, int v) {
            const unsigned int k = key16_of_bits(b);
            if (k < klow) return;
            if (use_p && (k < kp || (k == kp && v > p_cut))) return;
            if (use_k && k < kk) return;
            const float x = __uint_as_float((unsigned int)xtab[k] << 16);
            const float q = a.noise ? a.noise[(size_t)row * V + v] : -__logf(coda_hash_uniform(a.seed_lo, a.seed_hi, rowkey, (unsigned)v));
            const float key = x - __logf(q);
            if (key > bkey || (key == bkey && v < barg)) { bkey = key; bl = x; barg = v; }
        };
        int c = tid;
        for (; c + HT < nvec; c += 2 * HT) {
            const u32x4_t r0 = *(const u32x4_t*)(rp + (size_t)c * 16), r1 = *(const u32x4_t*)(rp + (size_t)(c + HT) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { race(r0[j] & 0xffffu, c * 8 + 2 * j); race(r0[j] >> 16, c * 8 + 2 * j + 1); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { race(r1[j] & 0xffffu, (c + HT) * 8 + 2 * j); race(r1[j] >> 16, (c + HT) * 8 + 2 * j + 1); }
        }
        for (; c < nvec; c += HT) {
            const u32x4_t r0 = *(const u32x4_t*)(rp + (size_t)c * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { race(r0[j] & 0xffffu, c * 8 + 2 * j); race(r0[j] >> 16, c * 8 + 2 * j + 1); }
        }
        const float bestk = hreduce<float>(bkey, red_f, tid, fmaxl);
        const int cand = hreduce<int>(bkey == bestk ? barg : 0x7fffffff, red_i, tid, iminl);
        if (!(bkey == bestk && barg == cand)) bl = -INFINITY;
        bl = hreduce<float>(bl, red_f, tid, fmaxl);
        arg = cand;
    }
    if (tid == 0) {
        float c;
        if (a.alg == CODA_ALG_TOPK_MARGIN) c = (1.0f - __expf(m2 - mx)) / z;
        else if (a.alg == CODA_ALG_ENTROPY) c = sv / z - __logf(z);
        else c = sampled ? __expf(bl - mx) / z : 1.0f / z;
        if (a.round_conf) c = round_bf16(c);
        a.conf[at] = c;
        a.x0[at] = arg;
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
    a.inv_temperature = (temperature > 0.f && is_bf16) ? coda_reciprocal_if_exact(temperature) : 0.f;   // 0: true division
    a.noise = noise; a.seed_lo = (unsigned int)seed; a.seed_hi = (unsigned int)(seed >> 32);
    a.round_conf = round_conf_bf16; a.conf = conf; a.x0 = x0;
    a.only_marked = 0;
    const bool filtered = (top_p > 0.f && top_p < 1.f) || (top_k > 0 && top_k < V);
    if (is_bf16 && filtered && V % 8 == 0 && (((uintptr_t)logits) & 15) == 0 && (ld * 2) % 16 == 0) {
        // counting kernel (one pass + list work); rows it cannot hold come back marked and the generic kernel redoes them.
        // (Temperature without a filter stays on the generic kernel: its two passes - maximum, then statistics + race - are
        // fewer than counting + race: 2.6 vs 2.9 ms per 8192 flat rows.)
        if (coda_ensure_dyn_lds((const void*)confidence_hist_kernel, HIST_LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
        hipLaunchKernelGGL(confidence_hist_kernel, dim3(N), dim3(HT), HIST_LDS_BYTES, (hipStream_t)stream, a);
        CODA_CHECK_LAUNCH("confidence_hist");
        a.only_marked = 1;
    }
    hipLaunchKernelGGL(confidence_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
    CODA_CHECK_LAUNCH("confidence_from_logits");
    return CODA_OK;
}
