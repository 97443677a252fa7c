// Fused lm-head GEMM + online softmax + confidence reduction.  The [N_m, 151936] logits of the
// masked rows live only in MFMA accumulators; per row the kernel carries
//   m1, m2   largest / second largest logit (duplicates counted)        -> argmax, topk_margin
//   Z, S     sum e^(l-m1), sum e^(l-m1) (l-m1)                          -> p_max = 1/Z, entropy = S/Z - log Z
//   arg      first index of the maximum                                 -> greedy x0
//   bkey.. for temperature > 0: the exponential race argmax(l - log q)  -> sampled x0 and its logit
// Replaces modeling_coda.py:411 and generation_utils.py:61-100,457,462.
//
// Grid: a fixed number of workgroups; each reads the DEVICE-side candidate count and derives
// (row block, vocab split) from it, so late denoise steps with few masked rows still spread
// the 622 MB weight stream over the whole chip and no host sync is needed to size the launch.
#include "gemm_tile128.h"
#include "coda_abi_internal.h"
#include <mutex>
#include <stdio.h>

using namespace tile128;

namespace {

constexpr int LM_GRID = 1024;            // workgroups per launch
constexpr int PART_FLOATS = 8;           // floats per partial record

struct Stat {
    float m1, m2, Z, S;
    int arg;
    float bkey, bl;
    int barg;
};

__device__ __forceinline__ Stat stat_init() { return Stat{-INFINITY, -INFINITY, 0.f, 0.f, 0x7fffffff, -INFINITY, 0.f, 0x7fffffff}; }

__device__ __forceinline__ Stat stat_merge(const Stat& a, const Stat& b) {
    Stat r;
    r.m1 = fmaxf(a.m1, b.m1);
    const float da = a.Z > 0.f ? a.m1 - r.m1 : 0.f, db = b.Z > 0.f ? b.m1 - r.m1 : 0.f;
    const float sa = a.Z > 0.f ? __expf(da) : 0.f, sb = b.Z > 0.f ? __expf(db) : 0.f;
    r.Z = a.Z * sa + b.Z * sb;
    r.S = sa * (a.S + da * a.Z) + sb * (b.S + db * b.Z);
    r.m2 = fmaxf(fminf(a.m1, b.m1), fmaxf(a.m2, b.m2));
    r.arg = a.m1 > b.m1 ? a.arg : (b.m1 > a.m1 ? b.arg : min(a.arg, b.arg));
    const bool ta = a.bkey > b.bkey || (a.bkey == b.bkey && a.barg <= b.barg);
    r.bkey = ta ? a.bkey : b.bkey;
    r.bl = ta ? a.bl : b.bl;
    r.barg = ta ? a.barg : b.barg;
    return r;
}

__device__ __forceinline__ Stat stat_shfl_xor(const Stat& s, int o) {
    Stat r;
    r.m1 = __shfl_xor(s.m1, o, 64); r.m2 = __shfl_xor(s.m2, o, 64);
    r.Z = __shfl_xor(s.Z, o, 64); r.S = __shfl_xor(s.S, o, 64);
    r.arg = __shfl_xor(s.arg, o, 64);
    r.bkey = __shfl_xor(s.bkey, o, 64); r.bl = __shfl_xor(s.bl, o, 64); r.barg = __shfl_xor(s.barg, o, 64);
    return r;
}

struct LmArgs {
    const char* hidden; long long ldh_b;
    const int* src_rows; const int* dst_pos; const int* count;
    const char* W; long long ldw_b;
    int V, H, alg, max_rows;
    float temperature, inv_temperature;
    const float* noise;
    unsigned int seed_lo, seed_hi;
    float* part;
};

__device__ __forceinline__ void split_geometry(int n, int tiles_total, int& active_rb, int& vsplits) {
    active_rb = (n + BM - 1) / BM;
    vsplits = max(1, min(tiles_total, LM_GRID / max(active_rb, 1)));
}

__global__ __launch_bounds__(256) void lmhead_partial_kernel(LmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n = min(a.count[0], a.max_rows);       // the records and conf / x0 are sized for max_rows: a larger device-side
    if (n <= 0) return;                              // count must not write past them (confidence_kernel clamps the same way)
    const int tiles_total = (a.V + BN - 1) / BN;
    int active_rb, vsplits;
    split_geometry(n, tiles_total, active_rb, vsplits);
    const bool sampled = a.temperature > 0.f;
    const int tok = lane & 15, fq = (lane >> 4) * 4;

    // Units that share a vocab split (same weight rows, different row blocks) are consecutive; the
    // XCD remap keeps them on one XCD so the weight tile is fetched into that L2 once, not once per
    // row block.  Row blocks beyond LM_GRID (huge candidate counts) are walked by striding.
    for (int unit = xcd_remap(blockIdx.x, LM_GRID); unit < active_rb * vsplits; unit += LM_GRID) {
        const int rb = unit % active_rb, vs = unit / active_rb;
        const int t_begin = (int)((long long)vs * tiles_total / vsplits);
        const int t_end = (int)((long long)(vs + 1) * tiles_total / vsplits);

        Frag f;
        const int ch = dma_chunk(lane) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cand = min(rb * BM + dma_row(i, wave, lane), n - 1);
            f.x[i] = a.hidden + (size_t)a.src_rows[cand] * a.ldh_b + ch;
        }
        f.xlive = dma_live_mask(rb * BM, n, __builtin_amdgcn_readfirstlane(wave));
        Stat st[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) st[mt] = stat_init();

        for (int vt = t_begin; vt < t_end; ++vt) {
            const int n0 = vt * BN;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gn = min(n0 + dma_row(i, wave, lane), a.V - 1);
                f.w[i] = a.W + (size_t)gn * a.ldw_b + ch;
            }
            f32x4_t acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            mainloop(f, a.H, lds, wave, lane, acc);

#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (rb * BM + wm * 64 + mt * 16 >= n) continue;      // token group beyond the candidate list (wave-uniform)
                Stat s = st[mt];
                const int cand = rb * BM + wm * 64 + mt * 16 + tok;
                float l[16];
                const float old_m1 = s.m1;
                // pass 1: logits (rounded as the reference's bf16 lm_head output, then /T in bf16), top-2, argmax
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int v = n0 + wn * 64 + nt * 16 + fq + r;
                        float x = round_bf16(acc[nt][mt][r]);
                        if (sampled) x = round_bf16(coda_div_by(x, a.temperature, a.inv_temperature));
                        if (v >= a.V) x = -INFINITY;
                        l[nt * 4 + r] = x;
                        if (x > s.m1) { s.m2 = s.m1; s.m1 = x; s.arg = v; }
                        else if (x > s.m2) s.m2 = x;
                    }
                // pass 2: rescale the running sums to the new maximum, then accumulate
                if (s.m1 > old_m1 && s.Z > 0.f) {
                    const float d = old_m1 - s.m1, sc = __expf(d);
                    s.S = sc * (s.S + d * s.Z);
                    s.Z *= sc;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (l[i] > -INFINITY) {
                        const float d = l[i] - s.m1, e = __expf(d);
                        s.Z += e;
                        s.S += e * d;
                    }
                }
                if (sampled) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int v = n0 + wn * 64 + nt * 16 + fq + r;
                            // s.m1 = the row's running maximum (this tile included): a lower bound of the final one
                            if (v < a.V && cand < n && (a.noise || l[nt * 4 + r] >= s.m1 - CODA_RACE_WINDOW)) {
                                float q;
                                if (a.noise) q = a.noise[(size_t)cand * a.V + v];
                                else q = -__logf(coda_hash_uniform(a.seed_lo, a.seed_hi, (unsigned)a.dst_pos[cand], (unsigned)v));
                                const float key = l[nt * 4 + r] - __logf(q);
                                if (key > s.bkey) { s.bkey = key; s.bl = l[nt * 4 + r]; s.barg = v; }
                            }
                        }
                }
                st[mt] = s;
            }
        }

        // merge the 4 feature quarters held by different lanes, then the two feature halves (wn)
        Stat* xch = (Stat*)lds;      // [2 wm][64 rows]; LDS is free after mainloop's trailing barrier
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            Stat s = st[mt];
            s = stat_merge(s, stat_shfl_xor(s, 16));
            s = stat_merge(s, stat_shfl_xor(s, 32));
            st[mt] = s;
            if (wn == 1 && lane < 16) xch[wm * 64 + mt * 16 + tok] = s;
        }
        __syncthreads();
        if (wn == 0 && lane < 16) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int cand = rb * BM + wm * 64 + mt * 16 + tok;
                if (cand < n) {
                    const Stat s = stat_merge(st[mt], xch[wm * 64 + mt * 16 + tok]);
                    float* p = a.part + ((size_t)vs * active_rb * BM + cand) * PART_FLOATS;
                    p[0] = s.m1; p[1] = s.m2; p[2] = s.Z; p[3] = s.S; p[4] = __int_as_float(s.arg);
                    p[5] = s.bkey; p[6] = s.bl; p[7] = __int_as_float(s.barg);
                }
            }
        }
        __syncthreads();
    }
}

// One wave per candidate row: lanes stride over the vocab splits, then a shuffle tree merges them.
__global__ __launch_bounds__(256) void lmhead_combine_kernel(const float* __restrict__ part, const int* __restrict__ count,
                                                             const int* __restrict__ dst_pos, int max_rows, int V, int alg,
                                                             int sampled, int round_bf16_conf, float* __restrict__ conf,
                                                             long long* __restrict__ x0) {
    const int n = min(count[0], max_rows);           // same clamp as lmhead_partial_kernel: same split geometry
    const int tiles_total = (V + BN - 1) / BN;
    int active_rb, vsplits;
    split_geometry(n, tiles_total, active_rb, vsplits);
    const int lane = threadIdx.x & 63;
    const int waves = gridDim.x * (blockDim.x >> 6);
    for (int cand = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); cand < n; cand += waves) {
        Stat s = stat_init();
        for (int vs = lane; vs < vsplits; vs += 64) {
            const f32x4_t* p = (const f32x4_t*)(part + ((size_t)vs * active_rb * BM + cand) * PART_FLOATS);
            const f32x4_t a = p[0], b = p[1];
            Stat t{a[0], a[1], a[2], a[3], __float_as_int(b[0]), b[1], b[2], __float_as_int(b[3])};
            s = stat_merge(s, t);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = stat_merge(s, stat_shfl_xor(s, o));
        if (lane == 0) {
            float c;
            const float pmax = sampled ? __expf(s.bl - s.m1) / s.Z : 1.0f / s.Z;
            if (alg == CODA_ALG_TOPK_MARGIN) c = (1.0f - __expf(s.m2 - s.m1)) / s.Z;
            else if (alg == CODA_ALG_ENTROPY) c = s.S / s.Z - __logf(s.Z);
            else c = pmax;
            if (round_bf16_conf) c = round_bf16(c);
            const int at = dst_pos[cand];
            conf[at] = c;
            x0[at] = sampled ? s.barg : s.arg;
        }
    }
}

}  // namespace

extern "C" long long coda_lmhead_workspace_bytes(int max_rows, int V) {
    (void)max_rows; (void)V;
    // live partials never exceed LM_GRID units x 128 rows (the split adapts to the candidate count);
    // when row blocks alone exceed LM_GRID the split count is 1 and max_rows records are needed.
    long long recs = (long long)LM_GRID * BM;
    const long long padded = ((long long)max_rows + BM - 1) / BM * BM;
    if (padded > recs) recs = padded;
    return recs * PART_FLOATS * 4 + 256;
}

// Both entry points: argument checks, then the two launches with `part` = where the partial records go.
static int lmhead_launch(const void* hidden, long long ld_hidden, const int* src_rows, const int* dst_pos, const int* count,
                         int max_rows, const void* W, long long ldw, int V, int H, int alg, float temperature,
                         const float* noise, unsigned long long seed, int round_conf_bf16, float* conf, long long* x0,
                         void* part, void* stream) {
    if (max_rows <= 0 || V <= 0 || H <= 0 || H % BK) return CODA_ERR_SHAPE;
    if (alg < 0 || alg > 2) return CODA_ERR_ARG;
    if (!hidden || !src_rows || !dst_pos || !count || !W || !conf || !x0 || !part) return CODA_ERR_ARG;
    if (((uintptr_t)hidden | (uintptr_t)W | (uintptr_t)part) & 15 || (ld_hidden * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    if (coda_ensure_dyn_lds((const void*)lmhead_partial_kernel, LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    LmArgs a;
    a.hidden = (const char*)hidden; a.ldh_b = ld_hidden * 2;
    a.src_rows = src_rows; a.dst_pos = dst_pos; a.count = count;
    a.W = (const char*)W; a.ldw_b = ldw * 2;
    a.V = V; a.H = H; a.alg = alg; a.max_rows = max_rows; a.temperature = temperature; a.inv_temperature = temperature > 0.f ? coda_reciprocal_if_exact(temperature) : 0.f;   // 0: true division
    a.noise = noise;
    a.seed_lo = (unsigned int)seed; a.seed_hi = (unsigned int)(seed >> 32);
    a.part = (float*)part;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lmhead_partial_kernel, dim3(LM_GRID), dim3(256), LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("lmhead_partial");
    const int cgrid = max(1, min(2048, (max_rows + 3) / 4));
    hipLaunchKernelGGL(lmhead_combine_kernel, dim3(cgrid), dim3(256), 0, s, (const float*)part, count, dst_pos, max_rows, V,
                       alg, temperature > 0.f ? 1 : 0, round_conf_bf16, conf, x0);
    CODA_CHECK_LAUNCH("lmhead_combine");
    return CODA_OK;
}

extern "C" int coda_lmhead_confidence(const void* hidden, long long ld_hidden, const int* src_rows, const int* dst_pos,
                                      const int* count, int max_rows, const void* W, long long ldw, int V, int H, int alg,
                                      float temperature, const float* noise, unsigned long long seed, int round_conf_bf16,
                                      float* conf, long long* x0, void* workspace, void* stream) {
    return lmhead_launch(hidden, ld_hidden, src_rows, dst_pos, count, max_rows, W, ldw, V, H, alg, temperature, noise, seed,
                         round_conf_bf16, conf, x0, workspace, stream);
}

// ---------------------------------------------------------------------------------------------
// The one entry point that allocates nothing and takes no workspace (callers under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc).  The partial-statistics records go to a file-scope device array instead of
// a caller workspace: LM_GRID x 128 records of 32 bytes = 4 MiB (+256) per GPU for the life of the process, enough for
// CODA_LMHEAD_NOALLOC_MAX_ROWS candidate rows (beyond LM_GRID row blocks the vocabulary is no longer split and one
// record per row is needed - coda_lmhead_workspace_bytes).
//
// `lm_scratch` is a HOST-SIDE STAND-IN in host code, not a device address: it is never passed to a kernel.  Its device
// address is asked for once per device with hipGetSymbolAddress (with that device current - every GPU holds its own
// copy at its own address) and kept; kernels, as in the entry point above, take the pointer as an argument.
// Nothing in the records must start at zero: lmhead_combine_kernel reads exactly the records (vocab split < vsplits,
// candidate < count) that lmhead_partial_kernel of the same call wrote.  All calls on one GPU share the array: calls that
// overlap on one GPU (two streams, two threads) race on it - serialise them, or use coda_lmhead_confidence with
// workspaces of your own.
static __device__ __align__(256) unsigned char lm_scratch[(size_t)LM_GRID * BM * PART_FLOATS * 4 + 256];
static_assert(CODA_LMHEAD_NOALLOC_MAX_ROWS == LM_GRID * BM, "header constant out of date");

namespace {
constexpr int LM_MAX_DEV = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[LM_MAX_DEV] = {};     // per device: device address of that GPU's lm_scratch
}  // namespace

static void* lm_scratch_for_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LM_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(lm_scratch));
        if (e != hipSuccess || !p) {
            coda_set_last_error("hipGetSymbolAddress(lm_scratch)", e);
            return nullptr;
        }
        g_scratch_ptr[dev] = p;
    }
    return g_scratch_ptr[dev];
}

extern "C" int coda_lmhead_confidence_noalloc(const void* hidden, long long ld_hidden, const int* src_rows,
                                              const int* dst_pos, const int* count, int max_rows, const void* W,
                                              long long ldw, int V, int H, int alg, float temperature, const float* noise,
                                              unsigned long long seed, int round_conf_bf16, float* conf, long long* x0,
                                              void* stream) {
    if (max_rows > CODA_LMHEAD_NOALLOC_MAX_ROWS) {
        char msg[200];
        snprintf(msg, sizeof(msg), "coda_lmhead_confidence_noalloc: max_rows %d exceeds the static scratch's %d candidate rows "
                 "(use coda_lmhead_confidence with a caller workspace)", max_rows, CODA_LMHEAD_NOALLOC_MAX_ROWS);
        coda_set_last_error_msg(msg);
        return CODA_ERR_SHAPE;
    }
    void* part = lm_scratch_for_current_device();
    if (!part) return CODA_ERR_LAUNCH;
    return lmhead_launch(hidden, ld_hidden, src_rows, dst_pos, count, max_rows, W, ldw, V, H, alg, temperature, noise, seed,
                         round_conf_bf16, conf, x0, part, stream);
}
