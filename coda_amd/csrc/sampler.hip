// Integer / selection side of one denoise step: candidate list, schedule, top-k selection and the
// in-place token scatter.  Everything stays on the device (the reference needs >= 3 host syncs
// per step: nonzero() behind the boolean gather, int(k), torch.any).
#include "coda_abi_internal.h"

// ------------------------------------------------------------------ candidate list
// generation_utils.py:455-462.  Single workgroup (ordered compaction; the list is KB-sized).
__global__ __launch_bounds__(1024) void build_candidates_kernel(const long long* __restrict__ x, int B, int L, int lo, int hi,
                                                                long long mask_id, int* __restrict__ src_rows,
                                                                int* __restrict__ dst_pos, int* __restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = hi - lo;
    const long long total = (long long)B * nw;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (long long start = 0; start < total; start += 1024) {
        const long long e = start + tid;
        int b = 0, p = 0;
        bool flag = false;
        if (e < total) {
            b = (int)(e / nw);
            p = lo + (int)(e % nw);
            flag = x[(size_t)b * L + p] == mask_id;
        }
        const unsigned long long bal = __ballot(flag);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (flag) {
            src_rows[off + pre] = b * L + (p > 0 ? p - 1 : 0);
            dst_pos[off + pre] = b * L + p;
        }
        __syncthreads();
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < 16; ++w) s += wave_tot[w];
            base_s += s;
        }
        __syncthreads();
    }
    if (tid == 0) count[0] = base_s;
}

extern "C" int coda_build_candidates(const long long* x, int B, int L, int win_lo, int win_hi, long long mask_id,
                                     int* src_rows, int* dst_pos, int* count, void* stream) {
    if (B <= 0 || L <= 0 || win_lo < 0 || win_hi > L || win_lo >= win_hi) return CODA_ERR_SHAPE;
    if ((long long)B * L > 2147483647LL) return CODA_ERR_SHAPE;
    if (!x || !src_rows || !dst_pos || !count) return CODA_ERR_ARG;
    hipLaunchKernelGGL(build_candidates_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, B, L, win_lo, win_hi, mask_id,
                       src_rows, dst_pos, count);
    CODA_CHECK_LAUNCH("build_candidates");
    return CODA_OK;
}

// ------------------------------------------------------------------ schedule + top-k + scatter
// generation_utils.py:507-538.  One workgroup per sequence.  The candidates of the row are
// compacted into LDS in position order, keyed, then ranked by an all-pairs count (each wave
// broadcasts key j while every lane holds its own candidate): exact, deterministic, ties go to
// the lowest position.  Window sizes are <= a few thousand, so the O(n^2/256) ranking is us-scale.
__global__ __launch_bounds__(256) void select_unmask_kernel(long long* __restrict__ x, const float* __restrict__ conf,
                                                            const long long* __restrict__ x0, const int* __restrict__ count,
                                                            int B, int L, int lo, int hi, long long mask_id, float ratio,
                                                            int take_all, float alg_temp, const float* __restrict__ noise,
                                                            unsigned int seed_lo, unsigned int seed_hi,
                                                            int* __restrict__ k_out, int k_fixed, float threshold,
                                                            int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nw = hi - lo;
    float* key = (float*)smem;                 // [nw]
    int* cpos = (int*)(smem + (size_t)nw * 4); // [nw]
    __shared__ int wave_tot[4];
    __shared__ float red[4];
    __shared__ int n_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;

    // k exactly as the reference: fp32 (sum / B) * (1 - s/t), truncated toward zero
    const float n_mask = k_fixed >= 0 ? 0.f : __fdiv_rn((float)count[0], (float)B);
    // k_fixed >= 0 (confidence_threshold alg): the host supplies the count; of the k best, the best is always
    // revealed and the others only when their confidence reaches `threshold`; info = {revealed, held back}
    const int k = k_fixed >= 0 ? k_fixed : (take_all ? (int)n_mask : (int)__fmul_rn(n_mask, ratio));
    if (b == 0 && tid == 0 && k_out) k_out[0] = k;
    if (b == 0 && tid == 0 && info) info[0] = info[1] = 0;
    if (k <= 0) return;

    // ordered compaction of this row's masked window positions
    if (tid == 0) n_s = 0;
    __syncthreads();
    for (int start = 0; start < nw; start += 256) {
        const int e = start + tid;
        const int p = lo + e;
        const bool flag = e < nw && x[(size_t)b * L + p] == mask_id;
        const unsigned long long bal = __ballot(flag);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int off = n_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (flag) {
            cpos[off + pre] = p;
            key[off + pre] = conf[(size_t)b * L + p];
        }
        __syncthreads();
        if (tid == 0) n_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    const int n = n_s;
    if (n == 0) return;
    int revealed = 0;

    if (alg_temp > 0.f) {
        // key = softmax(conf / alg_temp) / Exp(1) noise  (== multinomial without replacement)
        float mx = -INFINITY;
        for (int i = tid; i < n; i += 256) mx = fmaxf(mx, key[i] / alg_temp);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float z = 0.f;
        for (int i = tid; i < n; i += 256) z += expf(key[i] / alg_temp - mx);
        z = wave_sum(z);
        if (lane == 0) red[wave] = z;
        __syncthreads();
        z = red[0] + red[1] + red[2] + red[3];
        for (int i = tid; i < n; i += 256) {
            const float pr = expf(key[i] / alg_temp - mx) / z;
            // Exp(1) variate of position (b, p): injected, or -log(U) from the counter hash (index 0xFFFFFFFF: the
            // vocabulary ids the token noise uses never reach it)
            const float q = noise ? noise[(size_t)b * L + cpos[i]]
                                  : -__logf(coda_hash_uniform(seed_lo, seed_hi, (unsigned)(b * L + cpos[i]), 0xFFFFFFFFu));
            key[i] = pr / q;
        }
        __syncthreads();
    }

    for (int i = tid; i < n; i += 256) {
        bool take = true;
        if (k < n || k_fixed >= 0) {
            const float ki = key[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const float kj = key[j];
                rank += (kj > ki) || (kj == ki && j < i);
            }
            take = rank < k;
            if (k_fixed >= 0) take = take && (rank == 0 || !(ki < threshold));
        }
        if (take) {
            const size_t at = (size_t)b * L + cpos[i];
            x[at] = x0[at];
            ++revealed;
        }
    }
    if (info) {
        __syncthreads();
        revealed = (int)wave_sum((float)revealed);      // <= 16384 candidates: exact in fp32
        if (lane == 0) wave_tot[wave] = revealed;
        __syncthreads();
        if (tid == 0) {
            const int tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
            info[0] = tot;
            info[1] = min(k, n) - tot;
        }
    }
}

extern "C" int coda_select_unmask(long long* x, const float* conf, const long long* x0, const int* count, int B, int L,
                                  int win_lo, int win_hi, long long mask_id, float ratio, int take_all, float alg_temp,
                                  const float* noise, unsigned long long seed, int* k_out, void* stream) {
    if (B <= 0 || L <= 0 || win_lo < 0 || win_hi > L || win_lo >= win_hi) return CODA_ERR_SHAPE;
    if (win_hi - win_lo > 16384) return CODA_ERR_SHAPE;      // LDS budget: 8 B per window position
    if ((long long)B * L > 2147483647LL) return CODA_ERR_SHAPE;
    if (!x || !conf || !x0 || !count) return CODA_ERR_ARG;
    const size_t lds = (size_t)(win_hi - win_lo) * 8;
    if (coda_ensure_dyn_lds((const void*)select_unmask_kernel, 16384 * 8) != CODA_OK) return CODA_ERR_LAUNCH;
    hipLaunchKernelGGL(select_unmask_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, conf, x0, count, B, L, win_lo,
                       win_hi, mask_id, ratio, take_all, alg_temp, noise, (unsigned int)seed, (unsigned int)(seed >> 32), k_out,
                       -1, 0.f, (int*)nullptr);
    CODA_CHECK_LAUNCH("select_unmask");
    return CODA_OK;
}

// serving/dream/model/generation_utils.py:442-465 (`confidence_threshold`), one sequence
extern "C" int coda_select_threshold(long long* x, const float* conf, const long long* x0, int L, int win_lo, int win_hi,
                                     long long mask_id, int k, float threshold, int* info, void* stream) {
    if (L <= 0 || win_lo < 0 || win_hi > L || win_lo >= win_hi || k < 0) return CODA_ERR_SHAPE;
    if (win_hi - win_lo > 16384) return CODA_ERR_SHAPE;
    if (!x || !conf || !x0 || !info) return CODA_ERR_ARG;
    const size_t lds = (size_t)(win_hi - win_lo) * 8;
    if (coda_ensure_dyn_lds((const void*)select_unmask_kernel, 16384 * 8) != CODA_OK) return CODA_ERR_LAUNCH;
    hipLaunchKernelGGL(select_unmask_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, x, conf, x0, (const int*)nullptr, 1, L,
                       win_lo, win_hi, mask_id, 0.f, 0, 0.f, (const float*)nullptr, 0u, 0u, (int*)nullptr, k, threshold, info);
    CODA_CHECK_LAUNCH("select_threshold");
    return CODA_OK;
}

// ------------------------------------------------------------------ `origin` transfer
// generation_utils.py:466-483
__global__ void origin_transfer_kernel(long long* __restrict__ x, const long long* __restrict__ x0, const int* __restrict__ dst_pos,
                                       const int* __restrict__ count, const float* __restrict__ u, unsigned int seed_lo,
                                       unsigned int seed_hi, float p_transfer) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count[0]) return;
    const int at = dst_pos[i];
    // uniform of candidate i: injected, or the counter hash of its flat position (index 0xFFFFFFFE)
    const float ui = u ? u[i] : coda_hash_uniform(seed_lo, seed_hi, (unsigned)at, 0xFFFFFFFEu);
    if (ui < p_transfer) x[at] = x0[at];
}

extern "C" int coda_origin_transfer(long long* x, const long long* x0, const int* dst_pos, const int* count, int max_rows,
                                    const float* u, unsigned long long seed, float p_transfer, void* stream) {
    if (max_rows <= 0) return CODA_ERR_SHAPE;
    if (!x || !x0 || !dst_pos || !count) return CODA_ERR_ARG;
    hipLaunchKernelGGL(origin_transfer_kernel, dim3((max_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, x0, dst_pos,
                       count, u, (unsigned int)seed, (unsigned int)(seed >> 32), p_transfer);
    CODA_CHECK_LAUNCH("origin_transfer");
    return CODA_OK;
}
