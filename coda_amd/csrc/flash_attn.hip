// Non-causal, mask-free GQA flash attention forward for head_dim 128 on CDNA4 MFMA.
// Replaces attention.py:12-90 of the reference (repeat_kv + SDPA(is_causal=False), no mask: SURVEY F3).
//
// Work split: one workgroup = 4 waves; it walks query blocks qs, qs + qsplit, ... (128 queries each, 32 per
// wave) of one (batch, q head); from L = 512 on, 8 waves = the two q heads of a GQA pair on one K/V stream (PAIR).  K/V tiles of 64 keys are streamed through a double-buffered LDS ring by
// LDS-DMA; the ring and the Q prefetch run CONTINUOUSLY across query blocks, so the fixed cost of a workgroup
// (launch, first Q/K/V fetch: ~10 us, 44 % of the time at L = 384 with one block per workgroup) is paid once.
//
// MFMA orientation (v_mfma_f32_32x32x16_bf16 everywhere):
//   S^T[key][query] = K Q^T   A = K rows from LDS (ds_read_b128), B = Q^T held in registers.
//                             A lane then holds one query (lane&31) and 16 of the tile's keys per
//                             32-key sub-tile, so the softmax row reduction is in-lane + one
//                             cross-half shuffle, and the accumulator registers are directly the
//                             B operand of the second product (no LDS round trip for P).
//   O^T[d][query]  += V^T P^T A = V^T read with ds_read_b64_tr_b16 (hardware transpose),
//                             B = bf16(P) from the S^T accumulator.  O^T keeps the query on the
//                             lane, so the online-softmax rescale is a per-lane scalar.
// LDS image for both K and V tiles: 64 rows x 256 B, 16-B chunk ch of row r stored at chunk
// ch ^ (((r&3)<<2) | ((r>>2)&3)): conflict-free for the b128 row reads and the transposed reads.
#include "coda_abi_internal.h"
#include <type_traits>

namespace {

constexpr int QB = 128;      // queries per workgroup
constexpr int KB = 64;       // keys per tile
constexpr int TILE = KB * 256;                 // 16 KiB
constexpr int attn_lds(int ns) { return 2 * ns * TILE; }   // K[ns] + V[ns]
constexpr float RESCALE_THR = 6.0f;            // log2 units: P <= 64 between rescales

struct AttnArgs {
    const char* qkv; long long ld_b;
    char* out; long long ldo_b;
    int B, L, n_q, n_kv;
    int qsplit;                  // workgroups per (batch, head): block qs handles query blocks qs, qs + qsplit, ...
    float scale_log2e;
#ifdef CODA_CLOCK_PROBE
    unsigned long long* probe;   // diagnostic build only: per-workgroup (d s_memtime, d s_memrealtime)
#endif
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// PAIR: one workgroup = 8 waves = TWO query heads of one K/V head (GQA): waves 0-3 head 2j, waves 4-7 head 2j+1, same
// query block, ONE K/V stream through LDS for both (half the LDS-DMA and L2 traffic per query).
// NS: stages of the K/V ring (prefetch distance NS - 1 tiles).  4 waves: 2 stages (64 KiB, two workgroups per CU);
// PAIR has the CU to itself and takes a deeper ring.
template <bool PAIR, int NS>
__global__ __launch_bounds__(PAIR ? 512 : 256, PAIR ? 1 : 2) void flash_attn_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6;
    const int wave = wave_all & 3;            // position inside the head's 128-query block
    const int c = lane & 31, h = lane >> 5;
    // XCD-aware linear order: the (n_q / n_kv) query heads x query blocks that read one K/V head are consecutive
    // logical ids, so they run on ONE XCD at about the same time and the K/V head is fetched into that L2 once
    // (round-robin dispatch would otherwise spread them over 6 XCDs = 6 HBM/Infinity-Cache reads of every K/V byte)
    const int nqb = (a.L + QB - 1) / QB;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int heads_per_wg = PAIR ? 2 : 1, n_hg = a.n_q / heads_per_wg;
    const int qs = id % a.qsplit, head = ((id / a.qsplit) % n_hg) * heads_per_wg + (PAIR ? (wave_all >> 2) : 0), b = id / (a.qsplit * n_hg);
    const int kvh = head / (a.n_q / a.n_kv);
    const int L = a.L;
    const char* base = a.qkv + (size_t)b * L * a.ld_b;
    const char* kbase = base + (size_t)(a.n_q + kvh) * 256;
    const char* vbase = base + (size_t)(a.n_q + a.n_kv + kvh) * 256;

    // ---- Q^T fragments: lane (query c, half h), k-step ks: Q[q][16 ks + 8 h .. +7]
    bf16x8_t qf[8];
    auto load_q = [&](int qblk, bf16x8_t (&q)[8]) {
        const int q_ld = min(qblk * QB + wave * 32 + c, L - 1);
        const char* qp = base + (size_t)q_ld * a.ld_b + (size_t)head * 256 + h * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) q[ks] = *(const bf16x8_t*)(qp + ks * 32);
    };
    load_q(qs, qf);

    // ---- LDS-DMA source addressing: instruction i of this wave fills rows (i*4 + wave)*4 .. +3
    const int dma_r = lane >> 4;                                   // row within the 4-row piece
    const int dma_src_chunk = (lane & 15) ^ ((dma_r << 2) | wave); // swz(row) for row = (i*4+wave)*4 + dma_r
    auto stage = [&](int t, int buf) {
        char* kd = lds + buf * TILE;
        char* vd = lds + (NS + buf) * TILE;
        constexpr int NW = PAIR ? 8 : 4;              // waves sharing the 16 pieces of a tile
#pragma unroll
        for (int i = 0; i < 16 / NW; ++i) {
            const int piece = i * NW + wave_all;        // piece & 3 == wave: the swizzle term of dma_src_chunk
            const int r = piece * 4 + dma_r;
            const int key = min(t * KB + r, L - 1);
            const size_t go = (size_t)key * a.ld_b + dma_src_chunk * 16;
            glds16(kbase + go, kd + piece * 1024);
            glds16(vbase + go, vd + piece * 1024);
        }
    };

    f32x16_t oacc[4];
    float m_run, l_run;

    // per-lane LDS read offsets
    //  K row read (A operand of S^T): row = 32*kt2 + c, chunk = 2*ks + h
    //  V transposed read: lane i16 = lane&15 -> (q = i16>>2, p = i16&3); row = key0 + q,
    //  chunk = 4*db + 2*((lane>>4)&1) + (p>>1), byte 8*(p&1)
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int v_chunk0 = 2 * ((lane >> 4) & 1) + (tp >> 1);
    const int v_byte = 8 * (tp & 1);

    // loop-invariant LDS byte offsets (relative to the K / V tile base)
    //  K fragment (k2, ks): row 32 k2 + c, 16-B chunk 2 ks + h
    int k_off[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) k_off[k2] = (32 * k2 + c) * 256;
    const int k_sw = swz(c);                      // swz(32 k2 + c) == swz(c): only row bits 0..3 enter
    //  V fragment (st, jj): rows 16 st + 8 jj + 4 h + tq; chunk 4 db + v_chunk0
    int v_off[4][2], v_sw[4][2];
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = 16 * st + 8 * jj + 4 * h + tq;
            v_off[st][jj] = row * 256 + v_byte;
            v_sw[st][jj] = swz(row);
        }
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;

    auto lgkm0 = [] {
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_k = [&](const char* kt, int k2, int half, bf16x8_t (&kf)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = *(const bf16x8_t*)(kt + k_off[k2] + (((2 * (4 * half + j) + h) ^ k_sw) << 4));
    };
    auto load_v = [&](const char* vt, int st, bf16x8_t (&vf)[4]) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const s16x4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (CODA_LDS s16x4_t*)(vt + v_off[st][0] + (((4 * db + v_chunk0) ^ v_sw[st][0]) << 4)));
            const s16x4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (CODA_LDS s16x4_t*)(vt + v_off[st][1] + (((4 * db + v_chunk0) ^ v_sw[st][1]) << 4)));
            const s16x8_t v8 = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            vf[db] = __builtin_bit_cast(bf16x8_t, v8);
        }
    };

    // One K/V tile.  Fragment reads are issued one batch ahead of the MFMAs that consume them
    // (register double buffers kA/kB, vA/vB); the key mask runs only on a ragged last tile.
    auto tile = [&](int t, int buf, int next_qblk, bool ragged) {
        const char* kt = lds + buf * TILE;
        const char* vt = lds + (NS + buf) * TILE;
        bf16x8_t kA[4], kB[4], vA[4], vB[4];
        f32x16_t s[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[k2][i] = 0.f;

        // ---- S^T = K Q^T : two 32-key sub-tiles, four batches of 4 k-steps
        load_k(kt, 0, 0, kA);
        lgkm0(); load_k(kt, 0, 1, kB);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kA[j], qf[j], s[0], 0, 0, 0);
        lgkm0(); load_k(kt, 1, 0, kA);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kB[j], qf[4 + j], s[0], 0, 0, 0);
        lgkm0(); load_k(kt, 1, 1, kB);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kA[j], qf[j], s[1], 0, 0, 0);
        lgkm0(); load_v(vt, 0, vA);                 // first V batch flies under the last QK MFMAs and the softmax
#pragma unroll
        for (int j = 0; j < 4; ++j) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kB[j], qf[4 + j], s[1], 0, 0, 0);
        // Q of this block is dead from here on: on the block's last tile the next block's Q is fetched into the
        // same registers, under the softmax, the PV product and the epilogue
        __builtin_amdgcn_sched_barrier(0);            // keep the scheduler from hoisting these loads over the QK MFMAs
        if (next_qblk >= 0) load_q(next_qblk, qf);
        __builtin_amdgcn_sched_barrier(0);

        if (ragged) {                                   // uniform branch: only a sequence's last, partial tile
            asm volatile("" ::: "memory");              // not speculatable: keeps the branch from being if-converted
            const int lim = L - t * KB - 4 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // key = t*KB + 4h + (32 k2 + (i&3) + 8 (i>>2)); compare the compile-time part against one
                    // register (keeps 32 per-lane key indices from being hoisted out of the tile loop)
                    if (32 * k2 + (i & 3) + 8 * (i >> 2) >= lim) s[k2][i] = -1e30f;
                }
        }
        // ---- online softmax in the log2 domain (per query = per lane column; the two halves hold disjoint keys).
        // Deferred rescale: the running maximum (and with it the O accumulator) is only moved when some row's
        // maximum grew by more than RESCALE_THR; until then P is bounded by 2^RESCALE_THR instead of 1, which
        // bf16 P and the fp32 sums tolerate.
        float mx = s[0][0];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[k2][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2e;
        if (__any(mx > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // 0 on the first tile (m_run = -1e30)
            l_run *= alpha;
            m_run = m_new;
            if (t > 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;
            }
        }
        float psum = 0.f;
        const float neg_m = -m_run;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[k2][i], a.scale_log2e, neg_m));
                s[k2][i] = p;
                psum += p;
            }
        l_run += psum;

        // ---- O^T += V^T P^T : 4 k-steps of 16 keys x 4 d-blocks of 32
        auto pv = [&](int st, const bf16x8_t (&vf)[4]) {
            const int k2 = st >> 1, r0 = 8 * (st & 1);
            bf16x8_t pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[k2][r0 + j];
#pragma unroll
            for (int db = 0; db < 4; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], pf, oacc[db], 0, 0, 0);
        };
        lgkm0(); load_v(vt, 1, vB); pv(0, vA);
        lgkm0(); load_v(vt, 2, vA); pv(1, vB);
        lgkm0(); load_v(vt, 3, vB); pv(2, vA);
        lgkm0(); pv(3, vB);
    };

#ifdef CODA_CLOCK_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int nt = (L + KB - 1) / KB;
    const int n_full = L / KB;                     // tiles without a ragged edge
    // The ring runs over the FLAT tile sequence of all query blocks this workgroup walks (same K/V head: block after
    // block restarts at key tile 0).  Flat tile f lives in stage f % NS; its DMA is issued NS - 1 tiles ahead.
    int nblk = 0;
    for (int qblk = qs; qblk < nqb; qblk += a.qsplit) ++nblk;
    const int total = nblk * nt;
    constexpr int DMA_OPS = PAIR ? 4 : 8;          // LDS-DMA instructions per wave and stage
    int f = 0;                                     // flat tile index = ring position
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < total) stage(p % nt, p % NS);
    auto wait_dma = [](int n) {                    // s_waitcnt takes an immediate
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    for (int qblk = qs; qblk < nqb; qblk += a.qsplit) {
        const bool more = qblk + a.qsplit < nqb;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
        m_run = -1e30f;
        l_run = 0.f;
        for (int t = 0; t < nt; ++t, ++f) {
            // Tile f's DMA must have landed.  Vector-memory operations retire in issue order, so it is enough to allow
            // as many outstanding as were ISSUED after it: the DMA of the (up to NS - 2) later stages, and - while the
            // previous block's end lies inside that window - 8 more: its output stores, or for a wave that stored
            // nothing the Q prefetch (8 loads, issued before the stores and needed by this tile anyway)
            const int later = min(NS - 2, total - 1 - f);
            wait_dma(DMA_OPS * later + ((qblk != qs && t <= NS - 2) ? 8 : 0));
            __syncthreads();
            if (f + NS - 1 < total) stage((f + NS - 1) % nt, (f + NS - 1) % NS);
            const int nq = (t + 1 == nt && more) ? qblk + a.qsplit : -1;
            tile(t, f % NS, nq, t >= n_full);
        }

        // ---- epilogue: normalise, store bf16.  reg i of d-block db: d = 32 db + (i&3) + 8 (i>>2) + 4 h, i.e. a
        // lane holds 4 consecutive features per group g and its partner lane (other half h) the next 4.  The halves
        // trade groups with v_permlane32_swap so every lane stores 8 consecutive features (16 B): half as many
        // store instructions in the store-issue-bound tail.
        const int q_row = qblk * QB + wave * 32 + c;
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        char* op = a.out + ((size_t)b * L + min(q_row, L - 1)) * a.ldo_b + (size_t)head * 256;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gA = 2 * j, gB = 2 * j + 1;
                const unsigned a0 = pack_bf16x2(oacc[db][4 * gA] * inv, oacc[db][4 * gA + 1] * inv);
                const unsigned a1 = pack_bf16x2(oacc[db][4 * gA + 2] * inv, oacc[db][4 * gA + 3] * inv);
                const unsigned b0 = pack_bf16x2(oacc[db][4 * gB] * inv, oacc[db][4 * gB + 1] * inv);
                const unsigned b1 = pack_bf16x2(oacc[db][4 * gB + 2] * inv, oacc[db][4 * gB + 3] * inv);
                // swap: lanes 32..63 of the first operand <-> lanes 0..31 of the second
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const u32x4_t v = {s0[0], s1[0], s0[1], s1[1]};
                const int d = 32 * db + 8 * (2 * j + h);
                if (q_row < L) *(u32x4_t*)(op + d * 2) = v;
            }
    }
#ifdef CODA_CLOCK_PROBE
    if (a.probe && threadIdx.x == 0) {
        a.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_t0;
        a.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    }
#endif
}

}  // namespace

#ifdef CODA_CLOCK_PROBE
static unsigned long long* g_attn_probe = nullptr;
extern "C" int coda_attn_probe(void* buf) {
    g_attn_probe = (unsigned long long*)buf;
    return CODA_OK;
}
#endif

extern "C" int coda_flash_attn_noncausal_ex(const void* qkv, long long ld_qkv, void* out, long long ld_out, int B, int L,
                                            int n_q, int n_kv, float scale, int pair_mode, int qsplit, void* stream) {
    if (B <= 0 || L <= 0 || n_q <= 0 || n_kv <= 0 || n_q % n_kv) return CODA_ERR_SHAPE;
    if (ld_qkv < (long long)(n_q + 2 * n_kv) * 128 || ld_out < (long long)n_q * 128) return CODA_ERR_SHAPE;
    if ((long long)B * n_q * ((L + QB - 1) / QB) > 0x7fffffffLL) return CODA_ERR_SHAPE;
    if (!qkv || !out) return CODA_ERR_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || (ld_qkv * 2) % 16 || (ld_out * 2) % 16) return CODA_ERR_ALIGN;
    constexpr int NS_PAIR = 2;     // 3 and 4 stages measured: no change at L = 384 .. 2048 (the DMA latency is covered)
    if (coda_ensure_dyn_lds((const void*)flash_attn_kernel<false, 2>, attn_lds(2)) != CODA_OK ||
        coda_ensure_dyn_lds((const void*)flash_attn_kernel<true, NS_PAIR>, attn_lds(NS_PAIR)) != CODA_OK)
        return CODA_ERR_LAUNCH;
    bool pair = (n_q / n_kv) % 2 == 0 && L >= 512;
    if (pair_mode >= 0) pair = pair_mode != 0 && (n_q / n_kv) % 2 == 0;          // explicit per-call choice (A/B, tests)
    AttnArgs a;
    a.qkv = (const char*)qkv; a.ld_b = ld_qkv * 2;
    a.out = (char*)out; a.ldo_b = ld_out * 2;
    a.B = B; a.L = L; a.n_q = n_q; a.n_kv = n_kv;
    a.scale_log2e = scale * 1.4426950408889634f;
    // workgroups per (batch, head): few enough that each amortises its fixed cost over several query blocks, many
    // enough to fill 2 workgroups per CU in whole rounds.  Cost model (us, measured on this kernel): a round costs
    // 9.7 + 1.95 per K/V tile per query block
    const int slots = 2 * coda_device_cus();
    const int wg_slots = pair ? slots / 2 : slots;
    const int n_hg = pair ? n_q / 2 : n_q;
    const int nqb = (L + QB - 1) / QB, ntile = (L + KB - 1) / KB;
    double best = 1e30;
    a.qsplit = 1;
    for (int qsp = 1; qsp <= nqb; ++qsp) {
        const long long wgs = (long long)B * n_hg * qsp;
        const double rounds = (double)((wgs + wg_slots - 1) / wg_slots);
        const double cost = rounds * (9.7 + 1.95 * ntile * ((nqb + qsp - 1) / qsp));
        if (cost < best) { best = cost; a.qsplit = qsp; }
    }
    if (qsplit > 0) a.qsplit = qsplit < nqb ? qsplit : nqb;      // explicit per-call choice (A/B benches)
#ifdef CODA_CLOCK_PROBE
    a.probe = g_attn_probe;
#endif
    if (pair) hipLaunchKernelGGL((flash_attn_kernel<true, NS_PAIR>), dim3(a.qsplit * n_hg * B), dim3(512), attn_lds(NS_PAIR), (hipStream_t)stream, a);
    else hipLaunchKernelGGL((flash_attn_kernel<false, 2>), dim3(a.qsplit * n_q * B), dim3(256), attn_lds(2), (hipStream_t)stream, a);
    CODA_CHECK_LAUNCH("flash_attn");
    return CODA_OK;
}

extern "C" int coda_flash_attn_noncausal(const void* qkv, long long ld_qkv, void* out, long long ld_out, int B, int L,
                                         int n_q, int n_kv, float scale, void* stream) {
    return coda_flash_attn_noncausal_ex(qkv, ld_qkv, out, ld_out, B, L, n_q, n_kv, scale, -1, 0, stream);
}
