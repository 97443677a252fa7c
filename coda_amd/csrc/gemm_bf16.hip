// bf16 projection GEMMs (QKV / O / gate|up / down / plain lm-head rows) on CDNA4 MFMA.
// Replaces the nn.Linear calls of the reference decoder (modeling_coda.py:58-60,123-125,155)
// plus the ops fused behind them: residual add (:247,:253) and SiLU(gate)*up (:59).
#include "gemm_tile128.h"
#include "gemm_tile256.h"
#include "gemm_tile256x128.h"
#include "gemm_tile192.h"
#include "coda_abi_internal.h"

using namespace tile128;

enum { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_F32 = 3, EPI_QKROPE = 4 };

struct GemmArgs {
    const char* X; long long ldx_bytes; const int* x_rows;   // x_rows: optional row gather (nullable)
    const int* m_count;                                      // optional device-side row count: M = min(M, m_count[0])
    const char* W; long long ldw_bytes;
    char* C; long long ldc_bytes;
    const char* R; long long ldr_bytes;
    int M, N, K;
    int tiles_m, tiles_n;
    int wide;                       // C (and R) rows are 16-byte aligned: interior tiles may use 16-byte stores / residual loads
    int ksplit; long long slab_bytes;   // 128x128 engine only: K is cut into ksplit ranges, range s writes fp32 partials to C + s * slab_bytes
#ifdef CODA_CLOCK_PROBE
    unsigned long long* probe;      // diagnostic build only: per-workgroup (d s_memtime, d s_memrealtime)
#endif
    // EPI_QKROPE only: q/k RMSNorm weights, rotary tables [L,128], sequence length, feature bounds of q and q|k
    const char* qn_w; const char* kn_w; const char* rope_cos; const char* rope_sin;
    int L, q_feat, qk_feat; float eps;
};

// silu on the MFMA-free tail of a tile: v_exp + v_rcp (about 1 ulp each, far below the bf16 rounding that follows)
__device__ __forceinline__ float silu_f32(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896f));
}

// Two 8-byte pieces -> one 16-byte piece per lane (guide T21 with v_permlane16_swap).  After the 16x16x32 MFMA a lane
// of lane-row r = lane>>4 holds features 4r..4r+3 of every 16-feature block: 8 bytes, so a row of a block is written
// by four lanes in four-byte-pair pieces and the store tail is bound by the NUMBER of store instructions.  Given the
// lane's pieces a (block A) and b (block B = the next 16 features), the swap trades the odd lane-rows' a with the
// even lane-rows' b: even rows end up with A's features 4r..4r+7, odd rows with B's features 4(r-1)..4(r-1)+7 -
// 16 contiguous bytes each, one store instead of two, same bytes, same addresses.
__device__ __forceinline__ u32x4_t widen16(u32x2_t a, u32x2_t b) {
    const auto s0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    return (u32x4_t){s0[0], s1[0], s0[1], s1[1]};
}
// byte offset of the lane's 16-byte piece inside the 64 bytes of the block pair (A at 0, B at 32)
__device__ __forceinline__ int widen16_offset(int lane) {
    const int r = lane >> 4;
    return 8 * r + ((r & 1) ? 24 : 0);
}

// Epilogue of one wave's accumulator block acc[nt][mt] (4 x MT tiles of 16 features x 16 tokens):
// the lane owns token (lane&15) of each 16-token group and features (lane>>4)*4..+3 of each 16-feature
// group, i.e. 8 contiguous output bytes per (nt, mt).  FULL = the whole workgroup tile is inside the
// matrix (no bounds tests); FULL tiles of 16-byte aligned matrices (a.wide) go out as 16-byte pieces (widen16).
// Rounding points follow the reference's bf16 op sequence: the projection
// output is rounded first, then residual add / silu / gate*up each round again.
// residual values of a wave tile in the widened layout, fetched by the caller ahead of the epilogue (residual_prefetch)
template <int MT>
struct ResidualTile { u32x4_t v[MT][2]; };

// token group mt of the wave tile: the two 16-byte residual pieces the widened RESIDUAL epilogue adds to it
template <int MT>
__device__ __forceinline__ void residual_prefetch(const GemmArgs& a, ResidualTile<MT>& rt, int mt, int mrow0, int ncol0, int lane) {
    const char* r = a.R + (size_t)(mrow0 + (lane & 15) + mt * 16) * a.ldr_bytes + (size_t)ncol0 * 2 + widen16_offset(lane);
    rt.v[mt][0] = *(const u32x4_t*)r;
    rt.v[mt][1] = *(const u32x4_t*)(r + 64);
}

template <int EPI, int MT, bool FULL, bool PRE = false>
__device__ __forceinline__ void epilogue_wave(const GemmArgs& a, const f32x4_t (&acc)[4][MT], int mrow0, int ncol0, int lane,
                                              const ResidualTile<MT>* pre = nullptr) {
    const int tok = lane & 15, fq = (lane >> 4) * 4;
    const int mf = mrow0 + tok;
    const size_t rs = (size_t)16 * a.ldc_bytes;
    if constexpr (EPI == EPI_SWIGLU) {
        // weight rows are interleaved in blocks of 16: [gate 16 | up 16] -> even nt = gate, odd nt = up
        const int nb = ncol0 / 2 + fq;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 2;
        auto swiglu = [&](int mt, int p) -> u32x2_t {
            const unsigned g01 = pack_bf16x2(acc[2 * p][mt][0], acc[2 * p][mt][1]);
            const unsigned g23 = pack_bf16x2(acc[2 * p][mt][2], acc[2 * p][mt][3]);
            const unsigned u01 = pack_bf16x2(acc[2 * p + 1][mt][0], acc[2 * p + 1][mt][1]);
            const unsigned u23 = pack_bf16x2(acc[2 * p + 1][mt][2], acc[2 * p + 1][mt][3]);
            float s0 = silu_f32(bf16_lo(g01)), s1 = silu_f32(bf16_hi(g01));
            float s2 = silu_f32(bf16_lo(g23)), s3 = silu_f32(bf16_hi(g23));
            round_bf16_pair(s0, s1);
            round_bf16_pair(s2, s3);
            return (u32x2_t){pack_bf16x2(s0 * bf16_lo(u01), s1 * bf16_hi(u01)), pack_bf16x2(s2 * bf16_lo(u23), s3 * bf16_hi(u23))};
        };
        if (FULL && a.wide) {
            char* cw = a.C + (size_t)mf * a.ldc_bytes + (size_t)(ncol0 / 2) * 2 + widen16_offset(lane);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) *(u32x4_t*)(cw + mt * rs) = widen16(swiglu(mt, 0), swiglu(mt, 1));
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!FULL && mf + mt * 16 >= a.M) continue;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (!FULL && nb + p * 16 >= a.N / 2) continue;
                *(u32x2_t*)(c + mt * rs + p * 32) = swiglu(mt, p);
            }
        }
    } else if constexpr (EPI == EPI_F32) {
        const int nb = ncol0 + fq;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!FULL && mf + mt * 16 >= a.M) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (!FULL && nb + nt * 16 >= a.N) continue;
                *(f32x4_t*)(c + mt * rs + nt * 64) = acc[nt][mt];
            }
        }
    } else {
        // token groups in batches of up to 4: the batch's residual loads are all issued before its first
        // store (C may alias R: every lane reads exactly the bytes it later writes)
        constexpr int MB = MT % 4 == 0 ? 4 : 3;
        if (FULL && a.wide) {
            // 16-byte pieces: block pairs (nt 0,1) and (nt 2,3); the residual is loaded in the same widened layout
            const int wo = widen16_offset(lane);
            char* cw = a.C + (size_t)mf * a.ldc_bytes + (size_t)ncol0 * 2 + wo;
#pragma unroll
            for (int mb = 0; mb < MT; mb += MB) {
                u32x4_t rv[MB][2];
                if constexpr (EPI == EPI_RESIDUAL && PRE) {
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) rv[i][q] = pre->v[mb + i][q];
                } else if constexpr (EPI == EPI_RESIDUAL) {
                    const char* r = a.R + (size_t)mf * a.ldr_bytes + (size_t)ncol0 * 2 + wo;
                    const size_t rrs = (size_t)16 * a.ldr_bytes;
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) rv[i][q] = *(const u32x4_t*)(r + (mb + i) * rrs + q * 64);
                }
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int mt = mb + i;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const u32x2_t pa = {pack_bf16x2(acc[2 * q][mt][0], acc[2 * q][mt][1]), pack_bf16x2(acc[2 * q][mt][2], acc[2 * q][mt][3])};
                        const u32x2_t pb = {pack_bf16x2(acc[2 * q + 1][mt][0], acc[2 * q + 1][mt][1]),
                                            pack_bf16x2(acc[2 * q + 1][mt][2], acc[2 * q + 1][mt][3])};
                        u32x4_t v = widen16(pa, pb);
                        if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[e] = pack_bf16x2(bf16_lo(v[e]) + bf16_lo(rv[i][q][e]), bf16_hi(v[e]) + bf16_hi(rv[i][q][e]));
                        }
                        *(u32x4_t*)(cw + mt * rs + q * 64) = v;
                    }
                }
            }
            return;
        }
        const int nb = ncol0 + fq;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 2;
#pragma unroll
        for (int mb = 0; mb < MT; mb += MB) {
            u32x2_t rv[MB][4];
            if constexpr (EPI == EPI_RESIDUAL) {
                const char* r = a.R + (size_t)mf * a.ldr_bytes + (size_t)nb * 2;
                const size_t rrs = (size_t)16 * a.ldr_bytes;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        rv[i][nt] = (u32x2_t){0u, 0u};
                        if (FULL || (mf + (mb + i) * 16 < a.M && nb + nt * 16 < a.N)) rv[i][nt] = *(const u32x2_t*)(r + (mb + i) * rrs + nt * 32);
                    }
            }
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int mt = mb + i;
                if (!FULL && mf + mt * 16 >= a.M) continue;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (!FULL && nb + nt * 16 >= a.N) continue;
                    unsigned v0 = pack_bf16x2(acc[nt][mt][0], acc[nt][mt][1]);
                    unsigned v1 = pack_bf16x2(acc[nt][mt][2], acc[nt][mt][3]);
                    if constexpr (EPI == EPI_RESIDUAL) {
                        v0 = pack_bf16x2(bf16_lo(v0) + bf16_lo(rv[i][nt][0]), bf16_hi(v0) + bf16_hi(rv[i][nt][0]));
                        v1 = pack_bf16x2(bf16_lo(v1) + bf16_lo(rv[i][nt][1]), bf16_hi(v1) + bf16_hi(rv[i][nt][1]));
                    }
                    *(u32x2_t*)(c + mt * rs + nt * 32) = (u32x2_t){v0, v1};
                }
            }
        }
    }
}

template <int EPI, int MT>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& a, const f32x4_t (&acc)[4][MT], int mrow0, int ncol0, int lane, bool full) {
    if (full) epilogue_wave<EPI, MT, true>(a, acc, mrow0, ncol0, lane);
    else epilogue_wave<EPI, MT, false>(a, acc, mrow0, ncol0, lane);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware + grouped tile order: consecutive ids within an XCD chunk walk 8 row-tiles of one
    // column-tile group, so the W panel and the X panels stay hot in that XCD's L2.
    const int ntile = a.tiles_m * a.tiles_n;
    const int nwg = ntile * a.ksplit;
    int id = xcd_remap(blockIdx.x, nwg);
    // split-K (small M: too few tiles to engage the chip, and every CU's load path takes only 64 B/clk): K range ks of
    // tile id; the ranges of one tile are neighbours in the id order, so its X / W panels are read on one XCD
    const int ks = id % a.ksplit;
    id /= a.ksplit;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * a.tiles_n;
    const int g = id / per_group;
    const int gm0 = g * GROUP_M;
    const int gsz = min(a.tiles_m - gm0, GROUP_M);
    const int tm = gm0 + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    if (a.m_count) {                          // tile geometry stays the host's upper bound; rows beyond the count do no work
        a.M = min(a.M, a.m_count[0]);
        if (m0 >= a.M) return;
    }
    const int kc = a.K / a.ksplit;            // host guarantees kc % BK == 0
    a.X += (size_t)ks * kc * 2;
    a.W += (size_t)ks * kc * 2;
    a.C += (size_t)ks * a.slab_bytes;
    a.K = kc;

    Frag f;
    const int ch = dma_chunk(lane) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = dma_row(i, wave, lane);
        int gm = min(m0 + r, a.M - 1);
        if (a.x_rows) gm = a.x_rows[gm];
        const int gn = min(n0 + r, a.N - 1);
        f.x[i] = a.X + (size_t)gm * a.ldx_bytes + ch;
        f.w[i] = a.W + (size_t)gn * a.ldw_bytes + ch;
    }
    f.xlive = dma_live_mask(m0, a.M, __builtin_amdgcn_readfirstlane(wave));

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    mainloop(f, a.K, lds, wave, lane, acc);

    epilogue_tile<EPI, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, m0 + BM <= a.M && n0 + BN <= a.N);
}


// Two-block tail of a wave tile whose feature width is not a multiple of 64 (gemm_tile192.h: 96 = 64 + 32 features):
// the same rounding points as epilogue_wave, 8-byte pieces only.
template <int EPI, int MT, bool FULL>
__device__ __forceinline__ void epilogue_tail2(const GemmArgs& a, const f32x4_t (&acc)[2][MT], int mrow0, int ncol0, int lane) {
    const int tok = lane & 15, fq = (lane >> 4) * 4;
    const int mf = mrow0 + tok;
    const size_t rs = (size_t)16 * a.ldc_bytes;
    if constexpr (EPI == EPI_SWIGLU) {
        const int nb = ncol0 / 2 + fq;             // block 0 = 16 gate rows, block 1 = their 16 up rows
        if (!FULL && nb >= a.N / 2) return;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!FULL && mf + mt * 16 >= a.M) continue;
            const unsigned g01 = pack_bf16x2(acc[0][mt][0], acc[0][mt][1]), g23 = pack_bf16x2(acc[0][mt][2], acc[0][mt][3]);
            const unsigned u01 = pack_bf16x2(acc[1][mt][0], acc[1][mt][1]), u23 = pack_bf16x2(acc[1][mt][2], acc[1][mt][3]);
            float s0 = silu_f32(bf16_lo(g01)), s1 = silu_f32(bf16_hi(g01));
            float s2 = silu_f32(bf16_lo(g23)), s3 = silu_f32(bf16_hi(g23));
            round_bf16_pair(s0, s1);
            round_bf16_pair(s2, s3);
            *(u32x2_t*)(c + mt * rs) =
                (u32x2_t){pack_bf16x2(s0 * bf16_lo(u01), s1 * bf16_hi(u01)), pack_bf16x2(s2 * bf16_lo(u23), s3 * bf16_hi(u23))};
        }
    } else if constexpr (EPI == EPI_F32) {
        const int nb = ncol0 + fq;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!FULL && mf + mt * 16 >= a.M) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (!FULL && nb + nt * 16 >= a.N) continue;
                *(f32x4_t*)(c + mt * rs + nt * 64) = acc[nt][mt];
            }
        }
    } else {
        const int nb = ncol0 + fq;
        char* c = a.C + (size_t)mf * a.ldc_bytes + (size_t)nb * 2;
        u32x2_t rv[MT][2];
        if constexpr (EPI == EPI_RESIDUAL) {       // all residual loads before the first store (C may alias R)
            const char* r = a.R + (size_t)mf * a.ldr_bytes + (size_t)nb * 2;
            const size_t rrs = (size_t)16 * a.ldr_bytes;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    rv[mt][nt] = (u32x2_t){0u, 0u};
                    if (FULL || (mf + mt * 16 < a.M && nb + nt * 16 < a.N)) rv[mt][nt] = *(const u32x2_t*)(r + mt * rrs + nt * 32);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (!FULL && mf + mt * 16 >= a.M) continue;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (!FULL && nb + nt * 16 >= a.N) continue;
                unsigned v0 = pack_bf16x2(acc[nt][mt][0], acc[nt][mt][1]);
                unsigned v1 = pack_bf16x2(acc[nt][mt][2], acc[nt][mt][3]);
                if constexpr (EPI == EPI_RESIDUAL) {
                    v0 = pack_bf16x2(bf16_lo(v0) + bf16_lo(rv[mt][nt][0]), bf16_hi(v0) + bf16_hi(rv[mt][nt][0]));
                    v1 = pack_bf16x2(bf16_lo(v1) + bf16_lo(rv[mt][nt][1]), bf16_hi(v1) + bf16_hi(rv[mt][nt][1]));
                }
                *(u32x2_t*)(c + mt * rs + nt * 32) = (u32x2_t){v0, v1};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 96x192 / 192x96 engine (gemm_tile192.h): few token rows, one tile per CU, three ring stages.  No row gather, no device
// row count, no split-K.
template <int EPI, int MT, int STAGES, bool TALL>
__global__ __launch_bounds__(256, 2) void gemm192_bf16_kernel(GemmArgs a) {
    using T = tile192::Tile<MT, STAGES, TALL>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1;

    // consecutive ids walk the row tiles of one feature panel, an XCD's chunk of ids covers whole panels:
    // every weight panel is fetched from HBM by one XCD only, the (small) activation matrix lives in every L2
    const int ntile = a.tiles_m * a.tiles_n;
    int id = xcd_remap(blockIdx.x, ntile * a.ksplit);
    // split-K (EPI_F32 only): K range ks of tile id writes its fp32 partial tile to slab ks; the ranges of a tile are
    // neighbours in the id order
    const int ks = id % a.ksplit;
    id /= a.ksplit;
    const int tm = id % a.tiles_m, tn = id / a.tiles_m;
    const int m0 = tm * T::BM, n0 = tn * T::BN;
    const int kc = a.K / a.ksplit;            // host guarantees kc % BK == 0
    a.X += (size_t)ks * kc * 2;
    a.W += (size_t)ks * kc * 2;
    a.C += (size_t)ks * a.slab_bytes;
    a.K = kc;

    typename T::Src f;
    const int ch = T::dma_chunk(lane) * 16;
#pragma unroll
    for (int i = 0; i < T::XI; ++i) f.x[i] = a.X + (size_t)min(m0 + T::dma_row(i, wave, lane), a.M - 1) * a.ldx_bytes + ch;
#pragma unroll
    for (int i = 0; i < T::WI; ++i) f.w[i] = a.W + (size_t)min(n0 + T::dma_row(i, wave, lane), a.N - 1) * a.ldw_bytes + ch;

    f32x4_t acc[4][MT], tail[2][MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) tail[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

#ifdef CODA_CLOCK_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    T::mainloop(f, a.K, lds, wave, lane, acc, tail);
#ifdef CODA_CLOCK_PROBE
    if (a.probe && threadIdx.x == 0 && blockIdx.x < 256) {   // K loop only: shader cycles and 100 MHz ticks
        a.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_t0;
        a.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    }
#endif

    const bool full = m0 + T::BM <= a.M && n0 + T::BN <= a.N;
    const int mrow0 = m0 + wm * 16 * MT, ncol0 = n0 + wn * 96;
    epilogue_tile<EPI, MT>(a, acc, mrow0, ncol0, lane, full);
    if (full) epilogue_tail2<EPI, MT, true>(a, tail, mrow0, ncol0 + 64, lane);
    else epilogue_tail2<EPI, MT, false>(a, tail, mrow0, ncol0 + 64, lane);
}

template <int EPI, int MT, int STAGES = 3, bool TALL = false>
static int launch192(GemmArgs a, hipStream_t s) {
    using T = tile192::Tile<MT, STAGES, TALL>;
    if (coda_ensure_dyn_lds((const void*)gemm192_bf16_kernel<EPI, MT, STAGES, TALL>, T::LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    a.tiles_m = (a.M + T::BM - 1) / T::BM;
    a.tiles_n = (a.N + T::BN - 1) / T::BN;
    hipLaunchKernelGGL((gemm192_bf16_kernel<EPI, MT, STAGES, TALL>), dim3(a.tiles_m * a.tiles_n * a.ksplit), dim3(256), T::LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm192_bf16");
    return CODA_OK;
}
template <int EPI> static int launch96x192(GemmArgs a, hipStream_t s) { return launch192<EPI, 3>(a, s); }
template <int EPI> static int launch192x96(GemmArgs a, hipStream_t s) { return launch192<EPI, 3, 3, true>(a, s); }

// ---------------------------------------------------------------------------------------------
// 256x256 engine (gemm_tile256.h): same epilogues, no row gather.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmArgs a) {
    using T = tile256::Tile<4>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = a.tiles_m * a.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * a.tiles_n;
    const int g = id / per_group;
    const int gm0 = g * GROUP_M;
    const int gsz = min(a.tiles_m - gm0, GROUP_M);
    const int tm = gm0 + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    T::Src src;
    src.xbase = a.X;
    src.wbase = a.W;
    const int ch = tile256::dma_chunk(lane) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tile256::dma_row(i, wave, lane);
        const int gm = min(m0 + r, a.M - 1);
        const int gn = min(n0 + r, a.N - 1);
        src.xoff[i] = (unsigned)((size_t)gm * a.ldx_bytes + ch);
        src.woff[i] = (unsigned)((size_t)gn * a.ldw_bytes + ch);
    }

    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    T::mainloop(src, a.K, lds, wave, lane, acc);

    epilogue_tile<EPI, 8>(a, acc, m0 + wm * 128, n0 + wn * 64, lane, m0 + T::BM <= a.M && n0 + T::BN <= a.N);
}

template <int EPI>
static int launch256(GemmArgs a, hipStream_t s) {
    using T = tile256::Tile<4>;
    if (coda_ensure_dyn_lds((const void*)gemm256_bf16_kernel<EPI>, T::LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    a.tiles_m = (a.M + T::BM - 1) / T::BM;
    a.tiles_n = (a.N + T::BN - 1) / T::BN;
    hipLaunchKernelGGL(gemm256_bf16_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(512), T::LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm256_bf16");
    return CODA_OK;
}


// Fused q_norm / k_norm + rotary epilogue of the QKV projection (modeling_coda.py:128-148,
// modeling_utils.py:70-101) for the persistent 256x256 engine.  The weight rows of every q/k head are
// stored permuted as [0..31, 64..95, 32..63, 96..127], so the wave that owns 64 of the head's 128
// features holds, in accumulator pairs (nt, nt+2), exactly the (j, j+64) partners RoPE rotates;
// the RMS over the head's 128 features needs one 128-float exchange with the partner wave (LDS).
// V tiles take the plain epilogue.  Rounding points follow the reference's bf16 op sequence.
__device__ __forceinline__ void epilogue256_qkrope(const GemmArgs& a, f32x4_t (&acc)[4][8], int m0, int n0, int wm, int wn,
                                                   int wave, int lane, float* xch, bool full) {
    if (n0 >= a.qk_feat) {                    // V columns: tile-uniform branch
        epilogue_tile<EPI_PLAIN, 8>(a, acc, m0 + wm * 128, n0 + wn * 64, lane, full);
        return;
    }
    const int tok = lane & 15, fq = (lane >> 4) * 4;
    const int feat0 = n0 + (wn >> 1) * 128;   // first feature of this wave's head in the output row
    const int qh = wn & 1;
    const char* nw = feat0 < a.q_feat ? a.qn_w : a.kn_w;
    // rotary rows are fetched two token groups ahead of their use (the compiler cannot hoist them over
    // other memory operations itself); the prefetch cursor walks pos = m mod L incrementally
    const int mf = m0 + wm * 128 + tok;
    const bool step_pos = a.L >= 16;
    int ppos = mf % a.L;
    u32x2_t cq[8][2], sq[8][2];
    auto fetch = [&](int mt) {
        const char* crow = a.rope_cos + (size_t)ppos * 256 + (32 * qh + fq) * 2;
        const char* srow = a.rope_sin + (size_t)ppos * 256 + (32 * qh + fq) * 2;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            cq[mt][nt] = *(const u32x2_t*)(crow + nt * 32);
            sq[mt][nt] = *(const u32x2_t*)(srow + nt * 32);
        }
        if (step_pos) { ppos += 16; ppos -= ppos >= a.L ? a.L : 0; }
        else ppos = (mf + (mt + 1) * 16) % a.L;
    };
    fetch(0);
    fetch(1);
    // pass 1: round the projection to bf16 and keep it PACKED (two values per register: frees half of the
    // accumulator file for pass 2); partial sum of squares per token straight from the packed pairs
    unsigned pk[4][8][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            pk[nt][mt][0] = pack_bf16x2(acc[nt][mt][0], acc[nt][mt][1]);
            pk[nt][mt][1] = pack_bf16x2(acc[nt][mt][2], acc[nt][mt][3]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x2_t v = __builtin_bit_cast(bf16x2_t, pk[nt][mt][h]);
                t = __builtin_amdgcn_fdot2_f32_bf16(v, v, t, false);
            }
        }
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if ((lane >> 4) == 0) xch[wave * 128 + mt * 16 + tok] = t;
    }
    // norm weights of this lane's 2 x (4 + 4) features: loop-invariant over the tokens
    float w1[2][4], w2[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 32 * qh + nt * 16 + fq;
        const u32x2_t a1 = *(const u32x2_t*)(nw + j * 2), a2 = *(const u32x2_t*)(nw + (j + 64) * 2);
        w1[nt][0] = bf16_lo(a1[0]); w1[nt][1] = bf16_hi(a1[0]); w1[nt][2] = bf16_lo(a1[1]); w1[nt][3] = bf16_hi(a1[1]);
        w2[nt][0] = bf16_lo(a2[0]); w2[nt][1] = bf16_hi(a2[0]); w2[nt][2] = bf16_lo(a2[1]); w2[nt][3] = bf16_hi(a2[1]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // pass 2 keeps its packed results in registers and pass 3 stores them (vmcnt retires in order: a rotary
    // load issued behind output stores would wait for their write acknowledgements)
    u32x2_t outv[8][2][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const float tot = xch[wave * 128 + mt * 16 + tok] + xch[(wave ^ 1) * 128 + mt * 16 + tok];
        if (mt + 2 < 8) fetch(mt + 2);
        const float inv = __builtin_amdgcn_rsqf(tot * (1.0f / 128.0f) + a.eps);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const u32x2_t cv = cq[mt][nt], sv = sq[mt][nt];
            const float c[4] = {bf16_lo(cv[0]), bf16_hi(cv[0]), bf16_lo(cv[1]), bf16_hi(cv[1])};
            const float sn[4] = {bf16_lo(sv[0]), bf16_hi(sv[0]), bf16_lo(sv[1]), bf16_hi(sv[1])};
            float n1[4], n2[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                n1[2 * h] = bf16_lo(pk[nt][mt][h]) * inv; n1[2 * h + 1] = bf16_hi(pk[nt][mt][h]) * inv;
                n2[2 * h] = bf16_lo(pk[nt + 2][mt][h]) * inv; n2[2 * h + 1] = bf16_hi(pk[nt + 2][mt][h]) * inv;
            }
            round_bf16_pair(n1[0], n1[1]); round_bf16_pair(n1[2], n1[3]);
            round_bf16_pair(n2[0], n2[1]); round_bf16_pair(n2[2], n2[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { n1[r] *= w1[nt][r]; n2[r] *= w2[nt][r]; }
            round_bf16_pair(n1[0], n1[1]); round_bf16_pair(n1[2], n1[3]);
            round_bf16_pair(n2[0], n2[1]); round_bf16_pair(n2[2], n2[3]);
            float pa[4], pb[4], pc[4], pd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pa[r] = n1[r] * c[r]; pb[r] = n2[r] * sn[r]; pc[r] = n2[r] * c[r]; pd[r] = n1[r] * sn[r]; }
            round_bf16_pair(pa[0], pa[1]); round_bf16_pair(pa[2], pa[3]);
            round_bf16_pair(pb[0], pb[1]); round_bf16_pair(pb[2], pb[3]);
            round_bf16_pair(pc[0], pc[1]); round_bf16_pair(pc[2], pc[3]);
            round_bf16_pair(pd[0], pd[1]); round_bf16_pair(pd[2], pd[3]);
            // first half: x1 cos + (-x2) sin; second half: x2 cos + x1 sin
            outv[mt][nt][0] = (u32x2_t){pack_bf16x2(pa[0] - pb[0], pa[1] - pb[1]), pack_bf16x2(pa[2] - pb[2], pa[3] - pb[3])};
            outv[mt][nt][1] = (u32x2_t){pack_bf16x2(pc[0] + pd[0], pc[1] + pd[1]), pack_bf16x2(pc[2] + pd[2], pc[3] + pd[3])};
        }
    }
    const size_t rs = (size_t)16 * a.ldc_bytes;
    if (full && a.wide) {                     // 16-byte pieces (widen16): feature blocks nt = 0, 1 are neighbours in both halves
        char* ow = a.C + (size_t)mf * a.ldc_bytes + (size_t)(feat0 + 32 * qh) * 2 + widen16_offset(lane);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            *(u32x4_t*)(ow + mt * rs) = widen16(outv[mt][0][0], outv[mt][1][0]);
            *(u32x4_t*)(ow + mt * rs + 128) = widen16(outv[mt][0][1], outv[mt][1][1]);
        }
        return;
    }
    char* orow = a.C + (size_t)mf * a.ldc_bytes + (size_t)(feat0 + 32 * qh + fq) * 2;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        if (!full && mf + mt * 16 >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            *(u32x2_t*)(orow + mt * rs + nt * 32) = outv[mt][nt][0];
            *(u32x2_t*)(orow + mt * rs + nt * 32 + 128) = outv[mt][nt][1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent 256x256 engine: one workgroup per CU walks its output tiles; the LDS ring, the
// LDS-DMA stream and the fragment prefetch run CONTINUOUSLY across output-tile boundaries, so the
// first K tiles of the next output tile are already landing while the current tile's epilogue
// stores - no per-tile prologue bubble (worth ~8 % at K = 2048).
// The 192-token tile leaves LDS for ONE operand ring of three stages (its DMA then runs two K tiles ahead):
// XDEEP = false gives it to the weights, XDEEP = true to the activations.
template <int MTH, bool XDEEP = false>
struct PersistentLds {
    using T = tile256::Tile<MTH>;
    static constexpr int XST = XDEEP ? 3 : 2;                    // X ring depth (256-token tile + XDEEP: all 160 KiB)
    static constexpr int WST = MTH == 3 && !XDEEP ? 3 : 2;       // W ring depth
    static constexpr int XR = XST * T::X_BYTES;                  // W ring base (after the X ring)
    static constexpr int BYTES = XR + WST * T::W_BYTES;          // 128 KiB (256-token tile) / 144 or 136 KiB (192-token tile)
};

template <int EPI, int MTH, bool XDEEP = false>
__global__ __launch_bounds__(512, 2) void gemm256p_bf16_kernel(GemmArgs a) {
    using T = tile256::Tile<MTH>;
    static_assert(EPI != EPI_QKROPE || MTH == 4, "the fused q/k epilogue is written for 256-token tiles");
    if (a.m_count) {                          // device-side row count: shrink the problem before any tile is walked
        a.M = min(a.M, a.m_count[0]);
        if (a.M <= 0) return;
        a.tiles_m = (a.M + T::BM - 1) / T::BM;
    }
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ntiles = a.tiles_m * a.tiles_n;
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);
    const int nk = a.K / T::BK;
    // row tiles per group of the tile walk.  The fused-QKV launch uses 3: with 4, a workgroup's tiles
    // (slot + k*256) all fall in the same column band, so a quarter of the CUs would own nothing but the
    // cheap V tiles; 3 rotates the column band from round to round
    constexpr int GROUP_M = EPI == EPI_QKROPE ? 3 : 4;
    const int per_group = GROUP_M * a.tiles_n;
    auto coords = [&](int L, int& m0, int& n0) {
        const int g = L / per_group, gm0 = g * GROUP_M;
        const int gsz = min(a.tiles_m - gm0, GROUP_M);
        m0 = (gm0 + (L % per_group) % gsz) * T::BM;
        n0 = ((L % per_group) / gsz) * T::BN;
    };
    // LDS: X ring of XST stages, then W ring of WST stages.  The 192-token tile leaves room for a third stage of
    // one operand, whose DMA then runs one K tile further ahead.  Which operand is colder depends on the call site:
    // every layer's weight panel comes from HBM once, but with K = 6144 (down_proj) it is the 151 MB activation
    // matrix the previous kernel wrote that misses every cache (tools/cold_bench.py).
    using PL = PersistentLds<MTH, XDEEP>;
    constexpr int XST = PL::XST, WST = PL::WST, XR = PL::XR;
    constexpr int YOUNG = WST == 3 ? 4 : XST == 3 ? T::X_DMA : 0;   // DMA instructions of the deeper ring's newest tile
    const int ch = tile256::dma_chunk(lane) * 16;
    unsigned xoff[T::X_DMA], woff[4];           // per-lane byte offsets of the DMA instructions of each operand
    auto set_x = [&](int m0) {
#pragma unroll
        for (int i = 0; i < T::X_DMA; ++i)
            xoff[i] = (unsigned)((size_t)min(m0 + tile256::dma_row(i, wave, lane), a.M - 1) * a.ldx_bytes + ch);
    };
    auto set_w = [&](int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            woff[i] = (unsigned)((size_t)min(n0 + tile256::dma_row(i, wave, lane), a.N - 1) * a.ldw_bytes + ch);
    };

    int Lc = slot;
    if (Lc >= ntiles) return;
    int cm0, cn0;
    coords(Lc, cm0, cn0);
    set_x(cm0);
    set_w(cn0);
    // two DMA cursors over the workgroup's stream of K tiles (output tile, K tile): the next one to stage
    int xL = Lc, xk = 0, wL = Lc, wk = 0;
    auto dma_x = [&](char* dst) -> bool {
        if (xL >= ntiles) return false;
        {
#ifdef CODA_DIAG_HOTK                            // diagnostic builds only: every K tile re-reads K tile 0 (cache-hot operands)
            const size_t kb = 0;
#else
            const size_t kb = (size_t)xk * T::BK * 2;
#endif
#pragma unroll
            for (int i = 0; i < T::X_DMA; ++i) glds16(a.X + kb + xoff[i], dst + (i * 8 + wave) * 1024);
            if (++xk == nk) {
                xk = 0;
                xL += G;
                if (xL < ntiles) {
                    int m0, n0;
                    coords(xL, m0, n0);
                    set_x(m0);
                }
            }
        }
        return true;
    };
    auto dma_w = [&](char* dst) -> bool {
        if (wL >= ntiles) return false;
#ifdef CODA_DIAG_HOTK
        const size_t kb = 0;
#else
        const size_t kb = (size_t)wk * T::BK * 2;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a.W + kb + woff[i], dst + (i * 8 + wave) * 1024);
        if (++wk == nk) {
            wk = 0;
            wL += G;
            if (wL < ntiles) {
                int m0, n0;
                coords(wL, m0, n0);
                set_w(n0);
            }
        }
        return true;
    };
#pragma unroll
    for (int i = 0; i < XST; ++i) dma_x(lds + i * T::X_BYTES);
#pragma unroll
    for (int i = 0; i < WST; ++i) dma_w(lds + XR + i * T::W_BYTES);
    tile256::tile_sync();

    const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
    const int p0 = ((0 * 4 + kq) ^ sw) * 16, p1 = ((1 * 4 + kq) ^ sw) * 16;
    const char* xs = lds + (wm * (T::BM / 2) + r16) * 128;
    const char* ws = lds + XR + (wn * 64 + r16) * 128;
    typename T::Frags f;
    T::template load_w<0, 0>(f, ws, p0);
    T::template load_x<0, 0>(f, xs, p0);
#ifdef CODA_CLOCK_PROBE
    // diagnostic build only (tools/clock_probe.sh): in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around
    // the workgroup's whole tile stream; the stamps go to a buffer of their own (coda_gemm_probe)
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // stores one wave issues per full tile: 8-byte pieces, or 16-byte pieces on the widened path (epilogue_wave)
    constexpr int ST_NARROW = (EPI == EPI_SWIGLU ? 2 : 4) * T::MT;
    constexpr int ST_WIDE = EPI == EPI_F32 ? ST_NARROW : ST_NARROW / 2;
    bool young = false;                       // 3-stage ring: a DMA younger than what the next sync needs is in flight
    auto issue_dma = [&](int xdst, int wdst) {
#ifdef CODA_DIAG_NO_DMA
        return;
#endif
        if constexpr (XST == 3) {
            dma_w(lds + XR + wdst);
            young = dma_x(lds + xdst);
        } else {
            dma_x(lds + xdst);
            young = dma_w(lds + XR + wdst);
        }
    };
#ifndef CODA_DMA_LATE_HALF
#define CODA_DMA_LATE_HALF(w) ((w) >> 2)
#endif
    const bool late = CODA_DMA_LATE_HALF(wave) & 1;
    bool pend = false;
    int pend_x = 0, pend_w = 0;
    int xcur = 0, wcur = 0;                   // byte offsets of the ring stages holding the current K tile
    int stores_in_flight = 0;                 // stores of the previous tile's epilogue if it was the unconditional (full-tile) one
    // RESIDUAL on the 192-token tile (o_proj / down_proj: 2 rounds of 256 workgroups in lockstep): the residual tile is
    // fetched DURING the last K tiles, one token group per K tile, instead of at the head of the epilogue, where all 256
    // workgroups asked for their 98 KB at the same moment with the MFMA pipes idle (o_proj: 101 us with the residual
    // epilogue against 85 plain).  The loads stand in front of their K tile's DMA, so the K-tile waits (which let only what
    // is YOUNGER than the DMA stay in flight) cover them without a change.  48 registers (180 -> 228 of 256).
    constexpr bool RPF = EPI == EPI_RESIDUAL && MTH == 3;
    ResidualTile<T::MT> rtile;
    for (; Lc < ntiles; Lc += G) {
        coords(Lc, cm0, cn0);
        const bool full = cm0 + T::BM <= a.M && cn0 + T::BN <= a.N;
        const bool rpf = RPF && full && a.wide && nk >= T::MT;
        f32x4_t acc[4][T::MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < T::MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t) {
            if constexpr (RPF) {
                if (rpf && t >= nk - T::MT) {
                    const int g = t - (nk - T::MT);          // wave-uniform; the switch keeps the register index static
#pragma unroll
                    for (int mt = 0; mt < T::MT; ++mt)
                        if (g == mt) residual_prefetch<T::MT>(a, rtile, mt, cm0 + wm * (T::BM / 2), cn0 + wn * 64, lane);
                }
            }
            const int xnxt = xcur == (XST - 1) * T::X_BYTES ? 0 : xcur + T::X_BYTES;
            const int wnxt = wcur == (WST - 1) * T::W_BYTES ? 0 : wcur + T::W_BYTES;
            const char* xc = xs + xcur;
            const char* wc = ws + wcur;
#ifndef CODA_DMA_LATE_SLOT
#define CODA_DMA_LATE_SLOT 1                     // the late half issues after phase 1 of the next K tile (A/B: 0..3)
#endif
            auto late_dma = [&]() {               // late half: the DMA of the previous K tile's sync (see below)
                if (pend) {
                    issue_dma(pend_x, pend_w);
                    pend = false;
                }
            };
            tile256::phase_begin(); T::template load_w<1, 1>(f, wc, p0); T::template quad<0, 0, 0, 0>(f, acc);
            if (CODA_DMA_LATE_SLOT == 0) late_dma();
            tile256::phase_begin(); T::template load_x<1, 1>(f, xc, p0); T::template quad<1, 0, 1, 0>(f, acc);
            if (CODA_DMA_LATE_SLOT == 1) late_dma();
            tile256::phase_begin(); T::template load_x<0, 0>(f, xc, p1); T::template quad<1, 1, 1, 1>(f, acc);
            if (CODA_DMA_LATE_SLOT == 2) late_dma();
            tile256::phase_begin(); T::template load_w<1, 1>(f, wc, p1); T::template quad<0, 1, 0, 1>(f, acc);
            if (CODA_DMA_LATE_SLOT == 3) late_dma();
            tile256::phase_begin(); T::template load_w<0, 0>(f, wc, p1); T::template quad<1, 0, 1, 0>(f, acc);
            tile256::phase_begin(); T::template load_x<1, 1>(f, xc, p1); T::template quad<0, 0, 0, 0>(f, acc);
            // Every wave has retired its reads of these stages, and the next K tile of the stream must have
            // landed.  vmcnt retires in order, so what may stay in flight is exactly what was issued AFTER
            // the next tile's DMA: the deeper ring's tile one further ahead (YOUNG instructions, issued LAST in its
            // iteration) and the previous output tile's epilogue stores.
#ifdef CODA_DIAG_NO_SYNC                         // diagnostic builds only: no K-tile wait / barrier (wrong results)
            if (false)
#endif
#ifdef CODA_DIAG_NO_VMCNT                        // diagnostic builds only: barrier without waiting for the DMA
            if (true) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else
#endif
            {
                const int st = t == 0 ? stores_in_flight : 0;
                if (YOUNG && young) {
                    if (st == ST_NARROW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_NARROW + YOUNG) : "memory");
                    else if (st == ST_WIDE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_WIDE + YOUNG) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(YOUNG) : "memory");
                } else {
                    if (st == ST_NARROW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_NARROW) : "memory");
                    else if (st == ST_WIDE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_WIDE) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
#ifndef CODA_DIAG_NO_BARRIER
                __builtin_amdgcn_s_barrier();
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // The CU's address path takes one 1-KiB LDS-DMA piece per ~16 cycles, and a wave cannot issue its next MFMA
            // until its own pieces are accepted: eight waves issuing their eight pieces together after the barrier
            // leave the MFMA pipe idle for ~600 cycles per K tile (tools/clock_probe.py: 3000 cycles per K tile with
            // the DMA, 2400 without, whether or not anybody waits for the data).  So the two waves that share a SIMD
            // issue at different times: one right after the barrier while its partner multiplies alone, the partner two
            // phases into the next K tile (after phase 7 when an epilogue follows) while the first one catches up:
            // 901k -> 779k cycles per workgroup on the gate|up shape; the chip gives about half of it back as a lower
            // clock (2.01 -> 1.86 GHz), -6 % wall.
            if (!late) issue_dma(xcur, wcur);
            T::template load_x<0, 0>(f, xs + xnxt, p0);      // first fragments of the next K tile (stale bytes after the last one: unused)
            T::template quad<0, 1, 0, 1>(f, acc);
            tile256::phase_begin();
            T::template load_w<0, 0>(f, ws + wnxt, p0);
            T::template quad<1, 1, 1, 1>(f, acc);
            if (late) {
                if (t + 1 < nk) {                 // ... two phases into the next K tile, clear of the first half's burst
                    pend = true;
                    pend_x = xcur;
                    pend_w = wcur;
                } else {                          // before an epilogue: keeps [DMA][epilogue stores] in issue order
                    issue_dma(xcur, wcur);
                }
            }
            xcur = xnxt;
            wcur = wnxt;
        }
#ifdef CODA_DIAG_NO_EPI                          // diagnostic builds only: the epilogue never runs (the test keeps the accumulators alive)
        if (__builtin_amdgcn_readfirstlane(__float_as_int(acc[0][0][0])) == 0x12345678) {
#else
        {
#endif
            if constexpr (EPI == EPI_QKROPE) epilogue256_qkrope(a, acc, cm0, cn0, wm, wn, wave, lane, (float*)(lds + PL::BYTES), full);
            else if (RPF && rpf) epilogue_wave<EPI, T::MT, true, RPF>(a, acc, cm0 + wm * (T::BM / 2), cn0 + wn * 64, lane, &rtile);
            else epilogue_tile<EPI, T::MT>(a, acc, cm0 + wm * (T::BM / 2), cn0 + wn * 64, lane, full);
        }
        stores_in_flight = full ? (a.wide && EPI != EPI_F32 ? ST_WIDE : ST_NARROW) : 0;
#ifdef CODA_DIAG_NO_EPI
        stores_in_flight = 0;
#endif
    }
#ifdef CODA_CLOCK_PROBE
    if (a.probe && threadIdx.x == 0) {
        unsigned long long* out = a.probe + 2 * blockIdx.x;
        out[0] = __builtin_amdgcn_s_memtime() - probe_t0;
        out[1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    }
#endif
}

template <int EPI, int MTH = 4, bool XDEEP = false>
static int launch256p(GemmArgs a, hipStream_t s) {
    using T = tile256::Tile<MTH>;
    constexpr int lds_bytes = PersistentLds<MTH, XDEEP>::BYTES + (EPI == EPI_QKROPE ? 4096 : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS per CU");
    if (coda_ensure_dyn_lds((const void*)gemm256p_bf16_kernel<EPI, MTH, XDEEP>, lds_bytes) != CODA_OK) return CODA_ERR_LAUNCH;
    const int n_cu = coda_device_cus();
    a.tiles_m = (a.M + T::BM - 1) / T::BM;
    a.tiles_n = (a.N + T::BN - 1) / T::BN;
    const int grid = a.tiles_m * a.tiles_n < n_cu ? a.tiles_m * a.tiles_n : n_cu;
    hipLaunchKernelGGL((gemm256p_bf16_kernel<EPI, MTH, XDEEP>), dim3(grid), dim3(512), lds_bytes, s, a);
    CODA_CHECK_LAUNCH("gemm256p_bf16");
    return CODA_OK;
}


// ---------------------------------------------------------------------------------------------
// Persistent 256x128 engine (gemm_tile256x128.h) for the N = 2048 projections.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256x128p_bf16_kernel(GemmArgs a) {
    namespace T = tile256x128;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = a.tiles_m * a.tiles_n;
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);
    const int nk = a.K / T::BK;
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * a.tiles_n;
    auto coords = [&](int L, int& m0, int& n0) {
        const int g = L / per_group, gm0 = g * GROUP_M;
        const int gsz = min(a.tiles_m - gm0, GROUP_M);
        m0 = (gm0 + (L % per_group) % gsz) * T::BM;
        n0 = ((L % per_group) / gsz) * T::BN;
    };
    T::Src src;
    src.xbase = a.X;
    src.wbase = a.W;
    const int ch = tile256::dma_chunk(lane) * 16;
    auto make_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            src.xoff[i] = (unsigned)((size_t)min(m0 + tile256::dma_row(i, wave, lane), a.M - 1) * a.ldx_bytes + ch);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            src.woff[i] = (unsigned)((size_t)min(n0 + tile256::dma_row(i, wave, lane), a.N - 1) * a.ldw_bytes + ch);
    };

    int Lc = slot;
    if (Lc >= ntiles) return;
    int cm0, cn0;
    coords(Lc, cm0, cn0);
    make_src(cm0, cn0);
    int dL = Lc, dk = 0;
    auto dma_next = [&](char* dst) {
        if (dL < ntiles) {
            T::stage(src, (size_t)dk * T::BK * 2, dst, wave);
            if (++dk == nk) {
                dk = 0;
                dL += G;
                if (dL < ntiles) {
                    int m0, n0;
                    coords(dL, m0, n0);
                    make_src(m0, n0);
                }
            }
        }
    };
    // 3-stage ring: the DMA of K tile q+3 is issued at the sync of tile q, so two K tiles (~1 us of MFMA
    // work) are in flight ahead of the one being multiplied - enough to cover an HBM miss.
    int issued = 0;                           // K tiles staged so far (wave-uniform)
    for (int i = 0; i < 3; ++i) {
        const bool had = dL < ntiles;
        dma_next(lds + i * T::STAGE_BYTES);
        issued += had ? 1 : 0;
    }
    if (issued >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
    const int p0 = ((0 * 4 + kq) ^ sw) * 16, p1 = ((1 * 4 + kq) ^ sw) * 16;
    const char* xs = lds + (wm * 64 + r16) * 128;
    const char* ws = lds + T::X_BYTES + (wn * 64 + r16) * 128;
    T::Frags f;
    T::load_w<0, 0>(f, ws, p0);
    T::load_x<0>(f, xs, p0);

    constexpr int ST_NARROW = (EPI == EPI_SWIGLU) ? 8 : 16;
    constexpr int ST_WIDE = EPI == EPI_F32 ? ST_NARROW : ST_NARROW / 2;
    int cur = 0, q = 0;                       // LDS byte offset of the current stage; stream position
    int stores_in_flight = 0;
    for (; Lc < ntiles; Lc += G) {
        coords(Lc, cm0, cn0);
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t) {
            const int nxt = cur == 2 * T::STAGE_BYTES ? 0 : cur + T::STAGE_BYTES;
            const char* xc = xs + cur;
            const char* wc = ws + cur;
            T::phase_begin(); T::load_w<1, 1>(f, wc, p0); T::half<0, 0, 0>(f, acc);
            T::phase_begin(); T::load_x<1>(f, xc, p1); T::load_w<0, 0>(f, wc, p1); T::half<1, 0, 1>(f, acc);
            T::phase_begin(); T::load_w<1, 1>(f, wc, p1); T::half<0, 1, 0>(f, acc);
            // need K tile q+1 landed; tile q+2 (6 DMA instructions per wave) and the previous output tile's
            // epilogue stores may stay in flight (both are younger than tile q+1's DMA)
            {
                const bool ahead = issued > q + 2;
                const int st = t == 0 ? stores_in_flight : 0;
                if (ahead) {
                    if (st == ST_NARROW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_NARROW + 6) : "memory");
                    else if (st == ST_WIDE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_WIDE + 6) : "memory");
                    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                } else {
                    if (st == ST_NARROW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_NARROW) : "memory");
                    else if (st == ST_WIDE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(ST_WIDE) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const bool had = dL < ntiles;
                dma_next(lds + cur);
                issued += had ? 1 : 0;
            }
            ++q;
            T::load_x<0>(f, xs + nxt, p0);
            T::load_w<0, 0>(f, ws + nxt, p0);
            T::half<1, 1, 1>(f, acc);
            cur = nxt;
        }
        const bool full = cm0 + T::BM <= a.M && cn0 + T::BN <= a.N;
        epilogue_tile<EPI, 4>(a, acc, cm0 + wm * 64, cn0 + wn * 64, lane, full);
        stores_in_flight = full ? (a.wide && EPI != EPI_F32 ? ST_WIDE : ST_NARROW) : 0;
    }
}

template <int EPI>
static int launch256x128p(GemmArgs a, hipStream_t s) {
    if (coda_ensure_dyn_lds((const void*)gemm256x128p_bf16_kernel<EPI>, tile256x128::LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    const int n_cu = coda_device_cus();
    a.tiles_m = (a.M + tile256x128::BM - 1) / tile256x128::BM;
    a.tiles_n = (a.N + tile256x128::BN - 1) / tile256x128::BN;
    const int grid = a.tiles_m * a.tiles_n < n_cu ? a.tiles_m * a.tiles_n : n_cu;
    hipLaunchKernelGGL(gemm256x128p_bf16_kernel<EPI>, dim3(grid), dim3(512), tile256x128::LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm256x128p_bf16");
    return CODA_OK;
}

// ---------------------------------------------------------------------------------------------
// Persistent 256x256 (192x256) engine with FOUR waves, one per SIMD ("w4"): wave tile 128 (96) tokens x 128
// features, 256 (192) accumulator registers per lane out of the 512 a lone wave on a SIMD may hold.
//
// Why: with 8 waves the wave tile is 128 x 64 and every K tile costs 8 waves x 24 KB = 192 KB of LDS fragment reads
// per 2048 MFMA cycles; a 128 x 128 wave tile needs 4 x 32 KB = 128 KB for the same MFMAs (the fragment of one
// operand is reused by 8 MFMAs instead of 4), and no two waves compete for a SIMD's issue port.  This is the shape
// the vendor library's 256x256x64 kernels use (4 waves, wave tile 8 x 8 MFMA tiles).
//
// LDS image, DMA pattern and MFMA orientation are those of gemm_tile256.h (128-byte rows, 16-byte chunks XOR-swizzled
// with row & 7 on the DMA source side, weight fragment = MFMA operand A).  Two stages of one K tile (64 deep); the
// flat stream of K tiles q = 0, 1, ... runs across output tiles (persistent), stage = q & 1:
//     k-step 0 of q : 64 MFMAs on fragment set F0      | LDS reads: F1 <- k-step 1 of q      (stage q & 1)
//     lgkmcnt(0), vmcnt(stores of a just-finished epilogue), s_barrier      -> stage q & 1 is dead, tile q+1 has landed
//     k-step 1 of q : 64 MFMAs on F1                   | LDS-DMA of tile q+2 into stage q & 1, one X and one W piece per
//                                                        group of 8 MFMAs; LDS reads: F0 <- k-step 0 of q+1
// Every instruction that is not an MFMA is issued inside a group of 8 MFMAs (128 cycles of matrix pipe).
namespace tile256w4 {
template <int MT>                                       // token groups of 16 per wave: 8 -> 256-token tile, 6 -> 192
struct Tile {
    static constexpr int BM = 2 * MT * 16, BN = 256, BK = 64;
    static constexpr int X_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = X_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 / 112 KiB
    static constexpr int X_DMA = BM / 32, W_DMA = 8;    // LDS-DMA instructions per wave and K tile (4 waves x 8 rows each)
};
__device__ __forceinline__ int dma_row(int i, int wave, int lane) { return (i * 4 + wave) * 8 + (lane >> 3); }
}  // namespace tile256w4

// Fused q_norm / k_norm + rotary epilogue for the four-wave engine (modeling_coda.py:128-148, modeling_utils.py:70-101).
// A wave's 128 features are exactly ONE head (tiles are 256 features wide and head_dim is 128), so everything a token
// needs sits in the four lanes that share lane & 15: the RMS over the head is an in-lane sum plus two shuffles, and the
// rotary partner of feature j = 16 nt + 4 (lane >> 4) + r is j + 64 = the same register of block nt + 4.  No LDS, no
// partner wave, and the weight rows stay in their natural order.  Rounding points follow the reference's bf16 op
// sequence: projection -> bf16; x * rsqrt(mean x^2 + eps) -> bf16; * norm weight -> bf16; the four rotary products ->
// bf16 each; their sum / difference -> bf16.  V tiles take the plain epilogue.
__device__ __forceinline__ void epilogue256w4_qkrope(const GemmArgs& a, const f32x4_t (&acc)[8][8], int mrow0, int ncol0, int lane,
                                                     bool full) {
    if (ncol0 >= a.qk_feat) {                 // V columns: tile-uniform branch
        epilogue_tile<EPI_PLAIN, 8>(a, *(const f32x4_t(*)[4][8])&acc[0], mrow0, ncol0, lane, full);
        epilogue_tile<EPI_PLAIN, 8>(a, *(const f32x4_t(*)[4][8])&acc[4], mrow0, ncol0 + 64, lane, full);
        return;
    }
    const int tok = lane & 15, fq = (lane >> 4) * 4;
    const char* nw = ncol0 < a.q_feat ? a.qn_w : a.kn_w;
    const int mf = mrow0 + tok;
    // norm weights of the lane's 2 x 16 features (first half nt 0..3, second half nt 4..7): loop-invariant over tokens
    float w1[4][4], w2[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int j = nt * 16 + fq;
        const u32x2_t a1 = *(const u32x2_t*)(nw + j * 2), a2 = *(const u32x2_t*)(nw + (j + 64) * 2);
        w1[nt][0] = bf16_lo(a1[0]); w1[nt][1] = bf16_hi(a1[0]); w1[nt][2] = bf16_lo(a1[1]); w1[nt][3] = bf16_hi(a1[1]);
        w2[nt][0] = bf16_lo(a2[0]); w2[nt][1] = bf16_hi(a2[0]); w2[nt][2] = bf16_lo(a2[1]); w2[nt][3] = bf16_hi(a2[1]);
    }
    const size_t rs = (size_t)16 * a.ldc_bytes;
    const bool wide = full && a.wide;
    char* ow = a.C + (size_t)mf * a.ldc_bytes + (size_t)ncol0 * 2 + widen16_offset(lane);
    char* orow = a.C + (size_t)mf * a.ldc_bytes + (size_t)(ncol0 + fq) * 2;
    // rotary rows of token group mt + 1 are fetched while group mt is worked on
    u32x2_t cq[2][4], sq[2][4];
    auto fetch = [&](int mt) {
        const int pos = (mf + mt * 16) % a.L;
        const char* crow = a.rope_cos + (size_t)pos * 256 + fq * 2;
        const char* srow = a.rope_sin + (size_t)pos * 256 + fq * 2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            cq[mt & 1][nt] = *(const u32x2_t*)(crow + nt * 32);
            sq[mt & 1][nt] = *(const u32x2_t*)(srow + nt * 32);
        }
    };
    fetch(0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        if (mt + 1 < 8) fetch(mt + 1);
        unsigned pk[8][2];
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            pk[nt][0] = pack_bf16x2(acc[nt][mt][0], acc[nt][mt][1]);
            pk[nt][1] = pack_bf16x2(acc[nt][mt][2], acc[nt][mt][3]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x2_t v = __builtin_bit_cast(bf16x2_t, pk[nt][h]);
                t = __builtin_amdgcn_fdot2_f32_bf16(v, v, t, false);
            }
        }
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        const float inv = __builtin_amdgcn_rsqf(t * (1.0f / 128.0f) + a.eps);
        u32x2_t o1[4], o2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const u32x2_t cv = cq[mt & 1][nt], sv = sq[mt & 1][nt];
            const float c[4] = {bf16_lo(cv[0]), bf16_hi(cv[0]), bf16_lo(cv[1]), bf16_hi(cv[1])};
            const float sn[4] = {bf16_lo(sv[0]), bf16_hi(sv[0]), bf16_lo(sv[1]), bf16_hi(sv[1])};
            float n1[4], n2[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                n1[2 * h] = bf16_lo(pk[nt][h]) * inv; n1[2 * h + 1] = bf16_hi(pk[nt][h]) * inv;
                n2[2 * h] = bf16_lo(pk[nt + 4][h]) * inv; n2[2 * h + 1] = bf16_hi(pk[nt + 4][h]) * inv;
            }
            round_bf16_pair(n1[0], n1[1]); round_bf16_pair(n1[2], n1[3]);
            round_bf16_pair(n2[0], n2[1]); round_bf16_pair(n2[2], n2[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { n1[r] *= w1[nt][r]; n2[r] *= w2[nt][r]; }
            round_bf16_pair(n1[0], n1[1]); round_bf16_pair(n1[2], n1[3]);
            round_bf16_pair(n2[0], n2[1]); round_bf16_pair(n2[2], n2[3]);
            float pa[4], pb[4], pc[4], pd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { pa[r] = n1[r] * c[r]; pb[r] = n2[r] * sn[r]; pc[r] = n2[r] * c[r]; pd[r] = n1[r] * sn[r]; }
            round_bf16_pair(pa[0], pa[1]); round_bf16_pair(pa[2], pa[3]);
            round_bf16_pair(pb[0], pb[1]); round_bf16_pair(pb[2], pb[3]);
            round_bf16_pair(pc[0], pc[1]); round_bf16_pair(pc[2], pc[3]);
            round_bf16_pair(pd[0], pd[1]); round_bf16_pair(pd[2], pd[3]);
            // first half: x1 cos - x2 sin; second half: x2 cos + x1 sin
            o1[nt] = (u32x2_t){pack_bf16x2(pa[0] - pb[0], pa[1] - pb[1]), pack_bf16x2(pa[2] - pb[2], pa[3] - pb[3])};
            o2[nt] = (u32x2_t){pack_bf16x2(pc[0] + pd[0], pc[1] + pd[1]), pack_bf16x2(pc[2] + pd[2], pc[3] + pd[3])};
        }
        if (wide) {                           // 16-byte pieces: block pairs (0,1) and (2,3) of each half
            *(u32x4_t*)(ow + mt * rs) = widen16(o1[0], o1[1]);
            *(u32x4_t*)(ow + mt * rs + 64) = widen16(o1[2], o1[3]);
            *(u32x4_t*)(ow + mt * rs + 128) = widen16(o2[0], o2[1]);
            *(u32x4_t*)(ow + mt * rs + 192) = widen16(o2[2], o2[3]);
        } else if (full || mf + mt * 16 < a.M) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                *(u32x2_t*)(orow + mt * rs + nt * 32) = o1[nt];
                *(u32x2_t*)(orow + mt * rs + nt * 32 + 128) = o2[nt];
            }
        }
    }
}

template <int EPI, int MT>
__global__ __launch_bounds__(256, 1) void gemm256w4_bf16_kernel(GemmArgs a) {
    using T = tile256w4::Tile<MT>;
    if (a.m_count) {
        a.M = min(a.M, a.m_count[0]);
        if (a.M <= 0) return;
        a.tiles_m = (a.M + T::BM - 1) / T::BM;
    }
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = a.tiles_m * a.tiles_n;
    const int G = gridDim.x;
    const int slot = xcd_remap(blockIdx.x, G);
    const int nk = a.K / T::BK;
    // the fused-QKV launch uses 3 row tiles per group: with 4 a workgroup's tiles (slot + k * 256) all fall in one column band,
    // so a quarter of the CUs would own nothing but the cheap V tiles
    constexpr int GROUP_M = EPI == EPI_QKROPE ? 3 : 4;
    const int per_group = GROUP_M * a.tiles_n;
    auto coords = [&](int L, int& m0, int& n0) {
        const int g = L / per_group, gm0 = g * GROUP_M;
        const int gsz = min(a.tiles_m - gm0, GROUP_M);
        m0 = (gm0 + (L % per_group) % gsz) * T::BM;
        n0 = ((L % per_group) / gsz) * T::BN;
    };
    const int ch = tile256::dma_chunk(lane) * 16;
    unsigned xoff[T::X_DMA], woff[T::W_DMA];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < T::X_DMA; ++i)
            xoff[i] = (unsigned)((size_t)min(m0 + tile256w4::dma_row(i, wave, lane), a.M - 1) * a.ldx_bytes + ch);
#pragma unroll
        for (int i = 0; i < T::W_DMA; ++i)
            woff[i] = (unsigned)((size_t)min(n0 + tile256w4::dma_row(i, wave, lane), a.N - 1) * a.ldw_bytes + ch);
    };
    int Lc = slot;
    if (Lc >= ntiles) return;
    int cm0, cn0;
    coords(Lc, cm0, cn0);
    set_src(cm0, cn0);
    // DMA cursor over the workgroup's flat stream of K tiles: (output tile dL, K tile dk) is the next one to stage
    int dL = Lc, dk = 0;
    size_t dkb = 0;                            // byte offset of K tile dk inside a row
    // past the end of the stream the pieces keep re-reading the last valid K tile into a stage nobody reads again:
    // no branch around the DMA instructions, and the vmcnt bookkeeping stays uniform
    bool dma_live = true;
    auto dma_piece_x = [&](int i, char* stage) { glds16(a.X + dkb + xoff[i], stage + (i * 4 + wave) * 1024); };
    auto dma_piece_w = [&](int i, char* stage) { glds16(a.W + dkb + woff[i], stage + T::X_BYTES + (i * 4 + wave) * 1024); };
    auto dma_advance = [&]() {
        if (!dma_live) return;
        dkb += T::BK * 2;
        if (++dk == nk) {
            dk = 0;
            dkb = 0;
            dL += G;
            if (dL < ntiles) {
                int m0, n0;
                coords(dL, m0, n0);
                set_src(m0, n0);
            } else {
                dma_live = false;
            }
        }
    };
    for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < T::X_DMA; ++i) dma_piece_x(i, lds + st * T::STAGE_BYTES);
#pragma unroll
        for (int i = 0; i < T::W_DMA; ++i) dma_piece_w(i, lds + st * T::STAGE_BYTES);
        dma_advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
    const int p0 = ((0 * 4 + kq) ^ sw) * 16, p1 = ((1 * 4 + kq) ^ sw) * 16;
    const char* xs = lds + (wm * (T::BM / 2) + r16) * 128;
    const char* ws = lds + T::X_BYTES + (wn * 128 + r16) * 128;
    bf16x8_t fw[2][8], fx[2][MT];              // fragment sets F0 / F1: [set][nt] weights, [set][mt] activations
    auto read_w = [&](int set, int g, const char* base, int pos) { fw[set][g] = *(const bf16x8_t*)(base + g * 2048 + pos); };
    auto read_x = [&](int set, int g, const char* base, int pos) { fx[set][g] = *(const bf16x8_t*)(base + g * 2048 + pos); };
#pragma unroll
    for (int g = 0; g < 8; ++g) read_w(0, g, ws, p0);
#pragma unroll
    for (int g = 0; g < MT; ++g) read_x(0, g, xs, p0);
    __builtin_amdgcn_s_waitcnt(0xC07F);

    // epilogue stores per wave and full tile (two 64-feature halves; see epilogue_wave)
    constexpr int ST_NARROW = 2 * (EPI == EPI_SWIGLU ? 2 : 4) * MT;
    constexpr int ST_WIDE = EPI == EPI_F32 ? ST_NARROW : ST_NARROW / 2;
    int stores_in_flight = 0;
    int cur = 0;                               // byte offset of the stage that holds the current K tile
    for (; Lc < ntiles; Lc += G) {
        coords(Lc, cm0, cn0);
        f32x4_t acc[8][MT];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nk; ++t) {
            const int nxt = T::STAGE_BYTES - cur;
            const char* xc = xs + cur;
            const char* wc = ws + cur;
            // ---- k-step 0: MFMAs on F0, reads of F1 (k-step 1 of this tile)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                read_w(1, g, wc, p1);
                if (g < MT) read_x(1, g, xc, p1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][g], fx[0][i], acc[g][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- every wave's reads of this stage have retired; the next K tile of the stream has landed (vmcnt retires
            // in order: what may stay in flight is what was issued AFTER its DMA = a just-finished epilogue's stores)
            // (the builtin form of s_waitcnt: the compiler's own counter model sees it and adds no lgkmcnt(0) of its own
            // behind the next reads; simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14)
            {
                constexpr int SN = ST_NARROW > 63 ? 63 : ST_NARROW, SW = ST_WIDE > 63 ? 63 : ST_WIDE;
                const int stn = t == 0 ? stores_in_flight : 0;
                if (stn == ST_NARROW) __builtin_amdgcn_s_waitcnt((SN & 15) | 0x70 | ((SN >> 4) << 14));
                else if (stn == ST_WIDE) __builtin_amdgcn_s_waitcnt((SW & 15) | 0x70 | ((SW >> 4) << 14));
                else __builtin_amdgcn_s_waitcnt(0x0070);
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- k-step 1: MFMAs on F1, DMA of stream tile q+2 into this stage, reads of F0 (k-step 0 of the next tile)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g < T::X_DMA) dma_piece_x(g, lds + cur);
                dma_piece_w(g, lds + cur);
                read_w(0, g, ws + nxt, p0);
                if (g < MT) read_x(0, g, xs + nxt, p0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1][g], fx[1][i], acc[g][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): F0 of the next K tile
            __builtin_amdgcn_sched_barrier(0);
            dma_advance();
            cur = nxt;
        }
        const bool full = cm0 + T::BM <= a.M && cn0 + T::BN <= a.N;
        const int mrow0 = cm0 + wm * (T::BM / 2), ncol0 = cn0 + wn * 128;
        if constexpr (EPI == EPI_QKROPE) {
            static_assert(EPI != EPI_QKROPE || MT == 8, "the fused q/k epilogue is written for 256-token tiles");
            epilogue256w4_qkrope(a, acc, mrow0, ncol0, lane, full);
        } else {
            epilogue_tile<EPI, MT>(a, *(const f32x4_t(*)[4][MT])&acc[0], mrow0, ncol0, lane, full);
            epilogue_tile<EPI, MT>(a, *(const f32x4_t(*)[4][MT])&acc[4], mrow0, ncol0 + 64, lane, full);
        }
        stores_in_flight = full ? (a.wide && EPI != EPI_F32 ? ST_WIDE : ST_NARROW) : 0;
    }
}

template <int EPI, int MT>
static int launch256w4(GemmArgs a, hipStream_t s) {
    using T = tile256w4::Tile<MT>;
    if (coda_ensure_dyn_lds((const void*)gemm256w4_bf16_kernel<EPI, MT>, T::LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    const int n_cu = coda_device_cus();
    a.tiles_m = (a.M + T::BM - 1) / T::BM;
    a.tiles_n = (a.N + T::BN - 1) / T::BN;
    const int grid = a.tiles_m * a.tiles_n < n_cu ? a.tiles_m * a.tiles_n : n_cu;
    hipLaunchKernelGGL((gemm256w4_bf16_kernel<EPI, MT>), dim3(grid), dim3(256), T::LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm256w4_bf16");
    return CODA_OK;
}
template <int EPI>
static int launch256w4_8(GemmArgs a, hipStream_t s) { return launch256w4<EPI, 8>(a, s); }
template <int EPI>
static int launch256w4_6(GemmArgs a, hipStream_t s) { return launch256w4<EPI, 6>(a, s); }

template <int EPI>
static int launch192x256p(GemmArgs a, hipStream_t s) { return launch256p<EPI, 3>(a, s); }
template <int EPI>
static int launch192x256px(GemmArgs a, hipStream_t s) { return launch256p<EPI, 3, true>(a, s); }
template <int EPI>
static int launch256px(GemmArgs a, hipStream_t s) { return launch256p<EPI, 4, true>(a, s); }

#ifdef CODA_CLOCK_PROBE
// diagnostic build only (tools/build_probe.sh): where the kernels leave their cycle stamps
static unsigned long long* g_gemm_probe = nullptr;
extern "C" int coda_gemm_probe(void* buf) {
    g_gemm_probe = (unsigned long long*)buf;
    return CODA_OK;
}
#endif

// `forced` (the engine argument of coda_gemm_bf16_ex): 0 automatic, 1 128x128, 2 256x256 (one tile per workgroup),
// 3 persistent 256x256, 4 persistent 256x128, 5 persistent 192x256 (3-stage weight ring), 6 persistent 192x256 with the
// 3-stage ring on the activations, 7 persistent 256x256 with a 3-stage activation ring (160 KiB of LDS),
// 8 / 9 four-wave persistent 256x256 / 192x256 (wave tile 128 / 96 tokens x 128 features, one wave per SIMD),
// 10 / 11 one 96x192 / 192x96 tile per CU for a few hundred rows (gemm_tile192.h).
// -> 0: 128x128 engine, otherwise the engine number above.  The choice is a pure function of the call's arguments.
static int pick_engine(const GemmArgs& a, int forced) {
    if (a.x_rows) return 0;
    // per-lane 32-bit DMA offsets: the operand panels of one launch must span < 4 GiB
    if ((size_t)a.M * a.ldx_bytes >= 0xffffffffull || (size_t)a.N * a.ldw_bytes >= 0xffffffffull) return 0;
    if (forced == 1) return 0;
    if (forced >= 2 && !(a.m_count && (forced == 2 || forced == 4))) return forced;
    if (a.M < 256 || a.N < 128) return 0;
    // one 512-thread workgroup per CU: prefer the engine whose tile count fills whole rounds of 256 CUs
    auto fill = [](long tiles) { const long rounds = (tiles + 255) / 256; return (double)tiles / (double)(rounds * 256); };
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const long t128 = (long)((a.M + 255) / 256) * ((a.N + 127) / 128);
    const long t192 = (long)((a.M + 191) / 192) * ((a.N + 255) / 256);
    // whole rounds of 256x256 tiles: the four-wave engine (one wave per SIMD, 128x128 wave tiles) is 2-5 % ahead of the
    // eight-wave one on these shapes in one-process A/B (gate|up 416-436 vs 429-445 us, plain qkv 140-150 vs 148-157 us)
    if (a.N >= 256 && t256 >= 200 && fill(t256) >= 0.85) return 8;
    // e.g. 12288 x 2048: 512 tiles = 2 full rounds.  The third ring stage goes to the activations: in the decoder
    // they are what the previous kernel just wrote (down_proj: 151 MB, beyond every cache), -4..5 % with cold operands
    if (a.N >= 256 && t192 >= 200 && fill(t192) >= 0.85) return 6;
    if (a.m_count) return a.N >= 256 && t256 >= 200 ? 3 : 0;     // counted launches: engines that read m_count
    if (t128 >= 200 && fill(t128) >= 0.85) return 4;
    // a few hundred rows x many features (B = 1 gate|up: 384 x 12288): the 128x128 engine would need a second, mostly
    // empty round (288 tiles); the 192x96 / 96x192 tiles give every CU exactly one (27 vs 34 us, profiles/r03_b1_bench.txt)
    const long t0 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (!a.m_count && t0 > 256 && fill(t0) < 0.75) {
        const long tall = (long)((a.M + 191) / 192) * ((a.N + 95) / 96), wide = (long)((a.M + 95) / 96) * ((a.N + 191) / 192);
        if (tall >= 200 && tall <= 256) return 11;
        if (wide >= 200 && wide <= 256) return 10;
    }
    return 0;
}

template <int EPI>
static int launch(const GemmArgs& a, hipStream_t s) {
    if (coda_ensure_dyn_lds((const void*)gemm_bf16_kernel<EPI>, LDS_BYTES) != CODA_OK) return CODA_ERR_LAUNCH;
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(a.tiles_m * a.tiles_n * a.ksplit), dim3(256), LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm_bf16");
    return CODA_OK;
}

extern "C" int coda_gemm_bf16_ex(const void* X, long long ldx, const int* x_rows, const void* W, long long ldw,
                                 void* C, long long ldc, const void* R, long long ldr,
                                 int M, int N, int K, int epilogue, int forced_engine, int flags, void* stream) {
    if (forced_engine < 0 || forced_engine > 11) return CODA_ERR_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0) return CODA_ERR_SHAPE;
    if (epilogue == EPI_SWIGLU && N % 32 != 0) return CODA_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && R == nullptr) return CODA_ERR_ARG;
    if (!X || !W || !C) return CODA_ERR_ARG;
    const int esz = epilogue == EPI_F32 ? 4 : 2;
    if (((uintptr_t)X | (uintptr_t)W) & 15 || (ldx * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    if (((uintptr_t)C) & (4 * esz - 1) || (ldc * esz) % (4 * esz)) return CODA_ERR_ALIGN;
    if (R && ((((uintptr_t)R) & 7) || (ldr * 2) % 8)) return CODA_ERR_ALIGN;
    GemmArgs a{};
    a.ksplit = 1;
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = x_rows;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)C; a.ldc_bytes = ldc * esz;
    a.R = (const char*)R; a.ldr_bytes = ldr * 2;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.wide = !(flags & CODA_GEMM_NARROW_STORES) && (((uintptr_t)C | (uintptr_t)(ldc * esz)) & 15) == 0 &&
             (!R || (((uintptr_t)R | (uintptr_t)(ldr * 2)) & 15) == 0);
#ifdef CODA_CLOCK_PROBE
    a.probe = g_gemm_probe;
#endif
    hipStream_t s = (hipStream_t)stream;
    const int engine = pick_engine(a, forced_engine);
#define CODA_DISPATCH(fn)                                      \
    switch (epilogue) {                                        \
        case EPI_PLAIN: return fn<EPI_PLAIN>(a, s);            \
        case EPI_RESIDUAL: return fn<EPI_RESIDUAL>(a, s);      \
        case EPI_SWIGLU: return fn<EPI_SWIGLU>(a, s);          \
        case EPI_F32: return fn<EPI_F32>(a, s);                \
        default: return CODA_ERR_ARG;                          \
    }
    if (engine == 11) { CODA_DISPATCH(launch192x96) }
    if (engine == 10) { CODA_DISPATCH(launch96x192) }
    if (engine == 9) { CODA_DISPATCH(launch256w4_6) }
    if (engine == 8) { CODA_DISPATCH(launch256w4_8) }
    if (engine == 7) { CODA_DISPATCH(launch256px) }
    if (engine == 6) { CODA_DISPATCH(launch192x256px) }
    if (engine == 5) { CODA_DISPATCH(launch192x256p) }
    if (engine == 4) { CODA_DISPATCH(launch256x128p) }
    if (engine == 3) { CODA_DISPATCH(launch256p) }
    if (engine == 2) { CODA_DISPATCH(launch256) }
#undef CODA_DISPATCH
    switch (epilogue) {
        case EPI_PLAIN: return launch<EPI_PLAIN>(a, s);
        case EPI_RESIDUAL: return launch<EPI_RESIDUAL>(a, s);
        case EPI_SWIGLU: return launch<EPI_SWIGLU>(a, s);
        case EPI_F32: return launch<EPI_F32>(a, s);
        default: return CODA_ERR_ARG;
    }
}

extern "C" int coda_gemm_bf16(const void* X, long long ldx, const int* x_rows, const void* W, long long ldw,
                              void* C, long long ldc, const void* R, long long ldr,
                              int M, int N, int K, int epilogue, void* stream) {
    return coda_gemm_bf16_ex(X, ldx, x_rows, W, ldw, C, ldc, R, ldr, M, N, K, epilogue, 0, 0, stream);
}


// Plain projection whose row count lives on the device (the candidate rows of the top-p / top-k path:
// generation_utils.py:462 + modeling_coda.py:411 restricted to `logits[mask_index]`).
extern "C" int coda_gemm_bf16_counted(const void* X, long long ldx, const int* m_count, const void* W, long long ldw, void* C,
                                      long long ldc, int M_max, int N, int K, void* stream) {
    if (M_max <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0) return CODA_ERR_SHAPE;
    if (!X || !W || !C || !m_count) return CODA_ERR_ARG;
    if (((uintptr_t)X | (uintptr_t)W) & 15 || (ldx * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    if (((uintptr_t)C) & 7 || (ldc * 2) % 8) return CODA_ERR_ALIGN;
    GemmArgs a{};
    a.ksplit = 1;
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = nullptr; a.m_count = m_count;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)C; a.ldc_bytes = ldc * 2;
    a.M = M_max; a.N = N; a.K = K;
    a.tiles_m = (M_max + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.wide = (((uintptr_t)C | (uintptr_t)(ldc * 2)) & 15) == 0;
    const int engine = pick_engine(a, 0);
    if (engine == 8) return launch256w4<EPI_PLAIN, 8>(a, (hipStream_t)stream);    // (reads m_count like the engines below)
    if (engine == 6) return launch192x256px<EPI_PLAIN>(a, (hipStream_t)stream);
    if (engine == 5) return launch192x256p<EPI_PLAIN>(a, (hipStream_t)stream);
    if (engine == 3) return launch256p<EPI_PLAIN>(a, (hipStream_t)stream);
    return launch<EPI_PLAIN>(a, (hipStream_t)stream);
}

// QKV projection with the q/k RMSNorm + rotary embedding fused into the epilogue (one launch instead of
// coda_gemm_bf16 + coda_qknorm_rope).  W must be packed with coda_amd.engine.permute_qk_rows.
extern "C" int coda_qkv_proj_rope(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                                  const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T, int L,
                                  int n_q, int n_kv, int K, float eps, void* stream) {
    const int N = (n_q + 2 * n_kv) * 128;
    if (T < 256 || L <= 0 || T % L || n_q <= 0 || n_kv <= 0 || K <= 0 || K % 64) return CODA_ERR_SHAPE;
    if ((n_q * 128) % 256 || (n_kv * 128) % 256 || ld < N) return CODA_ERR_SHAPE;
    if (!X || !W || !qkv || !q_weight || !k_weight || !cos || !sin) return CODA_ERR_ARG;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)qkv | (uintptr_t)q_weight | (uintptr_t)k_weight | (uintptr_t)cos | (uintptr_t)sin) & 15 ||
        (ldx * 2) % 16 || (ldw * 2) % 16 || (ld * 2) % 16)
        return CODA_ERR_ALIGN;
    if ((size_t)T * ldx * 2 >= 0xffffffffull || (size_t)N * ldw * 2 >= 0xffffffffull) return CODA_ERR_SHAPE;
    GemmArgs a{};
    a.ksplit = 1;
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = nullptr;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)qkv; a.ldc_bytes = ld * 2;
    a.R = nullptr; a.ldr_bytes = 0;
    a.M = T; a.N = N; a.K = K;
    a.wide = 1;                               // alignment checked above
    a.qn_w = (const char*)q_weight; a.kn_w = (const char*)k_weight;
    a.rope_cos = (const char*)cos; a.rope_sin = (const char*)sin;
    a.L = L; a.q_feat = n_q * 128; a.qk_feat = (n_q + n_kv) * 128; a.eps = eps;
    return launch256p<EPI_QKROPE>(a, (hipStream_t)stream);
}


// QKV projection + q/k-norm + rotary on the four-wave engine; W in its natural row order (q | k | v heads).
extern "C" int coda_qkv_proj_rope_w4(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                                     const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T, int L,
                                     int n_q, int n_kv, int K, float eps, void* stream) {
    const int N = (n_q + 2 * n_kv) * 128;
    if (T < 256 || L <= 0 || T % L || n_q <= 0 || n_kv <= 0 || K <= 0 || K % 64) return CODA_ERR_SHAPE;
    if ((n_q * 128) % 256 || (n_kv * 128) % 256 || ld < N) return CODA_ERR_SHAPE;
    if (!X || !W || !qkv || !q_weight || !k_weight || !cos || !sin) return CODA_ERR_ARG;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)qkv | (uintptr_t)q_weight | (uintptr_t)k_weight | (uintptr_t)cos | (uintptr_t)sin) & 15 ||
        (ldx * 2) % 16 || (ldw * 2) % 16 || (ld * 2) % 16)
        return CODA_ERR_ALIGN;
    if ((size_t)T * ldx * 2 >= 0xffffffffull || (size_t)N * ldw * 2 >= 0xffffffffull) return CODA_ERR_SHAPE;
    GemmArgs a{};
    a.ksplit = 1;
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = nullptr;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)qkv; a.ldc_bytes = ld * 2;
    a.R = nullptr; a.ldr_bytes = 0;
    a.M = T; a.N = N; a.K = K;
    a.wide = 1;                               // alignment checked above
    a.qn_w = (const char*)q_weight; a.kn_w = (const char*)k_weight;
    a.rope_cos = (const char*)cos; a.rope_sin = (const char*)sin;
    a.L = L; a.q_feat = n_q * 128; a.qk_feat = (n_q + n_kv) * 128; a.eps = eps;
    return launch256w4<EPI_QKROPE, 8>(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Split-K for small M (the B = 1 serving shape): at M = 384 a projection has 48-96 tiles of 128x128, so 48-96 of the 256
// CUs do all the loading, each through a 64 B/clk load path - the launch is bound by how many CUs take part, not by HBM.
// The K range is cut into `ksplit` pieces (ksplit x more workgroups, each loading 1 / ksplit as much), fp32 partial tiles go
// to a caller workspace, and this kernel sums them in a fixed order (deterministic) and applies the epilogue with the
// reference's rounding points (projection -> bf16, then residual add / SiLU * up).
template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, long long slab_floats, int ksplit,
                                                              int M, int N, char* __restrict__ C, long long ldc_bytes,
                                                              const char* __restrict__ R, long long ldr_bytes) {
    const int ncol4 = (EPI == EPI_SWIGLU ? N / 2 : N) / 4;            // 4 output columns per thread
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * ncol4) return;
    const int m = (int)(idx / ncol4), j = (int)(idx % ncol4) * 4;
    if constexpr (EPI == EPI_SWIGLU) {
        // weight rows are interleaved in blocks of 16: [gate 16 | up 16]
        const int gcol = 32 * (j / 16) + (j % 16);
        f32x4_t gsum = {0.f, 0.f, 0.f, 0.f}, usum = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ksplit; ++s) {
            const float* p = part + (size_t)s * slab_floats + (size_t)m * N + gcol;
            gsum += *(const f32x4_t*)p;
            usum += *(const f32x4_t*)(p + 16);
        }
        const unsigned g01 = pack_bf16x2(gsum[0], gsum[1]), g23 = pack_bf16x2(gsum[2], gsum[3]);
        const unsigned u01 = pack_bf16x2(usum[0], usum[1]), u23 = pack_bf16x2(usum[2], usum[3]);
        float s0 = silu_f32(bf16_lo(g01)), s1 = silu_f32(bf16_hi(g01)), s2 = silu_f32(bf16_lo(g23)), s3 = silu_f32(bf16_hi(g23));
        round_bf16_pair(s0, s1);
        round_bf16_pair(s2, s3);
        *(u32x2_t*)(C + (size_t)m * ldc_bytes + (size_t)j * 2) =
            (u32x2_t){pack_bf16x2(s0 * bf16_lo(u01), s1 * bf16_hi(u01)), pack_bf16x2(s2 * bf16_lo(u23), s3 * bf16_hi(u23))};
    } else {
        f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < ksplit; ++s) sum += *(const f32x4_t*)(part + (size_t)s * slab_floats + (size_t)m * N + j);
        unsigned v0 = pack_bf16x2(sum[0], sum[1]), v1 = pack_bf16x2(sum[2], sum[3]);
        if constexpr (EPI == EPI_RESIDUAL) {
            const u32x2_t r = *(const u32x2_t*)(R + (size_t)m * ldr_bytes + (size_t)j * 2);
            v0 = pack_bf16x2(bf16_lo(v0) + bf16_lo(r[0]), bf16_hi(v0) + bf16_hi(r[0]));
            v1 = pack_bf16x2(bf16_lo(v1) + bf16_lo(r[1]), bf16_hi(v1) + bf16_hi(r[1]));
        }
        *(u32x2_t*)(C + (size_t)m * ldc_bytes + (size_t)j * 2) = (u32x2_t){v0, v1};
    }
}

// ---- split-K epilogues fused with the row-wise op that FOLLOWS the projection in the decoder (B = 1: ~230 launches of
// 4-35 us per denoise step; each of these removes one).  Both mirror the thread -> element mapping and the summation order
// of the kernels they replace (splitk_epilogue_kernel, then rmsnorm_kernel / qknorm_rope_kernel), so the results are
// bit-identical to the two-launch sequence.
//
// o_proj / down_proj + residual (modeling_coda.py:247,253), then the NEXT RMSNorm (:36-41: post-attention norm, the next
// layer's input norm, or the final norm): one wave per token row, a lane owns 16-byte pieces c = lane + 64 i of the row.
template <int MAXC>
__global__ __launch_bounds__(64) void splitk_residual_rmsnorm_kernel(const float* __restrict__ part, long long slab_floats, int ksplit,
                                                                     int M, int N, char* __restrict__ C, long long ldc_bytes,
                                                                     const char* __restrict__ R, long long ldr_bytes,
                                                                     const u32x4_t* __restrict__ nw, char* __restrict__ normed,
                                                                     long long ldn_bytes, float eps) {
    const int lane = threadIdx.x, row = blockIdx.x;
    if (row >= M) return;
    const int chunks = N / 8;
    u32x4_t keep[MAXC];
    float ss = 0.f;
    // partial sums: slabs in ascending order per element (the order of splitk_epilogue_kernel), but with the loads of four
    // slabs x MAXC pieces in flight per lane - a lone wave per row has only its own loads to hide the latency with
    f32x4_t lo[MAXC], hi[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) lo[i] = hi[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float* prow = part + (size_t)row * N;
    for (int s0 = 0; s0 < ksplit; s0 += 4) {
        f32x4_t a[4][MAXC], b[4][MAXC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                const bool on = s0 + u < ksplit && c < chunks;
                // lanes / slabs that are off re-read piece 0 of slab s0 of THIS row: always inside the row (rows shorter than
                // 512 features leave lanes without a piece; an address derived from the lane would leave the row - and, on the
                // last row of a buffer, the allocation)
                const float* p = prow + (size_t)(on ? s0 + u : s0) * slab_floats + (on ? c : 0) * 8;
                a[u][i] = *(const f32x4_t*)p;
                b[u][i] = *(const f32x4_t*)(p + 4);
                if (!on) a[u][i] = b[u][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + u < ksplit) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    lo[i] += a[u][i];
                    hi[i] += b[u][i];
                }
            }
    }
    u32x4_t res[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        res[i] = *(const u32x4_t*)(R + (size_t)row * ldr_bytes + (size_t)(c < chunks ? c : 0) * 16);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            const u32x4_t r = res[i];
            u32x4_t v = {pack_bf16x2(lo[i][0], lo[i][1]), pack_bf16x2(lo[i][2], lo[i][3]), pack_bf16x2(hi[i][0], hi[i][1]),
                         pack_bf16x2(hi[i][2], hi[i][3])};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(bf16_lo(v[j]) + bf16_lo(r[j]), bf16_hi(v[j]) + bf16_hi(r[j]));
            *(u32x4_t*)(C + (size_t)row * ldc_bytes + (size_t)c * 16) = v;
            keep[i] = v;
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / sqrtf(ss / (float)N + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < chunks) {
            float f[8], g[8];
            unpack8(keep[i], f);
            unpack8(nw[c], g);
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = pack_bf16x2(g[2 * j] * round_bf16(f[2 * j] * inv), g[2 * j + 1] * round_bf16(f[2 * j + 1] * inv));
            *(u32x4_t*)(normed + (size_t)row * ldn_bytes + (size_t)c * 16) = o;
        }
    }
}

// The same for rows of at most 2048 features (the decoder's hidden size), with a WORKGROUP per row: 384 lone waves that
// each stream 64 KB of partial tiles left three quarters of the chip's wave slots empty (7.9 us for 25 MB).  Thread t owns
// piece t; all eight slabs' loads of a piece are in flight at once; the sum of squares is then formed by wave 0 from an
// LDS copy of the row in exactly the order of the one-wave kernel (lane l: pieces l, l + 64, l + 128, l + 192 in sequence,
// then the butterfly), so the two kernels - and rmsnorm_kernel behind splitk_epilogue_kernel - agree bit for bit.
__global__ __launch_bounds__(256) void splitk_residual_rmsnorm_row256_kernel(const float* __restrict__ part, long long slab_floats,
                                                                            int ksplit, int M, int N, char* __restrict__ C,
                                                                            long long ldc_bytes, const char* __restrict__ R,
                                                                            long long ldr_bytes, const u32x4_t* __restrict__ nw,
                                                                            char* __restrict__ normed, long long ldn_bytes, float eps) {
    __shared__ u32x4_t vrow[256];
    __shared__ float s_inv;
    const int tid = threadIdx.x, row = blockIdx.x;
    const int chunks = N / 8;
    const bool on = tid < chunks;
    const int c = on ? tid : 0;                       // idle threads re-read piece 0 of this row (always in bounds)
    const float* prow = part + (size_t)row * N + (size_t)c * 8;
    const u32x4_t r = *(const u32x4_t*)(R + (size_t)row * ldr_bytes + (size_t)c * 16);
    f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ksplit; s0 += 8) {
        f32x4_t a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* p = prow + (size_t)(s0 + u < ksplit ? s0 + u : s0) * slab_floats;
            a[u] = *(const f32x4_t*)p;
            b[u] = *(const f32x4_t*)(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < ksplit) {
                lo += a[u];
                hi += b[u];
            }
    }
    u32x4_t v = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(bf16_lo(v[j]) + bf16_lo(r[j]), bf16_hi(v[j]) + bf16_hi(r[j]));
    if (on) *(u32x4_t*)(C + (size_t)row * ldc_bytes + (size_t)c * 16) = v;
    vrow[tid] = v;
    __syncthreads();
    if (tid < 64) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = tid + 64 * i;
            if (ci < chunks) {
                float f[8];
                unpack8(vrow[ci], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
        }
        ss = wave_sum(ss);
        if (tid == 0) s_inv = 1.0f / sqrtf(ss / (float)N + eps);
    }
    __syncthreads();
    if (on) {
        const float inv = s_inv;
        float f[8], g[8];
        unpack8(v, f);
        unpack8(nw[c], g);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(g[2 * j] * round_bf16(f[2 * j] * inv), g[2 * j + 1] * round_bf16(f[2 * j + 1] * inv));
        *(u32x4_t*)(normed + (size_t)row * ldn_bytes + (size_t)c * 16) = o;
    }
}

// q/k/v projection (modeling_coda.py:123-125), then q_norm / k_norm + rotary (:128-148, modeling_utils.py:70-101) on the q
// and k heads; v heads are only rounded.  16 lanes per (token, head) row of 128 features, as qknorm_rope_kernel.
__global__ __launch_bounds__(256) void splitk_qknorm_rope_kernel(const float* __restrict__ part, long long slab_floats, int ksplit,
                                                                 int T, int N, char* __restrict__ qkv, long long ld_b,
                                                                 const u32x4_t* __restrict__ qw, const u32x4_t* __restrict__ kw,
                                                                 const u32x4_t* __restrict__ cs, const u32x4_t* __restrict__ sn,
                                                                 int L, int n_q, int n_qk, float eps) {
    const int sub = threadIdx.x & 15;
    const int n_heads = N / 128;                            // q, k and v heads
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long total = (long long)T * n_heads;
    const bool live = row < total;
    const long long rr = live ? row : total - 1;            // keep all lanes in the shuffles
    const int t = (int)(rr / n_heads), h = (int)(rr % n_heads);
    f32x4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    // ascending slab order per element, up to eight slabs' loads in flight (a loop of load - add - load serialises the
    // latencies: the trip count is a run-time value, the compiler does not batch the loads by itself)
    const float* p0 = part + (size_t)t * N + h * 128 + sub * 8;
    for (int s0 = 0; s0 < ksplit; s0 += 8) {
        f32x4_t a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* p = p0 + (size_t)(s0 + u < ksplit ? s0 + u : s0) * slab_floats;
            a[u] = *(const f32x4_t*)p;
            b[u] = *(const f32x4_t*)(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < ksplit) {
                lo += a[u];
                hi += b[u];
            }
    }
    const u32x4_t v = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
    u32x4_t* out = (u32x4_t*)(qkv + (size_t)t * ld_b + (size_t)h * 256) + sub;
    float f[8];
    unpack8(v, f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
    float w[8], c[8], sv[8], n[8], other[8];
    const int hh = min(h, n_qk - 1);
    unpack8((hh < n_q ? qw : kw)[sub], w);
    const int pos = t % L;
    unpack8(cs[(size_t)pos * 16 + sub], c);
    unpack8(sn[(size_t)pos * 16 + sub], sv);
#pragma unroll
    for (int j = 0; j < 8; ++j) n[j] = round_bf16(w[j] * round_bf16(f[j] * inv));
#pragma unroll
    for (int j = 0; j < 8; ++j) other[j] = __shfl_xor(n[j], 8, 64);
    const float sign = sub < 8 ? -1.0f : 1.0f;              // rotate_half: first half takes -x2, second half x1
    u32x4_t o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a0 = round_bf16(n[2 * i] * c[2 * i]) + round_bf16(sign * other[2 * i] * sv[2 * i]);
        const float a1 = round_bf16(n[2 * i + 1] * c[2 * i + 1]) + round_bf16(sign * other[2 * i + 1] * sv[2 * i + 1]);
        o[i] = pack_bf16x2(a0, a1);
    }
    if (live) *out = h < n_qk ? o : v;
}

// Which engine writes the partial tiles - a pure function of the call's (M, N, ksplit).  When the 192x96 tiles times the K
// ranges fit ONE workgroup per CU, that engine takes them (three-stage ring, 81 vs ~40 GB/s per CU through the load path
// at one workgroup per CU: profiles/r03b_t192_probe.txt); otherwise the 128x128 engine with up to two workgroups per CU.
// Every engine sums a K range in the same order, so the choice does not change a bit of the result
// (test_gemm_few_rows_picker_takes_the_one_tile_per_cu_engine, test_gemm_splitk_exact_integer).
static int launch_splitk_partials(const GemmArgs& a, hipStream_t s) {
    const long wgs = (long)((a.M + 191) / 192) * ((a.N + 95) / 96) * a.ksplit;
    if (a.M > 128 && wgs >= 128 && wgs <= 256) return launch192x96<EPI_F32>(a, s);
    return launch<EPI_F32>(a, s);
}

// shared front half of the split-K entry points: argument checks + the partial-tile GEMM into the workspace
static int splitk_partials(const void* X, long long ldx, const void* W, long long ldw, int M, int N, int K, int ksplit,
                           void* workspace, long long workspace_bytes, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || ksplit < 2 || N % 32 != 0 || K % ksplit != 0 || (K / ksplit) % BK != 0) return CODA_ERR_SHAPE;
    if (!X || !W || !workspace) return CODA_ERR_ARG;
    if ((long long)ksplit * M * N * 4 > workspace_bytes) return CODA_ERR_SHAPE;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)workspace) & 15 || (ldx * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    GemmArgs a{};
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = nullptr;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)workspace; a.ldc_bytes = (long long)N * 4;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.wide = 0;
    a.ksplit = ksplit; a.slab_bytes = (long long)M * N * 4;
    return launch_splitk_partials(a, s);
}

extern "C" int coda_gemm_bf16_splitk_norm(const void* X, long long ldx, const void* W, long long ldw, void* C, long long ldc,
                                          const void* R, long long ldr, int M, int N, int K, int ksplit, void* workspace,
                                          long long workspace_bytes, const void* norm_weight, void* normed, long long ld_normed,
                                          float eps, void* stream) {
    if (!C || !R || !norm_weight || !normed) return CODA_ERR_ARG;
    if (N % 8 || N > 8 * 64 * 8) return CODA_ERR_SHAPE;                  // a lane keeps at most 8 pieces of the row in registers
    if (((uintptr_t)C | (uintptr_t)R | (uintptr_t)norm_weight | (uintptr_t)normed) & 15 || (ldc * 2) % 16 || (ldr * 2) % 16 ||
        (ld_normed * 2) % 16)
        return CODA_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int rc = splitk_partials(X, ldx, W, ldw, M, N, K, ksplit, workspace, workspace_bytes, s);
    if (rc != CODA_OK) return rc;
    const float* part = (const float*)workspace;
    const long long slab = (long long)M * N;
    if (N <= 256 * 8)
        hipLaunchKernelGGL(splitk_residual_rmsnorm_row256_kernel, dim3(M), dim3(256), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2,
                           (const char*)R, ldr * 2, (const u32x4_t*)norm_weight, (char*)normed, ld_normed * 2, eps);
    else if (N <= 4 * 64 * 8)
        hipLaunchKernelGGL(splitk_residual_rmsnorm_kernel<4>, dim3(M), dim3(64), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2,
                           (const char*)R, ldr * 2, (const u32x4_t*)norm_weight, (char*)normed, ld_normed * 2, eps);
    else
        hipLaunchKernelGGL(splitk_residual_rmsnorm_kernel<8>, dim3(M), dim3(64), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2,
                           (const char*)R, ldr * 2, (const u32x4_t*)norm_weight, (char*)normed, ld_normed * 2, eps);
    CODA_CHECK_LAUNCH("splitk_residual_rmsnorm");
    return CODA_OK;
}

extern "C" int coda_qkv_proj_rope_splitk(const void* X, long long ldx, const void* W, long long ldw, void* qkv, long long ld,
                                         const void* q_weight, const void* k_weight, const void* cos, const void* sin, int T,
                                         int L, int n_q, int n_kv, int K, float eps, int ksplit, void* workspace,
                                         long long workspace_bytes, void* stream) {
    const int N = (n_q + 2 * n_kv) * 128;
    if (T <= 0 || L <= 0 || T % L || n_q <= 0 || n_kv <= 0 || ld < N) return CODA_ERR_SHAPE;
    if (!qkv || !q_weight || !k_weight || !cos || !sin) return CODA_ERR_ARG;
    if (((uintptr_t)qkv | (uintptr_t)q_weight | (uintptr_t)k_weight | (uintptr_t)cos | (uintptr_t)sin) & 15 || (ld * 2) % 16)
        return CODA_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int rc = splitk_partials(X, ldx, W, ldw, T, N, K, ksplit, workspace, workspace_bytes, s);
    if (rc != CODA_OK) return rc;
    const long long rows = (long long)T * (N / 128);
    hipLaunchKernelGGL(splitk_qknorm_rope_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, (const float*)workspace,
                       (long long)T * N, ksplit, T, N, (char*)qkv, ld * 2, (const u32x4_t*)q_weight, (const u32x4_t*)k_weight,
                       (const u32x4_t*)cos, (const u32x4_t*)sin, L, n_q, n_q + n_kv, eps);
    CODA_CHECK_LAUNCH("splitk_qknorm_rope");
    return CODA_OK;
}

extern "C" int coda_gemm_bf16_splitk(const void* X, long long ldx, const void* W, long long ldw, void* C, long long ldc,
                                     const void* R, long long ldr, int M, int N, int K, int epilogue, int ksplit,
                                     void* workspace, long long workspace_bytes, void* stream) {
    if (ksplit <= 1) return coda_gemm_bf16(X, ldx, nullptr, W, ldw, C, ldc, R, ldr, M, N, K, epilogue, stream);
    if (M <= 0 || N <= 0 || K <= 0 || N % 32 != 0 || K % ksplit != 0 || (K / ksplit) % BK != 0) return CODA_ERR_SHAPE;
    if (epilogue != EPI_PLAIN && epilogue != EPI_RESIDUAL && epilogue != EPI_SWIGLU) return CODA_ERR_ARG;
    if (epilogue == EPI_RESIDUAL && R == nullptr) return CODA_ERR_ARG;
    if (!X || !W || !C || !workspace) return CODA_ERR_ARG;
    if ((long long)ksplit * M * N * 4 > workspace_bytes) return CODA_ERR_SHAPE;
    if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)workspace) & 15 || (ldx * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    if (((uintptr_t)C & 7) || (ldc * 2) % 8 || (R && ((((uintptr_t)R) & 7) || (ldr * 2) % 8))) return CODA_ERR_ALIGN;
    GemmArgs a{};
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = nullptr;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)workspace; a.ldc_bytes = (long long)N * 4;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    a.wide = 0;
    a.ksplit = ksplit; a.slab_bytes = (long long)M * N * 4;
    hipStream_t s = (hipStream_t)stream;
    const int rc = launch_splitk_partials(a, s);
    if (rc != CODA_OK) return rc;
    const long long threads = (long long)M * ((epilogue == EPI_SWIGLU ? N / 2 : N) / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    const float* part = (const float*)workspace;
    const long long slab = (long long)M * N;
    if (epilogue == EPI_SWIGLU) hipLaunchKernelGGL(splitk_epilogue_kernel<EPI_SWIGLU>, grid, dim3(256), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2, (const char*)R, ldr * 2);
    else if (epilogue == EPI_RESIDUAL) hipLaunchKernelGGL(splitk_epilogue_kernel<EPI_RESIDUAL>, grid, dim3(256), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2, (const char*)R, ldr * 2);
    else hipLaunchKernelGGL(splitk_epilogue_kernel<EPI_PLAIN>, grid, dim3(256), 0, s, part, slab, ksplit, M, N, (char*)C, ldc * 2, (const char*)R, ldr * 2);
    CODA_CHECK_LAUNCH("splitk_epilogue");
    return CODA_OK;
}
