// bf16 projection GEMMs (QKV / O / gate|up / down / plain lm-head rows) on CDNA4 MFMA.
// Replaces the nn.Linear calls of the reference decoder (modeling_coda.py:58-60,123-125,155)
// plus the ops fused behind them: residual add (:247,:253) and SiLU(gate)*up (:59).
#include "gemm_tile128.h"
#include "gemm_tile256.h"
#include "coda_abi_internal.h"

using namespace tile128;

enum { EPI_PLAIN = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_F32 = 3 };

struct GemmArgs {
    const char* X; long long ldx_bytes; const int* x_rows;   // x_rows: optional row gather (nullable)
    const char* W; long long ldw_bytes;
    char* C; long long ldc_bytes;
    const char* R; long long ldr_bytes;
    int M, N, K;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware + grouped tile order: consecutive ids within an XCD chunk walk 8 row-tiles of one
    // column-tile group, so the W panel and the X panels stay hot in that XCD's L2.
    const int nwg = a.tiles_m * a.tiles_n;
    const int id = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * a.tiles_n;
    const int g = id / per_group;
    const int gm0 = g * GROUP_M;
    const int gsz = min(a.tiles_m - gm0, GROUP_M);
    const int tm = gm0 + (id % per_group) % gsz;
    const int tn = (id % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

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

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    mainloop(f, a.K, lds, wave, lane, acc);

    // epilogue: lane owns token (lane&15) of each 16-token group, features (lane>>4)*4 .. +3 of each 16-feature group
    const int tok = lane & 15, fq = (lane >> 4) * 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + tok;
        if (m >= a.M) continue;
        if constexpr (EPI == EPI_SWIGLU) {
            // weight rows are interleaved in blocks of 16: [gate 16 | up 16] -> even nt = gate, odd nt = up
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int n = (n0 + wn * 64) / 2 + p * 16 + fq;     // output feature (of N/2)
                if (n >= a.N / 2) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gte = round_bf16(acc[2 * p][mt][r]);
                    const float up = round_bf16(acc[2 * p + 1][mt][r]);
                    o[r] = round_bf16(silu_f32(gte)) * up;
                }
                u32x2_t v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *(u32x2_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 2) = v;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + wn * 64 + nt * 16 + fq;
                if (n >= a.N) continue;
                if constexpr (EPI == EPI_F32) {
                    *(f32x4_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 4) = acc[nt][mt];
                } else {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[nt][mt][r];
                    if constexpr (EPI == EPI_RESIDUAL) {
                        const u32x2_t rv = *(const u32x2_t*)(a.R + (size_t)m * a.ldr_bytes + (size_t)n * 2);
                        o[0] = round_bf16(o[0]) + bf16_bits_to_f32(rv[0] & 0xffff);
                        o[1] = round_bf16(o[1]) + bf16_bits_to_f32(rv[0] >> 16);
                        o[2] = round_bf16(o[2]) + bf16_bits_to_f32(rv[1] & 0xffff);
                        o[3] = round_bf16(o[3]) + bf16_bits_to_f32(rv[1] >> 16);
                    }
                    u32x2_t v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *(u32x2_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 2) = v;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// 256x256 engine (gemm_tile256.h): same epilogues, no row gather.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmArgs a) {
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
    const int m0 = tm * tile256::BM, n0 = tn * tile256::BN;

    tile256::Src src;
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

    tile256::mainloop(src, a.K, lds, wave, lane, acc);

    const int tok = lane & 15, fq = (lane >> 4) * 4;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + wm * 128 + mt * 16 + tok;
        if (m >= a.M) continue;
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int n = (n0 + wn * 64) / 2 + p * 16 + fq;
                if (n >= a.N / 2) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gte = round_bf16(acc[2 * p][mt][r]);
                    const float up = round_bf16(acc[2 * p + 1][mt][r]);
                    o[r] = round_bf16(silu_f32(gte)) * up;
                }
                u32x2_t v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *(u32x2_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 2) = v;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + wn * 64 + nt * 16 + fq;
                if (n >= a.N) continue;
                if constexpr (EPI == EPI_F32) {
                    *(f32x4_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 4) = acc[nt][mt];
                } else {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[nt][mt][r];
                    if constexpr (EPI == EPI_RESIDUAL) {
                        const u32x2_t rv = *(const u32x2_t*)(a.R + (size_t)m * a.ldr_bytes + (size_t)n * 2);
                        o[0] = round_bf16(o[0]) + bf16_bits_to_f32(rv[0] & 0xffff);
                        o[1] = round_bf16(o[1]) + bf16_bits_to_f32(rv[0] >> 16);
                        o[2] = round_bf16(o[2]) + bf16_bits_to_f32(rv[1] & 0xffff);
                        o[3] = round_bf16(o[3]) + bf16_bits_to_f32(rv[1] >> 16);
                    }
                    u32x2_t v = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *(u32x2_t*)(a.C + (size_t)m * a.ldc_bytes + (size_t)n * 2) = v;
                }
            }
        }
    }
}

template <int EPI>
static int launch256(GemmArgs a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm256_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                tile256::LDS_BYTES) != hipSuccess)
            return CODA_ERR_LAUNCH;
        attr_done = true;
    }
    a.tiles_m = (a.M + tile256::BM - 1) / tile256::BM;
    a.tiles_n = (a.N + tile256::BN - 1) / tile256::BN;
    hipLaunchKernelGGL(gemm256_bf16_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(512), tile256::LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm256_bf16");
    return CODA_OK;
}

static int g_gemm_variant = 0;   // 0 auto, 1 force 128x128 engine, 2 force 256x256 engine
extern "C" int coda_gemm_select(int variant) {
    if (variant < 0 || variant > 2) return CODA_ERR_ARG;
    g_gemm_variant = variant;
    return CODA_OK;
}

static bool use_tile256(const GemmArgs& a) {
    if (a.x_rows) return false;
    // per-lane 32-bit DMA offsets: the operand panels of one launch must span < 4 GiB
    if ((size_t)a.M * a.ldx_bytes >= 0xffffffffull || (size_t)a.N * a.ldw_bytes >= 0xffffffffull) return false;
    if (g_gemm_variant == 1) return false;
    if (g_gemm_variant == 2) return true;
    const long tiles = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    // one 512-thread workgroup per CU: want whole waves of 256 tiles without a long tail
    const long rounds = (tiles + 255) / 256;
    return a.M >= 256 && a.N >= 256 && tiles >= 200 && (double)tiles / (double)(rounds * 256) >= 0.85;
}

template <int EPI>
static int launch(const GemmArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return CODA_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(a.tiles_m * a.tiles_n), dim3(256), LDS_BYTES, s, a);
    CODA_CHECK_LAUNCH("gemm_bf16");
    return CODA_OK;
}

extern "C" int coda_gemm_bf16(const void* X, long long ldx, const int* x_rows, const void* W, long long ldw,
                              void* C, long long ldc, const void* R, long long ldr,
                              int M, int N, int K, int epilogue, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0 || N % 4 != 0) return CODA_ERR_SHAPE;
    if (epilogue == EPI_SWIGLU && N % 32 != 0) return CODA_ERR_SHAPE;
    if (epilogue == EPI_RESIDUAL && R == nullptr) return CODA_ERR_ARG;
    if (!X || !W || !C) return CODA_ERR_ARG;
    const int esz = epilogue == EPI_F32 ? 4 : 2;
    if (((uintptr_t)X | (uintptr_t)W) & 15 || (ldx * 2) % 16 || (ldw * 2) % 16) return CODA_ERR_ALIGN;
    if (((uintptr_t)C) & (4 * esz - 1) || (ldc * esz) % (4 * esz)) return CODA_ERR_ALIGN;
    if (R && ((((uintptr_t)R) & 7) || (ldr * 2) % 8)) return CODA_ERR_ALIGN;
    GemmArgs a;
    a.X = (const char*)X; a.ldx_bytes = ldx * 2; a.x_rows = x_rows;
    a.W = (const char*)W; a.ldw_bytes = ldw * 2;
    a.C = (char*)C; a.ldc_bytes = ldc * esz;
    a.R = (const char*)R; a.ldr_bytes = ldr * 2;
    a.M = M; a.N = N; a.K = K;
    a.tiles_m = (M + BM - 1) / BM; a.tiles_n = (N + BN - 1) / BN;
    hipStream_t s = (hipStream_t)stream;
    if (use_tile256(a)) {
        switch (epilogue) {
            case EPI_PLAIN: return launch256<EPI_PLAIN>(a, s);
            case EPI_RESIDUAL: return launch256<EPI_RESIDUAL>(a, s);
            case EPI_SWIGLU: return launch256<EPI_SWIGLU>(a, s);
            case EPI_F32: return launch256<EPI_F32>(a, s);
            default: return CODA_ERR_ARG;
        }
    }
    switch (epilogue) {
        case EPI_PLAIN: return launch<EPI_PLAIN>(a, s);
        case EPI_RESIDUAL: return launch<EPI_RESIDUAL>(a, s);
        case EPI_SWIGLU: return launch<EPI_SWIGLU>(a, s);
        case EPI_F32: return launch<EPI_F32>(a, s);
        default: return CODA_ERR_ARG;
    }
}
