// 256(tokens) x 128(features) x 64 bf16 MFMA tile engine for the narrow projections (N = 2048: o_proj,
// down_proj), where 256x256 tiles would leave half of the CUs idle in the last round (384 tiles on
// 256 CUs) while 256x128 gives exactly 3 tiles per CU.  Same LDS image, DMA scheme and MFMA
// orientation as gemm_tile256.h; 8 waves as 4 (token quarters) x 2 (feature halves), each wave owns
// 64 tokens x 64 features = acc[4 nt][4 mt].
//
// A K tile (64 deep) is four phases of 8 MFMAs (one feature half x 4 token groups x one k-step):
//   enters W0 = W(n0,k0), X0 = X(k0)
//   P1  read W(n1,k0)->W1              mma(W0, X0 -> n0)
//   P2  read X(k1)->X1, W(n0,k1)->W0   mma(W1, X0 -> n1)
//   P3  read W(n1,k1)->W1              mma(W0, X1 -> n0)
//   ---- all LDS reads of the tile retired: counted vmcnt + barrier, LDS-DMA of tile t+3 into this stage (3-stage ring)
//   P4  read X'(k0)->X0, W'(n0,k0)->W0 from the other stage;  mma(W1, X1 -> n1)
#pragma once
#include "gemm_tile256.h"

namespace tile256x128 {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int X_BYTES = 256 * BK * 2;             // 32 KiB
constexpr int W_BYTES = 128 * BK * 2;             // 16 KiB
constexpr int STAGE_BYTES = X_BYTES + W_BYTES;    // 48 KiB
constexpr int LDS_BYTES = 3 * STAGE_BYTES;        // 144 KiB: 3-stage ring

struct Src {
    const char* xbase; const char* wbase;
    unsigned xoff[4], woff[2];
};

__device__ __forceinline__ void stage(const Src& s, size_t kbyte, char* lds_stage, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(s.xbase + kbyte + s.xoff[i], lds_stage + (i * 8 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(s.wbase + kbyte + s.woff[i], lds_stage + X_BYTES + (i * 8 + wave) * 1024);
}

using Frags = tile256::Tile<4>::Frags;
using tile256::phase_begin;
using tile256::tile_sync;

template <int SET, int NH>
__device__ __forceinline__ void load_w(Frags& f, const char* ws, int pos) {
#pragma unroll
    for (int j = 0; j < 2; ++j) f.w[SET][j] = *(const bf16x8_t*)(ws + (NH * 2 + j) * 2048 + pos);
}

template <int SET>
__device__ __forceinline__ void load_x(Frags& f, const char* xs, int pos) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.x[SET][i] = *(const bf16x8_t*)(xs + i * 2048 + pos);
}

template <int WSET, int XSET, int NH>
__device__ __forceinline__ void half(const Frags& f, f32x4_t (&acc)[4][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[NH * 2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[WSET][j], f.x[XSET][i], acc[NH * 2 + j][i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

}  // namespace tile256x128
