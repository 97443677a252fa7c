// 128x128x64 bf16 MFMA tile engine shared by the projection GEMMs and the fused lm-head kernel.
//
//   out[m][n] = sum_k X[m][k] * W[n][k]        X: activations [M,K], W: nn.Linear weight [N,K]
//
// Both operands are K-contiguous, so one staging routine serves both.  The MFMA is issued with
// the WEIGHT fragment as operand A and the ACTIVATION fragment as operand B, i.e. each wave
// computes a tile of out^T: accumulator rows = output features n, columns = tokens m.  A lane then
// owns 4 consecutive features of one token, which makes the epilogue stores 8 bytes wide and lets
// per-token reductions over features (SwiGLU pairing, softmax statistics) stay mostly in-lane.
//
// LDS image per operand tile: 128 rows x 128 B (64 bf16), filled by global_load_lds_dwordx4
// (LDS-DMA, 1 KiB per wave instruction = 8 rows).  The DMA destination is lane-linear, so the
// bank-conflict swizzle is applied on the SOURCE address: LDS slot (row r, 16-B position p) holds
// global chunk p ^ (r & 7); fragment reads apply the same XOR (conflict-free ds_read_b128).
#pragma once
#include "coda_common.h"

namespace tile128 {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // X tile + W tile
constexpr int LDS_BYTES = 2 * STAGE_BYTES;        // double buffered: 64 KiB

struct Frag {
    // per-lane global source pointers (bytes) for the 4 LDS-DMA instructions of each operand,
    // already offset to this lane's (row, swizzled chunk); advance by k0*2 bytes per K tile
    const char* x[4];
    const char* w[4];
    // bit i: DMA instruction i of the ACTIVATION tile carries rows of the problem (wave-uniform).  A tile that sticks out
    // of a short matrix (32 candidate rows in a 128-row tile) would otherwise re-read its last row 96 times per K tile
    // through the CU's load path - the path these small-M launches are bound by; the rows it leaves stale in LDS only
    // feed accumulator rows nobody reads
    unsigned xlive = 0xfu;
};

// xlive of a tile whose first row is row0 in a matrix of `rows` rows
__device__ __forceinline__ unsigned dma_live_mask(int row0, int rows, int wave) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) m |= (row0 + (i * 4 + wave) * 8 < rows ? 1u : 0u) << i;
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

// row_of(i): tile-local row handled by this lane in DMA instruction i (4 waves x 4 instr x 8 rows)
__device__ __forceinline__ int dma_row(int i, int wave, int lane) { return (i * 4 + wave) * 8 + (lane >> 3); }
__device__ __forceinline__ int dma_chunk(int lane) { return (lane & 7) ^ (lane >> 3); }

__device__ __forceinline__ void stage(const Frag& f, size_t kbyte, char* lds_stage, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (f.xlive >> i & 1) glds16(f.x[i] + kbyte, lds_stage + (i * 4 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(f.w[i] + kbyte, lds_stage + TILE_BYTES + (i * 4 + wave) * 1024);
}

// acc[nt][mt]: 16 features x 16 tokens each; wave (wm, wn) owns tokens wm*64.., features wn*64..
__device__ __forceinline__ void compute(const char* lds_stage, int wm, int wn, int lane, f32x4_t (&acc)[4][4]) {
    const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
    const char* xs = lds_stage + (wm * 64 + r16) * 128;
    const char* ws = lds_stage + TILE_BYTES + (wn * 64 + r16) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pos = ((kk * 4 + kq) ^ sw) * 16;
        bf16x8_t wf[4], xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wf[t] = *(const bf16x8_t*)(ws + t * 16 * 128 + pos);
            xf[t] = *(const bf16x8_t*)(xs + t * 16 * 128 + pos);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
}

// Full K loop for one output tile.  One barrier per K tile; the DMA for tile t+1 is in flight
// while tile t is multiplied.
__device__ __forceinline__ void mainloop(const Frag& f, int K, char* lds, int wave, int lane, f32x4_t (&acc)[4][4]) {
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / BK;
    stage(f, 0, lds, wave);
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();     // tile t landed everywhere; every wave is done reading the other buffer
        if (t + 1 < nk) stage(f, (size_t)(t + 1) * BK * 2, lds + ((t + 1) & 1) * STAGE_BYTES, wave);
        compute(lds + (t & 1) * STAGE_BYTES, wm, wn, lane, acc);
    }
    __syncthreads();         // LDS may be reused by the caller (next tile's prologue DMA)
}

}  // namespace tile128
