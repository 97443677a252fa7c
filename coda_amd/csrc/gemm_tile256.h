// 256x256x64 (and 192x256x64) bf16 MFMA tile engine (8 waves, one workgroup per CU, 128 / 112 KiB LDS).
//
// Why a second engine: at ~1 PFLOP/s the 128x128 engine pulls ~50 B/clk/CU out of L2, close to
// what the L2 delivers; a 256x256 tile halves the L2 bytes per FLOP.
//
// Orientation and LDS image are those of gemm_tile128.h (weight fragment = MFMA operand A,
// 128-B rows, 16-B chunks XOR-swizzled with row&7 on the DMA source side).
//
// Pipeline (per wave: 128 tokens x 64 features = acc[4 nt][8 mt], 128 accumulator registers):
// each 32-deep k-step is consumed as four QUADRANT phases of 8 MFMAs (2 nt x 4 mt).  Quadrants are
// walked in a Gray-code cycle so consecutive phases share one operand set; the fragments of the
// next phase are read from LDS while the current phase's MFMAs run (register sets W0/W1 of 8 and
// X0/X1 of 16 VGPRs).  There is ONE barrier per K tile (64 deep), placed after the sixth of the
// eight phases, when every LDS read of the current tile has retired:
//     ... P5 | lgkmcnt(0) vmcnt(0) s_barrier | issue LDS-DMA of tile t+2 | P6 | P7 ...
// After it, (a) tile t+1 (whose DMA was issued one full tile earlier) is visible to every wave, so
// P6/P7 prefetch the first fragments of tile t+1 from the other buffer, and (b) the current buffer
// is dead, so the DMA of tile t+2 is issued into it with a whole tile of time to land.
#pragma once
#include "coda_common.h"
#include <type_traits>

namespace tile256 {

// dma_row/dma_chunk: one LDS-DMA wave-instruction moves 8 rows of 128 B; lane -> (row lane>>3, 16-B chunk lane&7),
// the chunk XOR-swizzled with the row on the global side
__device__ __forceinline__ int dma_row(int i, int wave, int lane) { return (i * 8 + wave) * 8 + (lane >> 3); }
__device__ __forceinline__ int dma_chunk(int lane) { return (lane & 7) ^ (lane >> 3); }

// Retire the LDS reads issued during the previous phase BEFORE this phase issues its own, so the new
// reads stay in flight under this phase's MFMAs (hipcc would otherwise wait lgkmcnt(0) after them).
__device__ __forceinline__ void phase_begin() {
#ifndef CODA_DIAG_NO_LGKM                        // diagnostic builds only (tools/clock_probe.sh): wrong results, cycle accounting
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only
#endif
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void tile_sync() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// MTH = 16-token groups per wave and quadrant: 4 -> 256-token tile, 3 -> 192-token tile (same 256 features;
// used when 256-token tiles would leave the last round of the 256 CUs half empty, e.g. 12288 x 2048 outputs)
template <int MTH>
struct Tile {
    static constexpr int MT = 2 * MTH;                   // token groups per wave
    static constexpr int BM = 2 * MT * 16, BN = 256, BK = 64;
    static constexpr int X_BYTES = BM * BK * 2;          // 32 / 24 KiB
    static constexpr int W_BYTES = BN * BK * 2;          // 32 KiB
    static constexpr int OP_BYTES = X_BYTES;             // offset of the W tile inside a stage
    static constexpr int STAGE_BYTES = X_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;    // 128 / 112 KiB
    static constexpr int X_DMA = BM / 64;                // LDS-DMA instructions per wave for the X tile (4 / 3)

    struct Src {
        const char* xbase; const char* wbase;   // wave-uniform operand bases
        unsigned xoff[X_DMA], woff[4];          // per-lane byte offsets of the DMA instructions per operand
    };

    static __device__ __forceinline__ void stage(const Src& s, size_t kbyte, char* lds_stage, int wave) {
#pragma unroll
        for (int i = 0; i < X_DMA; ++i) glds16(s.xbase + kbyte + s.xoff[i], lds_stage + (i * 8 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(s.wbase + kbyte + s.woff[i], lds_stage + OP_BYTES + (i * 8 + wave) * 1024);
    }

    struct Frags {
        bf16x8_t w[2][2];     // [set][nt within half]   one 32-deep k-step
        bf16x8_t x[2][MTH];   // [set][mt within half]
    };

    template <int SET, int NH>
    static __device__ __forceinline__ void load_w(Frags& f, const char* ws, int pos) {
#pragma unroll
        for (int j = 0; j < 2; ++j) f.w[SET][j] = *(const bf16x8_t*)(ws + (NH * 2 + j) * 2048 + pos);
    }

    template <int SET, int MH>
    static __device__ __forceinline__ void load_x(Frags& f, const char* xs, int pos) {
#pragma unroll
        for (int i = 0; i < MTH; ++i) f.x[SET][i] = *(const bf16x8_t*)(xs + (MH * MTH + i) * 2048 + pos);
    }

    // 2*MTH MFMAs: one quadrant (2 nt x MTH mt) of the wave's tile for one 32-deep k-step
    template <int WSET, int XSET, int NH, int MH>
    static __device__ __forceinline__ void quad(const Frags& f, f32x4_t (&acc)[4][MT]) {
#ifndef CODA_DIAG_NO_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MTH; ++i)
                acc[NH * 2 + j][MH * MTH + i] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.w[WSET][j], f.x[XSET][i], acc[NH * 2 + j][MH * MTH + i], 0, 0, 0);
#ifndef CODA_DIAG_NO_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }

    // acc[nt][mt]; wave (wm = wave>>2: token half, wn = wave&3: feature quarter).
    // A K tile = two 32-deep k-steps; k-step 0 runs quadrant pattern A, k-step 1 pattern B:
    //   A (enters W0=W(n0), X0=X(m0)):  (n0,m0) (n1,m0) (n1,m1) (n0,m1)   leaves W1, X0 prefetched for B
    //   B (enters W1=W(n1), X0=X(m0)):  (n1,m0) (n0,m0) | sync | (n0,m1) (n1,m1)   leaves W0, X0 prefetched for A
    // Every phase issues the LDS reads of the operand set the NEXT phase needs, into the register set
    // the running phase no longer reads.
    static __device__ __forceinline__ void mainloop(const Src& s, int K, char* lds, int wave, int lane, f32x4_t (&acc)[4][MT]) {
        const int wm = wave >> 2, wn = wave & 3;
        const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
        const int p0 = ((0 * 4 + kq) ^ sw) * 16, p1 = ((1 * 4 + kq) ^ sw) * 16;
        const char* xs = lds + (wm * (BM / 2) + r16) * 128;
        const char* ws = lds + OP_BYTES + (wn * 64 + r16) * 128;
        const int nk = K / BK;
        Frags f;

        stage(s, 0, lds, wave);
        if (nk > 1) stage(s, (size_t)BK * 2, lds + STAGE_BYTES, wave);
        tile_sync();
        load_w<0, 0>(f, ws, p0);
        load_x<0, 0>(f, xs, p0);

        auto tile = [&](int t, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const int cur = (t & 1) * STAGE_BYTES, nxt = STAGE_BYTES - cur;
            const char* xc = xs + cur;
            const char* wc = ws + cur;
            // ---- k-step 0, pattern A
            phase_begin(); load_w<1, 1>(f, wc, p0); quad<0, 0, 0, 0>(f, acc);
            phase_begin(); load_x<1, 1>(f, xc, p0); quad<1, 0, 1, 0>(f, acc);
            phase_begin(); load_x<0, 0>(f, xc, p1); quad<1, 1, 1, 1>(f, acc);
            phase_begin(); load_w<1, 1>(f, wc, p1); quad<0, 1, 0, 1>(f, acc);
            // ---- k-step 1, pattern B
            phase_begin(); load_w<0, 0>(f, wc, p1); quad<1, 0, 1, 0>(f, acc);
            phase_begin(); load_x<1, 1>(f, xc, p1); quad<0, 0, 0, 0>(f, acc);
            if constexpr (!LAST) {
                tile_sync();                          // tile t fully read by every wave; tile t+1 (DMA issued a tile ago) landed
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < nk) stage(s, (size_t)(t + 2) * BK * 2, lds + cur, wave);
                load_x<0, 0>(f, xs + nxt, p0);
                quad<0, 1, 0, 1>(f, acc);
                phase_begin();
                load_w<0, 0>(f, ws + nxt, p0);
                quad<1, 1, 1, 1>(f, acc);
            } else {
                phase_begin();
                quad<0, 1, 0, 1>(f, acc);
                quad<1, 1, 1, 1>(f, acc);
            }
        };
        for (int t = 0; t + 1 < nk; ++t) tile(t, std::false_type{});
        tile(nk - 1, std::true_type{});
    }
};

}  // namespace tile256
