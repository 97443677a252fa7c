// 96x192x64 / 192x96x64 bf16 MFMA tile engine with a three-stage LDS ring, for projections with FEW token rows and many
// features (the B = 1 decoder: 384 tokens x 12288 gate|up features).
//
// Why another tile shape: with a few hundred rows a launch has about as many tiles as the chip has CUs, one workgroup
// per CU, and the time of the launch is the time ONE CU needs to pull its tile's operand panels through its LDS-DMA
// path (56-60 GB/s per CU with every CU pulling, 81 on an idle chip: profiles/r03b_t192_probe.txt) - not HBM, not MFMA.  Two things then matter:
//   * every CU gets exactly one tile (the 128x128 engine makes 288 tiles of 384 x 12288: 32 CUs do two), and the tile
//     is as square as that allows (bytes pulled per CU = (BM + BN) * K * 2);
//   * the single workgroup of a CU keeps more than one K tile in flight: three ring stages and a counted vmcnt
//     wait, instead of the 128x128 engine's wait-for-everything before every tile (which relies on a second resident
//     workgroup to cover the gap).
//
// Operand convention, LDS image and swizzle are those of gemm_tile128.h (weights as MFMA operand A, lane-linear DMA
// destination with the XOR applied on the source address).  Four waves as 2 x 2: wave (wm, wn) owns tokens
// wm*16*MT.., features wn*96..: 6 x MT accumulator tiles, kept as a 4 x MT block and a 2 x MT block so the shared
// epilogues (four feature blocks per call) serve the first and a two-block tail serves the second.  With ksplit > 1
// (fp32 partial tiles, split-K path) a workgroup takes one K range of its tile.
#pragma once
#include <type_traits>

#include "coda_common.h"

namespace tile192 {

constexpr int BK = 64;

// TALL = false: waves as 2 x 2, tile (32*MT) x 192.  TALL = true: waves as 4 x 1, tile (64*MT) x 96 (MT = 3 is what is
// instantiated; MT = 4 compiles, but its 96 accumulators + 20 fragment registers spill at 256 VGPRs) - the same wave
// tile, but the workgroup pulls twice the activation rows (L2 hits: every workgroup of the launch reads them) and
// half the weight rows (misses: a weight row is new to the chip when its first reader asks for it, and readers of
// the same row in other CUs wait on that same miss).  A CU's load path sustains 66-73 GB/s on L2 hits but 24-33 GB/s
// on misses (MI355X_MICROARCH.md, 'Indexed rows: gather into LDS'), so its time is W_bytes / r_miss + X_bytes / r_hit
// and the best tile is taller than wide by about r_hit / r_miss.
template <int MT, int STAGES_ = 3, bool TALL = false>
struct Tile {
    static constexpr int BM = (TALL ? 64 : 32) * MT;
    static constexpr int BN = TALL ? 96 : 192;
    static constexpr int XI = BM / 32, WI = BN / 32;           // LDS-DMA instructions per wave, K tile and operand
    static constexpr int X_BYTES = BM * BK * 2;
    static constexpr int W_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = X_BYTES + W_BYTES;      // 36 KiB (MT = 3) / 40 KiB (MT = 4)
    static constexpr int STAGES = STAGES_;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;     // 3 stages: 108 / 120 KiB, 4 stages: 144 / 160 KiB - one workgroup per CU
    static constexpr int DMA_PER_STAGE = XI + WI;

    struct Src {
        const char* x[XI];      // per-lane global source (row, swizzled chunk) of each DMA instruction
        const char* w[WI];
    };

    static __device__ __forceinline__ int dma_row(int i, int wave, int lane) { return (i * 4 + wave) * 8 + (lane >> 3); }
    static __device__ __forceinline__ int dma_chunk(int lane) { return (lane & 7) ^ (lane >> 3); }

    static __device__ __forceinline__ void stage(const Src& f, size_t kbyte, char* lds_stage, int wave) {
#pragma unroll
        for (int i = 0; i < XI; ++i) glds16(f.x[i] + kbyte, lds_stage + (i * 4 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < WI; ++i) glds16(f.w[i] + kbyte, lds_stage + X_BYTES + (i * 4 + wave) * 1024);
    }

    struct Frags { bf16x8_t w[6], x[MT]; };      // one 32-deep K step of the wave tile: 6 weight + MT activation fragments

    // Fragment requests are inline asm on purpose: the compiler's wait-count pass, given its own LDS loads, answers a
    // request that is still wanted LATER with s_waitcnt lgkmcnt(0) in front of the MFMAs that only need the OLDER
    // request (seen with ROCm 7.2: builtin loads, sched_barrier and the waitcnt builtin did not change that).  So the
    // requests are invisible to it and the waits are placed here: landed() names the registers it releases, which
    // keeps every MFMA that reads them behind it.
    template <int OFF>
    static __device__ __forceinline__ bf16x8_t lds16(unsigned addr) {
        bf16x8_t v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
        return v;
    }

    static __device__ __forceinline__ void read(const char* lds_stage, int kk, int wm, int wn, int lane, Frags& g) {
        const int r16 = lane & 15, kq = lane >> 4, sw = lane & 7;
        const int pos = ((kk * 4 + kq) ^ sw) * 16;
        const unsigned base = (unsigned)(size_t)(CODA_LDS const char*)lds_stage;
        const unsigned xs = base + (wm * 16 * MT + r16) * 128 + pos;
        const unsigned ws = base + X_BYTES + (wn * 96 + r16) * 128 + pos;
        g.w[0] = lds16<0 * 2048>(ws); g.w[1] = lds16<1 * 2048>(ws); g.w[2] = lds16<2 * 2048>(ws);
        g.w[3] = lds16<3 * 2048>(ws); g.w[4] = lds16<4 * 2048>(ws); g.w[5] = lds16<5 * 2048>(ws);
        g.x[0] = lds16<0 * 2048>(xs); g.x[1] = lds16<1 * 2048>(xs); g.x[2] = lds16<2 * 2048>(xs);
        if constexpr (MT == 4) g.x[3] = lds16<3 * 2048>(xs);
    }

    // g's request has landed once at most the ONE request issued after it (6 + MT reads, returned in order) is outstanding
    static __device__ __forceinline__ void landed(Frags& g) {
        if constexpr (MT == 3)
            asm volatile("s_waitcnt lgkmcnt(9)"
                         : "+v"(g.w[0]), "+v"(g.w[1]), "+v"(g.w[2]), "+v"(g.w[3]), "+v"(g.w[4]), "+v"(g.w[5]), "+v"(g.x[0]),
                           "+v"(g.x[1]), "+v"(g.x[2]));
        else
            asm volatile("s_waitcnt lgkmcnt(10)"
                         : "+v"(g.w[0]), "+v"(g.w[1]), "+v"(g.w[2]), "+v"(g.w[3]), "+v"(g.w[4]), "+v"(g.w[5]), "+v"(g.x[0]),
                           "+v"(g.x[1]), "+v"(g.x[2]), "+v"(g.x[3]));
    }

    // one LDS-DMA instruction of a stage: pieces 0..XI-1 are activation rows, XI..DMA_PER_STAGE-1 weight rows
    template <int J>
    static __device__ __forceinline__ void piece(const Src& f, size_t kbyte, char* lds_stage, int wave) {
        if constexpr (J < XI) glds16(f.x[J] + kbyte, lds_stage + (J * 4 + wave) * 1024);
        else glds16(f.w[J - XI] + kbyte, lds_stage + X_BYTES + ((J - XI) * 4 + wave) * 1024);
    }

    // The MFMAs of one K step in six groups (one weight fragment each); with DMA, group g is preceded by piece J0 + g
    // of the stage being refilled.  All four waves issuing a whole stage back to back right after the barrier kept
    // every wave in the CU's address path (one 1-KiB piece per ~16 cycles) with its MFMA pipe idle - one wave per
    // SIMD, nobody to cover; spread out, a piece is accepted while the group before it is still in the pipe.
    template <bool DMA, int J0>
    static __device__ __forceinline__ void mma(const Frags& g, f32x4_t (&acc)[4][MT], f32x4_t (&tail)[2][MT], const Src& f,
                                               size_t kbyte, char* lds_stage, int wave) {
        auto group = [&](auto NT) {
            constexpr int nt = decltype(NT)::value;
            if constexpr (DMA && J0 + nt < DMA_PER_STAGE) piece<J0 + nt>(f, kbyte, lds_stage, wave);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (nt < 4) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g.w[nt], g.x[mt], acc[nt][mt], 0, 0, 0);
                else tail[nt - 4][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g.w[nt], g.x[mt], tail[nt - 4][mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{}); group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{}); group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
    }

    // s_waitcnt vmcnt(N tiles' worth of this wave's DMAs) lgkmcnt(0).
    // gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt = imm[6:4] (7 = no wait), lgkmcnt = imm[11:8]
    template <int N>
    static __device__ __forceinline__ void wait_all_but() {
        constexpr int vm = N * DMA_PER_STAGE;
        static_assert(vm < 64, "vmcnt is a 6-bit field");
        __builtin_amdgcn_s_waitcnt(((vm >> 4) << 14) | (vm & 15) | (7 << 4));
    }

    // K loop of one tile, software-pipelined inside the wave: with one wave per SIMD nobody else covers a wave's LDS
    // reads, so the fragments of K step s+1 are requested BEFORE the MFMAs of step s are issued (measured without this:
    // 0.64 us per K tile for 0.24 us of MFMA).  A K tile is two steps; iteration t:
    //     request step (t,1) | MFMAs of (t,0) [+ the last pieces of tile t+2] | wait: own reads done, tile t+1 landed
    //     (tile t+2 may stay in flight) | barrier: tile t+1 is complete and NOBODY reads tile t's buffer any more |
    //     request step (t+1,0) | MFMAs of (t,1) [+ the first six pieces of tile t+3, into tile t's buffer]
    // so three ring stages keep up to three tiles in flight.  A stage's pieces are all issued before the wait that
    // counts them (six behind the barrier of iteration t-1, the rest ahead of the wait of iteration t), so the wait
    // immediates do not depend on where the pieces stand.  (A plain __syncthreads() would drain vmcnt to 0: its fence
    // covers the DMA.)  The first iteration and the last three have no (or only half of the) DMA: separate
    // instances of the body, no branch inside it.
    static __device__ __forceinline__ void mainloop(const Src& f, int K, char* lds, int wave, int lane, f32x4_t (&acc)[4][MT],
                                                    f32x4_t (&tail)[2][MT]) {
        static_assert(STAGES == 3, "the wait immediates below assume three stages");
        const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1;
        const int nk = K / BK;
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nk) stage(f, (size_t)s * BK * 2, lds + s * STAGE_BYTES, wave);
        if (nk >= 3) wait_all_but<2>();
        else if (nk == 2) wait_all_but<1>();
        else wait_all_but<0>();
        __builtin_amdgcn_s_barrier();
        Frags ga, gb;
        read(lds, 0, wm, wn, lane, ga);
        int cur = 0;                              // ring position of tile t
        // DA: the tail pieces of tile t+2 go out under step (t,0); DB: the first six pieces of tile t+3 under step (t,1)
        auto iteration = [&](int t, auto DA_, auto DB_) {
            constexpr bool DA = decltype(DA_)::value, DB = decltype(DB_)::value;
            const int nx = cur == 2 ? 0 : cur + 1, pv = cur == 0 ? 2 : cur - 1;
            read(lds + cur * STAGE_BYTES, 1, wm, wn, lane, gb);
            landed(ga);
            mma<DA, 6>(ga, acc, tail, f, (size_t)(t + 2) * BK * 2, lds + pv * STAGE_BYTES, wave);
            if (t + 2 < nk) wait_all_but<1>();
            else wait_all_but<0>();
            __builtin_amdgcn_s_barrier();
            if (t + 1 < nk) read(lds + nx * STAGE_BYTES, 0, wm, wn, lane, ga);   // (never leave a request in flight at the
            landed(gb);                                                          //  end: its registers get reused)
            mma<DB, 0>(gb, acc, tail, f, (size_t)(t + 3) * BK * 2, lds + cur * STAGE_BYTES, wave);
            cur = nx;
        };
        using Y = std::true_type;
        using N = std::false_type;
        if (nk >= 4) {
            iteration(0, N{}, Y{});
            for (int t = 1; t + 3 < nk; ++t) iteration(t, Y{}, Y{});
            iteration(nk - 3, Y{}, N{});
            iteration(nk - 2, N{}, N{});
            iteration(nk - 1, N{}, N{});
        } else {
            for (int t = 0; t < nk; ++t) iteration(t, N{}, N{});
        }
    }
};

}  // namespace tile192
