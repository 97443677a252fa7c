// Shared device helpers for the CoDA MI355X (gfx950 / CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define CODA_LDS __attribute__((address_space(3)))
#define CODA_GLOBAL __attribute__((address_space(1)))

#include "coda_hip.h"   // public C ABI (include/): error codes, enums, prototypes

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

// round-to-nearest-even f32 -> bf16 bits (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950, NaN-safe)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}

// value of f after a round trip through bf16 (emulates an op whose output dtype is bf16)
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// two f32 -> one dword of two bf16 (RNE): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    const bf16x2_t h = __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t);
    return __builtin_bit_cast(unsigned int, h);
}

__device__ __forceinline__ float bf16_lo(unsigned int pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int pk) { return __uint_as_float(pk & 0xffff0000u); }

// round_bf16 of two values for the price of three instructions
__device__ __forceinline__ void round_bf16_pair(float& a, float& b) {
    const unsigned int pk = pack_bf16x2(a, b);
    a = bf16_lo(pk);
    b = bf16_hi(pk);
}

// 8 bf16 (one 16-byte piece) -> 8 floats
__device__ __forceinline__ void unpack8(const u32x4_t v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}

// Counter-based uniform in (0, 1): a pure function of (seed, row key, index).  All sampling noise that is not injected
// by the caller comes from here (token noise: row key = flat position of the candidate, index = vocabulary id;
// selection / origin noise: index = 0xFFFFFFFF / 0xFFFFFFFE), so a run is reproducible from its seed alone and no
// RNG state or noise tensor lives in memory.
__device__ __forceinline__ float coda_hash_uniform(unsigned int seed_lo, unsigned int seed_hi, unsigned int rowkey, unsigned int v) {
    unsigned int h = v * 0x9E3779B1u ^ rowkey * 0x85EBCA77u ^ seed_lo;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += seed_hi;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    // 23 bits + 0.5: every value is exact in fp32 and u lies in [2^-24, 1 - 2^-24].  (24 bits + 0.5 needs 25 significant
    // bits: the top value rounds to 1.0, -log(u) = 0, and the race key of that entry becomes +inf - it wins whatever its logit.)
    return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// x / t for a loop-invariant divisor, correctly rounded like the IEEE quotient (what `logits / temperature` is in the
// reference), in three operations instead of the ~12 of a full division: q0 = RN(x rt), r = x - q0 t (exact in one FMA),
// q1 = RN(q0 + r rt) with rt = RN(1 / t).  Markstein's sequence gives the correctly rounded quotient for MOST (x, t), not
// all (rt is rounded twice, from double; exceptional pairs exist), so the fast form is used only where it has been
// VERIFIED: `rt` comes from coda_reciprocal_if_exact(t) (abi_misc.hip), which runs the sequence on the host for every
// bf16 x in the fast-path domain below against the IEEE quotient and returns 0 if a single one differs - and 0 for fp32
// logits, whose x are not enumerable.  rt == 0, or an x outside the domain (quotient or intermediate could leave the
// normal range; +-inf from a logits hook that bans tokens), takes the true division.
// fast-path domain of coda_div_by: +-0, or a biased exponent in [28, 226), i.e. 2^-99 <= |x| < 2^99 (quotient and
// intermediates stay normal for every licensed t in [1e-3, 1e3]); the host check enumerates exactly this set
__host__ __device__ __forceinline__ bool coda_div_fast_domain(unsigned int bits) {
    const unsigned int e = (bits >> 23) & 0xffu;
    return (e - 28u) < 198u || (bits << 1) == 0u;
}

__device__ __forceinline__ float coda_div_by(float x, float t, float rt) {
    const float q0 = x * rt;
    const float r = __builtin_fmaf(-q0, t, x);
    float q = __builtin_fmaf(r, rt, q0);
    // rare: rt == 0 is uniform over the launch (temperature not licensed / fp32 logits); the domain test fails only for
    // values no logit has (|x| >= 2^99, denormal-range quotients, +-inf from a hook that bans tokens)
    if (__builtin_expect(rt == 0.f || !coda_div_fast_domain(__float_as_uint(x)), 0)) q = __fdiv_rn(x, t);
    return q;
}

// Exponential race with coda_hash_uniform noise: key = logit - log(-log(u)), u in [2^-24, 1 - 2^-24], so the noise term
// lies in [-2.82, 16.64]; an entry more than this far below a known lower bound of the winner's logit cannot win.
#define CODA_RACE_WINDOW 20.5f

// 16-byte asynchronous global -> LDS copy.  The LDS destination is wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const CODA_GLOBAL void*)gsrc, (CODA_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups that share an XCD (id % 8 equal
// under round-robin dispatch) receive one contiguous chunk of the logical tile order.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
