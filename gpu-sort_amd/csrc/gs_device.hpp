// gs_device.hpp -- wave64 device helpers shared by the gfx950 kernels.
// Written for CDNA4 only: 64-lane wavefronts, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gs {

constexpr int WAVE = 64;
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// Order-preserving key -> u32 map and its inverse.  `f32` selects the
// data-dependent float twiddle; `x` is a wave-uniform xor mask that carries the
// i32 sign flip and the descending complement.  Reference semantics:
// cub::Traits<K>::TwiddleIn/Out, lsb/cub/cub/util_type.cuh:966-974,1009-1017,1079-1089.
__device__ __forceinline__ uint32_t twiddle_in(uint32_t k, int f32, uint32_t x)
{
    if (f32) k ^= (uint32_t)((int32_t)k >> 31) | 0x80000000u;
    return k ^ x;
}
__device__ __forceinline__ uint32_t twiddle_out(uint32_t k, int f32, uint32_t x)
{
    k ^= x;
    if (f32) k ^= ~(uint32_t)((int32_t)k >> 31) | 0x80000000u;
    return k;
}

__device__ __forceinline__ uint32_t count_lower_mask(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Lanes of this wave whose 8-bit digit equals mine (all 64 lanes must be
// active), as two 32-bit mask halves.  Per digit bit: v_bfe_i32 replicates my
// bit to 32 bits (s), v_cmp ballots it into an SGPR pair (m), and one
// v_bitop3_b32 per half keeps the lanes that agree with me:
// p & ~(m ^ s), truth table 0x90 for (p, m, s).  32 VALU in all.  The stream is
// software-pipelined over two SGPR pairs so every ballot is read at least two
// instructions after the v_cmp that wrote it (VALU-write-SGPR -> VALU-read
// hazard on gfx950; nothing pads hazards inside an asm statement).
__device__ __forceinline__ void match_digit(uint32_t d, uint32_t &lo, uint32_t &hi)
{
    uint32_t t0, t1, t2;
    asm volatile(
        "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
        "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t1]\n\t"
        "v_bfe_i32 %[t2], %[d], 2, 1\n\t"
        "v_xnor_b32 %[lo], s68, %[t0]\n\t"
        "v_xnor_b32 %[hi], s69, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t2]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t1] bitop3:0x90\n\t"
        "v_bfe_i32 %[t0], %[d], 3, 1\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t0]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t2] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t2] bitop3:0x90\n\t"
        "v_bfe_i32 %[t1], %[d], 4, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t1]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t0] bitop3:0x90\n\t"
        "v_bfe_i32 %[t2], %[d], 5, 1\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t2]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t1] bitop3:0x90\n\t"
        "v_bfe_i32 %[t0], %[d], 6, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t0]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t2] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t2] bitop3:0x90\n\t"
        "v_bfe_i32 %[t1], %[d], 7, 1\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t1]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t1] bitop3:0x90"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [d] "v"(d)
        : "s68", "s69", "s70", "s71");
}

// the same for digits below 32 (5 ballots, 20 instructions): local-sort passes on few bits
__device__ __forceinline__ void match_digit5(uint32_t d, uint32_t &lo, uint32_t &hi)
{
    uint32_t t0, t1, t2;
    asm volatile(
        "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
        "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t1]\n\t"
        "v_bfe_i32 %[t2], %[d], 2, 1\n\t"
        "v_xnor_b32 %[lo], s68, %[t0]\n\t"
        "v_xnor_b32 %[hi], s69, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t2]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t1] bitop3:0x90\n\t"
        "v_bfe_i32 %[t0], %[d], 3, 1\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t0]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t2] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t2] bitop3:0x90\n\t"
        "v_bfe_i32 %[t1], %[d], 4, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t1]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t1] bitop3:0x90"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [d] "v"(d)
        : "s68", "s69", "s70", "s71");
}

// digits below 16 (4 ballots, 16 instructions) and below 4 (2 ballots, 8 instructions)
__device__ __forceinline__ void match_digit4(uint32_t d, uint32_t &lo, uint32_t &hi)
{
    uint32_t t0, t1, t2;
    asm volatile(
        "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
        "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t1]\n\t"
        "v_bfe_i32 %[t2], %[d], 2, 1\n\t"
        "v_xnor_b32 %[lo], s68, %[t0]\n\t"
        "v_xnor_b32 %[hi], s69, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t2]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t1] bitop3:0x90\n\t"
        "v_bfe_i32 %[t0], %[d], 3, 1\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t0]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s68, %[t2] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s69, %[t2] bitop3:0x90\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t0] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t0] bitop3:0x90"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [d] "v"(d)
        : "s68", "s69", "s70", "s71");
}
__device__ __forceinline__ void match_digit2(uint32_t d, uint32_t &lo, uint32_t &hi)
{
    uint32_t t0, t1;
    asm volatile(
        "v_bfe_i32 %[t0], %[d], 0, 1\n\t"
        "v_bfe_i32 %[t1], %[d], 1, 1\n\t"
        "v_cmp_ne_u32_e64 s[68:69], 0, %[t0]\n\t"
        "v_cmp_ne_u32_e64 s[70:71], 0, %[t1]\n\t"
        "s_nop 0\n\t"
        "v_xnor_b32 %[lo], s68, %[t0]\n\t"
        "v_xnor_b32 %[hi], s69, %[t0]\n\t"
        "v_bitop3_b32 %[lo], %[lo], s70, %[t1] bitop3:0x90\n\t"
        "v_bitop3_b32 %[hi], %[hi], s71, %[t1] bitop3:0x90"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [d] "v"(d)
        : "s68", "s69", "s70", "s71");
}

// rank of my lane among the set lanes of (hi:lo)
__device__ __forceinline__ uint32_t count_lower(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
}

// histogram increment that stays cheap under skew: when every active lane of the wave
// holds the same digit (a hot bucket) one lane adds the lane count, otherwise each lane
// adds 1 (LDS atomic; same-address lanes serialise, so the uniform case must not collide)
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t d)
{
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
    if (__builtin_amdgcn_ballot_w64(d == d0) == act) {
        if (count_lower_mask(act) == 0) atomicAdd(&hist[d0], (uint32_t)__popcll(act));
    } else {
        atomicAdd(&hist[d], 1u);
    }
}

// popcount of `mask` restricted to lanes below mine
__device__ __forceinline__ uint32_t count_lower(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive wave scan (64 lanes) with DPP row shifts / row broadcasts: seven
// VALU ops, no LDS traffic.  update_dpp(old=0, ...) yields 0 for lanes whose
// source lane is outside the row or that are masked off by row/bank masks.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);  // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);  // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);  // row_shr:8, banks 2-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Exclusive scan of one value per thread over the first 256 threads (4 waves)
// of the block.  `scratch` is 8 dwords of LDS.  Every thread of the block must
// call it (it contains block barriers); threads >= 256 pass v = 0 and get junk.
// Returns the exclusive prefix; *total receives the sum of all 256 values.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *scratch, uint32_t *total)
{
    const int w = wave_id();
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == 63 && w < 4) scratch[w] = inc;
    __syncthreads();
    const uint32_t s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
    uint32_t base = 0;
    if (w == 1) base = s0;
    else if (w == 2) base = s0 + s1;
    else if (w == 3) base = s0 + s1 + s2;
    if (total) *total = s0 + s1 + s2 + s3;
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace gs
