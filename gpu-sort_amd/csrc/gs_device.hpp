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

// Lanes of this wave whose 8-bit digit equals mine (all 64 lanes must be
// active).  One ballot per digit bit; the per-lane select keeps the lanes that
// agree with my bit.
__device__ __forceinline__ uint64_t match_digit(uint32_t d)
{
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// popcount of `mask` restricted to lanes below mine
__device__ __forceinline__ uint32_t count_lower(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive wave scan (64 lanes)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
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
