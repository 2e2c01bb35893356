// Device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "fzb_internal.h"

#define FZB_WAVE 64

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename ET>
__device__ __forceinline__ void haystack_span(const ET* __restrict__ ends, u64 i, u64& start, u32& len) {
    // padded-16 layout: start(i) = i ? roundup16(ends[i-1]) : 0
    u64 e = (u64)ends[i];
    u64 ep = i ? (u64)ends[i - 1] : 0;
    start = (ep + 15) & ~(u64)15;
    len = (u32)(e - start);
}

// Read the 32-bit word holding bytes [p, p+4) of `base` for an arbitrary byte offset p
// (two aligned loads + v_alignbyte; the corpus has >= 80 readable bytes past the end).
__device__ __forceinline__ u32 load_u32_unaligned(const u8* __restrict__ base, u64 p) {
    const u32* a = (const u32*)(base + (p & ~(u64)3));
    u32 lo = a[0], hi = a[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (u32)(p & 3));
}

// Queue slots for the lanes of a wave that want one: ONE atomic per wave instead of one per lane.  Every currently active
// lane must call it (with its own flag); the returned slot is meaningful where `want` is true.
__device__ __forceinline__ u32 wave_alloc(u32* counter, bool want) {
    const u64 mask = __ballot(want);
    u32 base = 0;
    if (mask) {
        const int leader = __builtin_ctzll(mask);
        if (lane_id() == leader) base = atomicAdd(counter, (u32)__popcll(mask));
        base = __shfl(base, leader);
    }
    return base + (u32)__popcll(mask & (((u64)1 << lane_id()) - 1));
}
