// Device helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "fzb_internal.h"
#include "knobs.h"

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

// End offsets of either width behind ONE kernel instantiation: a wave-uniform branch per read instead of a template parameter that doubled every
// scorer (round 6: the u32 / u64 twins were 2.6 MB of the library).  The streaming filters, which read one offset per haystack of the whole
// list, keep the template.
struct EndsAny {
    const void* p;
    int is64;
    __device__ __forceinline__ u64 operator[](u64 i) const { return is64 ? ((const u64*)p)[i] : (u64)((const u32*)p)[i]; }
};
__device__ __forceinline__ void haystack_span(const EndsAny& ends, u64 i, u64& start, u32& len) {
    u64 e = ends[i];
    u64 ep = i ? ends[i - 1] : 0;
    start = (ep + 15) & ~(u64)15;
    len = (u32)(e - start);
}
__device__ __forceinline__ void haystack_span_u(const EndsAny& ends, u32 ulen, u64 i, u64& start, u32& len) {
    if (ulen) {
        start = i * (u64)((ulen + 15u) & ~15u);
        len = ulen;
    } else {
        haystack_span(ends, i, start, len);
    }
}

// the same when every haystack of the corpus has `ulen` bytes (CorpusDev::uniform_len): no loads, and the haystack's vectors can be
// requested without waiting for an end offset
template <typename ET>
__device__ __forceinline__ void haystack_span_u(const ET* __restrict__ ends, u32 ulen, u64 i, u64& start, u32& len) {
    if (ulen) {
        start = i * (u64)((ulen + 15u) & ~15u);
        len = ulen;
    } else {
        haystack_span(ends, i, start, len);
    }
}

// Read the 32-bit word holding bytes [p, p+4) of `base` for an arbitrary byte offset p
// (two aligned loads + v_alignbyte; the corpus has >= 80 readable bytes past the end).
__device__ __forceinline__ u32 load_u32_unaligned(const u8* __restrict__ base, u64 p) {
    const u32* a = (const u32*)(base + (p & ~(u64)3));
    u32 lo = a[0], hi = a[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (u32)(p & 3));
}

// 4 / 8 / 12 bytes of a once-read stream (the narrow tail rows of the filter view: lane * width, so 8 bytes are 8-aligned and 12 bytes only
// 4-aligned), with or without the non-temporal hint; the upper dwords of the result are zero
__device__ __forceinline__ uint4 load_narrow_stream(const u8* p, u32 width, bool nt) {
    typedef u32 v2u __attribute__((ext_vector_type(2)));
    typedef u32 v3u4 __attribute__((ext_vector_type(3), aligned(4)));
    uint4 r = make_uint4(0, 0, 0, 0);
#ifndef FZB_HOST_SHIM
    if (nt) {
        if (width == 12) { const v3u4 t = __builtin_nontemporal_load((const v3u4*)p); r.x = t.x; r.y = t.y; r.z = t.z; }
        else if (width == 8) { const v2u t = __builtin_nontemporal_load((const v2u*)p); r.x = t.x; r.y = t.y; }
        else r.x = __builtin_nontemporal_load((const u32*)p);
        return r;
    }
#endif
    if (width == 12) { const v3u4 t = *(const v3u4*)p; r.x = t.x; r.y = t.y; r.z = t.z; }
    else if (width == 8) { const v2u t = *(const v2u*)p; r.x = t.x; r.y = t.y; }
    else r.x = *(const u32*)p;
    return r;
}

// A 16-byte load of data that is read ONCE by a streaming kernel: the non-temporal hint keeps the stream from displacing what later
// stages re-read from the caches (measured on the view filter: 190 -> 183 us on the C4 shard).
template <bool NT>
__device__ __forceinline__ uint4 load16_stream(const uint4* p) {
#ifndef FZB_HOST_SHIM
    if (NT) {
        typedef u32 v4u __attribute__((ext_vector_type(4)));
        const v4u t = __builtin_nontemporal_load((const v4u*)p);
        return make_uint4(t.x, t.y, t.z, t.w);
    }
#endif
    return *p;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() = fence + s_barrier, and the fence waits for every outstanding memory
// operation of the wave (s_waitcnt vmcnt(0)), global STORES included; a kernel that stores to global memory for LATER kernels only and
// synchronises its waves over LDS state need not expose that round trip.
#ifdef FZB_HOST_SHIM
__device__ __forceinline__ void barrier_lds_only() {}
#else
__device__ __forceinline__ void barrier_lds_only() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

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

// An optimisation barrier on a vector value: what is derived from x after this point is computed after this point (keeps the compiler
// from hoisting loop-invariant values out of a loop at the price of registers, see dp_unicode.h).  No instruction is emitted.
#ifdef FZB_HOST_SHIM
#define FZB_OPAQUE_V(x) ((void)0)
#else
#define FZB_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

// A scheduling fence: the compiler does not move instructions across it.  Used between the unrolled iterations of the register-heavy
// scorers so that the temporaries of sixteen iterations are not all live at once (dp_unicode.h: spills inside the row loop otherwise).
#ifdef FZB_HOST_SHIM
#define FZB_SCHED_FENCE() ((void)0)
#else
#define FZB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// a + b on the scalar unit, opaque to the optimiser: a chain `x = fzb_sadd(x, step)` over an unrolled loop stays one s_add per
// link (the compiler otherwise rewrites it into a multiply and an add per element)
__device__ __forceinline__ u32 fzb_sadd(u32 a, u32 b) {
#ifdef FZB_HOST_SHIM  // tests/kernel_host: the same arithmetic compiled for the host
    return a + b;
#else
    u32 r;  // operands must be wave-uniform; readfirstlane is free when the compiler already holds them in scalar registers
    asm volatile("s_add_u32 %0, %1, %2" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane(a)), "s"(__builtin_amdgcn_readfirstlane(b)) : "scc");
    return r;
#endif
}

// Issue priority of the calling wave (s_setprio takes an immediate, hence the switch); p must be wave-uniform.
// The SIMD's arbiter serves its waves by priority, then age: at equal priority the oldest wave runs nearly unimpeded and the youngest
// gets the leftover issue slots, so equal shares of work finish one after the other and the last wave runs alone - at about half the
// VALU rate, because a single wave cannot hide its own scalar instructions, waits and dependencies.  The persistent scorers therefore
// lower a wave's priority as it progresses through its share (progress in quarters), which keeps the waves of a SIMD within a quarter of
// each other until the end (MI355X_MICROARCH.md, "VALU issue is arbitrated ... by priority, then age").
__device__ __forceinline__ void fzb_set_wave_priority(u32 p) {
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
// progress of a wave through its share of a persistent kernel, in quarters: called once per item with the item's number and the
// wave's item count (both wave-uniform SCALARS - pass them through __builtin_amdgcn_readfirstlane), and once more half way
// through the item
struct FzbProgressPrio {
    u32 k2, n2;  // 2 * item number (+ 1 in the second half), 2 * items of this wave
    __device__ __forceinline__ void apply() const {
        const u32 q = 4 * k2;  // level = floor(4 * k2 / n2), without a division
        fzb_set_wave_priority(q < n2 ? 3u : q < 2 * n2 ? 2u : q < 3 * n2 ? 1u : 0u);
    }
};
