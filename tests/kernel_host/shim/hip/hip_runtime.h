// TEST INFRASTRUCTURE.  A stand-in for <hip/hip_runtime.h> that lets the *device* arithmetic of frizbee_amd/csrc/dp_*.h be compiled
// for the host (clang++ -x c++) so that tests can fuzz it against the oracle without a GPU.  Nothing in the product path includes this.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>

#define __device__
#define __host__
#define __global__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct dim3 { unsigned x = 1, y = 1, z = 1; };
static thread_local dim3 threadIdx{0, 0, 0};  // the harness steps it where a kernel fills a table cooperatively (thread-local: the four-lane harness below runs a host thread per lane)
static const dim3 blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
typedef void* hipStream_t;
typedef void* hipEvent_t;

// one-thread "wave": the cross-lane helpers degenerate
static inline uint64_t __ballot(int p) { return p ? 1ull : 0ull; }
static inline int __all(int p) { return p != 0; }
static inline int __any(int p) { return p != 0; }
template <typename T> static inline T __shfl(T v, int) { return v; }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
static inline void __syncthreads() {}
static inline void __threadfence() {}
using std::max;
using std::min;

// v_alignbit_b32 / v_alignbyte_b32 / v_perm_b32 (ISA semantics)
static inline uint32_t fzb_host_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31)); }
static inline uint32_t fzb_host_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3))); }
static inline uint32_t fzb_host_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t src = (((uint64_t)s0) << 32) | s1;
    uint32_t out = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFF;
        uint32_t b;
        if (s <= 7) b = (uint32_t)(src >> (8 * s)) & 0xFF;
        else if (s == 8) b = (s1 >> 15) & 1 ? 0xFF : 0;
        else if (s == 9) b = (s1 >> 31) & 1 ? 0xFF : 0;
        else if (s == 10) b = (s0 >> 15) & 1 ? 0xFF : 0;
        else if (s == 11) b = (s0 >> 31) & 1 ? 0xFF : 0;
        else if (s == 12) b = 0;
        else b = 0xFF;
        out |= b << (8 * i);
    }
    return out;
}
#define __builtin_amdgcn_alignbit fzb_host_alignbit
#define __builtin_amdgcn_alignbyte fzb_host_alignbyte
#define __builtin_amdgcn_perm fzb_host_perm
static inline void __builtin_amdgcn_s_setprio_host(int) {}
#define __builtin_amdgcn_s_setprio __builtin_amdgcn_s_setprio_host
#define FZB_HOST_SHIM 1
static inline uint32_t fzb_host_rfl(uint32_t v) { return v; }
#define __builtin_amdgcn_readfirstlane fzb_host_rfl

// ---- four lanes in lockstep for dp_quad.h (tests/kernel_host/dp_host.cpp: a host thread per quad lane) ------------------------------------
// __builtin_amdgcn_update_dpp with row_shr:K / row_ror:K, K = 4 or 8, on dp_quad.h's layout (quad lane L = row-lanes w + 4 L): every lane
// reaches the same call in the same order (the control flow of a window is the same in its four lanes), so a call is: publish my source,
// wait for the other three, read the source lane's (out of the row: keep `old`; rotate: wrap), wait again before the slots are reused.
#include <atomic>
struct FzbQuadBus {
    std::atomic<int> arrived{0}, gen{0};
    uint32_t slot[4];
};
static FzbQuadBus fzb_quad_bus;
static thread_local int fzb_quad_lane = 0;
static inline void fzb_quad_barrier() {
    const int g = fzb_quad_bus.gen.load(std::memory_order_acquire);
    if (fzb_quad_bus.arrived.fetch_add(1, std::memory_order_acq_rel) == 3) {
        fzb_quad_bus.arrived.store(0, std::memory_order_relaxed);
        fzb_quad_bus.gen.fetch_add(1, std::memory_order_release);
    } else {
        while (fzb_quad_bus.gen.load(std::memory_order_acquire) == g) {}
    }
}
static inline int fzb_host_update_dpp(int old, int src, int ctrl, int, int, bool) {
    fzb_quad_bus.slot[fzb_quad_lane] = (uint32_t)src;
    fzb_quad_barrier();
    const bool ror = (ctrl & 0x1F0) == 0x120;
    int from = fzb_quad_lane - (ctrl & 0xF) / 4;
    int v = old;
    if (from >= 0) v = (int)fzb_quad_bus.slot[from];
    else if (ror) v = (int)fzb_quad_bus.slot[from + 4];
    fzb_quad_barrier();
    return v;
}
#define __builtin_amdgcn_update_dpp fzb_host_update_dpp
