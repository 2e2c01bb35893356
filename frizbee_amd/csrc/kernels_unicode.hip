// gfx950 kernels, stage 2b (unicode): thread-per-haystack single-chunk unicode Smith-Waterman (dp_unicode.h) over the
// survivors of the lane-exact unicode prefilter.  Windows wider than one chunk (or > 1024 bytes) are queued for the
// generic wave-per-haystack kernel (kernels_generic.hip), exactly like the ASCII single-chunk kernel does.
#include "dp_unicode.h"

template <int SWL, typename ET>
__global__ __launch_bounds__(128) void k2u_dp_unicode(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 index_offset,
                                                      const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                                      const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity,
                                                      const u32* __restrict__ base_ptr, u32* __restrict__ dev_count, u32* __restrict__ overflow, u32 qcap,
                                                      u32* __restrict__ counters) {
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 M = *n_items_ptr;
    const u32 base = base_ptr ? *base_ptr : 0u;
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) *dev_count = (base + M) < capacity ? (base + M) : capacity;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        if (base + j >= capacity) continue;
        const u32 li = items ? items[j] : j;
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        const u8* hay = bytes + s;
        u32 ws, we;
        if (wmode == 2) { ws = 0; we = L; } else { ws = win[2 * j]; we = win[2 * j + 1]; }
        const u32 sp = ws ? ws - 1 : 0;
        const bool include_exact = sp == 0 && we == L;
        const u32 m = we - sp;
        if (m > (u32)SWL) {
            const u32 slot = atomicAdd(&counters[4], 1u);
            u32* qe = overflow + 4 * (size_t)(qcap - 1 - slot);  // back of the queue slice: consumed by the generic kernel
            qe[0] = base + j;
            qe[1] = ws;
            qe[2] = we;
            qe[3] = li;
            continue;
        }
        u32 score = 0;
        if (m > 0 && nd.rows > 0) score = dp_unicode_single_chunk<SWL>(nd, hay + sp, m, sp == 0, cls);
        bool exact = include_exact && m == (u32)nd.nbytes;
        if (exact)
            for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
        if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = exact ? 1 : 0;
        rec.valid = 0;
        out[base + j] = rec;
    }
}

void fzb_launch_dp_unicode(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                           int sw_lanes, int wmode, fzb_match_rec* out, u32 capacity, const u32* base_ptr, u32* dev_count, u32* overflow, u32 qcap, u32* counters,
                           int grid, hipStream_t st) {
#define FZB_K2U(SWL, ET) hipLaunchKernelGGL((k2u_dp_unicode<SWL, ET>), dim3(grid), dim3(128), 0, st, c.bytes, (const ET*)c.ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, base_ptr, dev_count, overflow, qcap, counters)
#define FZB_K2U_ET(SWL) do { if (c.ends_u64) FZB_K2U(SWL, u64); else FZB_K2U(SWL, u32); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2U_ET(64); break;
        case 32: FZB_K2U_ET(32); break;
        case 16: FZB_K2U_ET(16); break;
        default: FZB_K2U_ET(8); break;
    }
}
