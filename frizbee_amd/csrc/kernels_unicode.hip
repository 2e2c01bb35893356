// gfx950 kernels, stage 2b (unicode): thread-per-haystack single-chunk unicode Smith-Waterman (dp_unicode.h) over the
// survivors of the lane-exact unicode prefilter.  Windows wider than one chunk (or > 1024 bytes) are queued for the
// generic wave-per-haystack kernel (kernels_generic.hip), exactly like the ASCII single-chunk kernel does.
#include "dp_unicode.h"

template <int SWL, bool HALFONLY, typename ET>
__device__ __forceinline__ void k2u_body(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 index_offset,
                                                      const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                                      const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity,
                                                      u32* __restrict__ dev_count, u32* __restrict__ overflow, u32 qcap,
                                                      u32* __restrict__ counters, u32 ulen) {
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 M = *n_items_ptr;
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = M < capacity ? M : capacity; dev_count[1] = M; }
    // Persistent threads with the same three-deep software pipeline as k2b_dp over the dependent loads (item / window -> end
    // offsets -> haystack bytes).  This kernel needs nearly the whole register file (one wave per SIMD), so no second wave covers
    // a stall: the stages are requested one iteration ahead, and the haystack's first line is touched one iteration ahead so that
    // the scorer's own byte loads find it in L2.
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 j0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    auto load_item = [&](u64 j, u32& li, u32& ws, u32& we) {
        li = 0; ws = 0; we = 0;
        if (j < M) {
            li = items ? items[j] : (u32)j;
            if (wmode == 0) { const uint2 w = *(const uint2*)(win + 2 * j); ws = w.x; we = w.y; }
        }
    };
    auto load_span = [&](u64 j, u32 li, u64& s, u32& L) {
        s = 0; L = 0;
        if (j < M) haystack_span_u(ends, ulen, first + li, s, L);
    };
    u32 li_c, ws_c, we_c, L_c, li_n, ws_n, we_n, L_n, li_m, ws_m, we_m;
    u64 s_c, s_n;
    load_item(j0, li_c, ws_c, we_c);
    load_item(j0 + stride, li_n, ws_n, we_n);
    load_item(j0 + 2 * stride, li_m, ws_m, we_m);
    load_span(j0, li_c, s_c, L_c);
    load_span(j0 + stride, li_n, s_n, L_n);
    u32 warm = 0;
    for (u64 j = j0; j < M; j += stride) {
        u32 warm_n = 0;
        if (L_n > 0) warm_n = *(const u32*)(bytes + s_n + (ws_n & ~3u));  // the next item's window start: brings its line(s) into L2
        u64 s_m;
        u32 L_m;
        load_span(j + 2 * stride, li_m, s_m, L_m);
        u32 li_f, ws_f, we_f;
        load_item(j + 3 * stride, li_f, ws_f, we_f);
        do {
            if (j >= capacity) break;
            const u32 li = li_c, L = L_c;
            const u8* hay = bytes + s_c;
            u32 ws = ws_c, we = we_c;
            if (wmode == 2) { ws = 0; we = L; }
            else if (wmode == 1) unicode_window_first_last(nd, hay, L, ws, we);
            const u32 sp = ws ? ws - 1 : 0;
            const bool include_exact = sp == 0 && we == L;
            const u32 m = we - sp;
            if (m > (u32)SWL) {
                const u32 slot = atomicAdd(&counters[4], 1u);
                u32* qe = overflow + 4 * (size_t)(qcap - 1 - slot);  // back of the queue slice: consumed by the generic kernel
                qe[0] = (u32)j;
                qe[1] = ws;
                qe[2] = we;
                qe[3] = li;
                break;
            }
            u32 score = 0;
            if (m > 0 && nd.rows > 0) {
                // wave-uniform choice: if every window of the wave fits the low half of the chunk, the upper half is pure padding
                const bool half = HALFONLY || (SWL >= 16 && __all((int)(m <= (u32)SWL / 2)));
                if (half) score = dp_unicode_single_chunk<SWL, (SWL >= 16 ? SWL / 4 : SWL / 2)>(nd, hay + sp, m, sp == 0, cls);
                else if (!HALFONLY) score = dp_unicode_single_chunk<SWL>(nd, hay + sp, m, sp == 0, cls);
            }
            bool exact = include_exact && m == (u32)nd.nbytes;
            if (exact)
                for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li;
            rec.score = (u16)(score + (warm & 0u));  // keeps the warm-up load of the previous iteration alive until here
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            out[j] = rec;
        } while (0);
        li_c = li_n; ws_c = ws_n; we_c = we_n; s_c = s_n; L_c = L_n;
        li_n = li_m; ws_n = ws_m; we_n = we_m; s_n = s_m; L_n = L_m;
        li_m = li_f; ws_m = ws_f; we_m = we_f;
        warm = warm_n;
    }
}

#define FZB_K2U_PARAMS const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 index_offset, const u32* __restrict__ items, const u32* __restrict__ win, \
    const u32* __restrict__ n_items_ptr, const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ overflow, \
    u32 qcap, u32* __restrict__ counters, u32 ulen
#define FZB_K2U_ARGS bytes, ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, qcap, counters, ulen
template <int SWL, typename ET>
__global__ __launch_bounds__(128) void k2u_dp_unicode(FZB_K2U_PARAMS) { k2u_body<SWL, false, ET>(FZB_K2U_ARGS); }
// every haystack of the list fits the low half of a chunk (host-known: corpus max_len <= SWL / 2): the general form is compiled out,
// which lets the kernel fit 168 VGPRs (a few spills outside the row loop) and run three waves per SIMD
template <int SWL, typename ET>
__global__ __launch_bounds__(128, 3) void k2u_dp_unicode_half(FZB_K2U_PARAMS) { k2u_body<SWL, true, ET>(FZB_K2U_ARGS); }

void fzb_launch_dp_unicode(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                           int sw_lanes, int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters,
                           int grid, hipStream_t st) {
    // `grid` = number of CUs: the kernel is persistent, launch exactly the resident workgroups
    const bool half_only = sw_lanes >= 16 && c.max_len != 0 && c.max_len <= (u32)sw_lanes / 2;
#define FZB_K2U(SWL, ET)                                                                                                               \
    do {                                                                                                                               \
        static int per_cu = 0, per_cu_half = 0;                                                                                        \
        if (half_only) {                                                                                                               \
            if (!per_cu_half && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_half, k2u_dp_unicode_half<SWL, ET>, 128, 0) != hipSuccess || per_cu_half < 1)) per_cu_half = 4; \
            hipLaunchKernelGGL((k2u_dp_unicode_half<SWL, ET>), dim3(grid * per_cu_half), dim3(128), 0, st, c.bytes, (const ET*)c.ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, qcap, counters, c.uniform_len); \
        } else {                                                                                                                       \
            if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2u_dp_unicode<SWL, ET>, 128, 0) != hipSuccess || per_cu < 1)) per_cu = 2; \
            hipLaunchKernelGGL((k2u_dp_unicode<SWL, ET>), dim3(grid * per_cu), dim3(128), 0, st, c.bytes, (const ET*)c.ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, qcap, counters, c.uniform_len); \
        }                                                                                                                              \
    } while (0)
#define FZB_K2U_ET(SWL) do { if (c.ends_u64) FZB_K2U(SWL, u64); else FZB_K2U(SWL, u32); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2U_ET(64); break;
        case 32: FZB_K2U_ET(32); break;
        case 16: FZB_K2U_ET(16); break;
        default: FZB_K2U_ET(8); break;
    }
}
