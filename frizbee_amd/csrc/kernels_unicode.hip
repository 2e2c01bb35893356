// gfx950 kernels, stage 2b (unicode): thread-per-haystack unicode Smith-Waterman (dp_unicode.h) over the survivors of the unicode
// prefilter.  Windows wider than one chunk are queued: up to 1024 bytes from the FRONT of the queue for k2u_dp_unicode_multi (thread per
// haystack, chunk by chunk; round 4), beyond that from the back for the generic wave-per-haystack kernel (kernels_generic.hip: the greedy
// fallback), exactly like the ASCII kernels do.
#include "dp_unicode.h"
#include <cstdlib>

template <int SWL, bool HALFONLY, bool TF>
__device__ __forceinline__ void k2u_body(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                                      const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                                      const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity,
                                                      u32* __restrict__ dev_count, u32* __restrict__ overflow, u32 qcap,
                                                      u32* __restrict__ counters, u32 ulen, u32 multi_front) {
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
    // REGS: every haystack of the list fits two 16-byte vectors (HALFONLY at 64 lanes and below) and the biased-throughout form runs: the
    // vectors are requested one iteration ahead, the 0-typo window is found in them (unicode_window_regs) and the window's bytes are shifted
    // out of them (load_window_regs) - no byte-wise scan with a dependent load per dword, no second read of the window
    constexpr bool REGS = HALFONLY && TF && SWL <= 64;
    u32 warm = 0;
    uint4 q0_c = make_uint4(0, 0, 0, 0), q1_c = q0_c;
    if (REGS && j0 < M && L_c > 0) { q0_c = *(const uint4*)(bytes + s_c); q1_c = *(const uint4*)(bytes + s_c + 16); }
    for (u64 j = j0; j < M; j += stride) {
        u32 warm_n = 0;
        uint4 q0_n = make_uint4(0, 0, 0, 0), q1_n = q0_n;
        if (REGS) {
            if (L_n > 0) { q0_n = *(const uint4*)(bytes + s_n); q1_n = *(const uint4*)(bytes + s_n + 16); }
        } else if (L_n > 0) warm_n = *(const u32*)(bytes + s_n + (ws_n & ~3u));  // the next item's window start: brings its line(s) into L2
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
            else if (wmode == 1) {
                if (REGS) unicode_window_regs(nd, q0_c, q1_c, L, ws, we);
                else unicode_window_first_last(nd, hay, L, ws, we);
            } else if (wmode == 3) {  // a typo query decided by the scalar-LCS automaton: the lane-free typo window
                if (REGS) unicode_window_typos_regs(nd, q0_c, q1_c, L, (u32)nd.max_typos, ws, we);
                else unicode_window_typos(nd, hay, L, (u32)nd.max_typos, ws, we);
            }
            const u32 sp = ws ? ws - 1 : 0;
            const bool include_exact = sp == 0 && we == L;
            const u32 m = we - sp;
            if (m > (u32)SWL) {
                if (multi_front == 2) break;  // (queued before this kernel started: k2u_split_wide)
                // multi-chunk windows from the front (counters[3]: k2u_dp_unicode_multi), > 1024 bytes from the back (counters[4]: generic, greedy)
                const bool greedy = !multi_front || m > FZB_MAX_HAYSTACK_LEN;  // (multi_front == 0: every wide window to the generic kernel)
                // (two atomics with a uniform address each: the compiler turns those into one per wave; with the counter chosen per lane the
                // 45 k wide windows of the Arabic-shaped list cost 0.35 ms of serialised returning atomics)
                u32* qe;
                if (greedy) { const u32 slot = atomicAdd(&counters[4], 1u); qe = overflow + 4 * (size_t)(qcap - 1 - slot); }
                else { const u32 slot = atomicAdd(&counters[3], 1u); qe = overflow + 4 * (size_t)slot; }
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
                // TF: the biased-throughout form (dp_unicode_single_chunk_t; LaunchCfg::cfu_ok), else the first form
                constexpr int HR = SWL >= 16 ? SWL / 4 : SWL / 2;
                if (REGS) {
                    u32 hb[HR / 2 + 1];
                    u32 hw[HR / 2];
                    load_window_regs<HR / 2>(q0_c, q1_c, sp, m, hw);
#pragma unroll
                    for (int k = 0; k < HR / 2; k++) hb[k] = hw[k];
                    hb[HR / 2] = 0;
                    // the UTF-8 shortcut of the gap scan is taken when no window of the wave holds four continuation bytes in a row
                    score = dp_unicode_single_chunk_tr<SWL, HR>(nd, hb, m, sp == 0, cls, [](bool b) { return __all((int)b) != 0; });
                } else if (TF) {
                    // wave-uniform: every window of the wave is free of four-continuation-byte runs (always, for UTF-8 text)
                    const bool utf8 = __all((int)!(half ? unicode_has_cont_run4<(SWL >= 16 ? SWL / 2 : SWL)>(hay + sp, m) : unicode_has_cont_run4<SWL>(hay + sp, m)));
                    if (half) score = utf8 ? dp_unicode_single_chunk_t<SWL, HR, true>(nd, hay + sp, m, sp == 0, cls) : dp_unicode_single_chunk_t<SWL, HR, false>(nd, hay + sp, m, sp == 0, cls);
                    else if (!HALFONLY) score = utf8 ? dp_unicode_single_chunk_t<SWL, SWL / 2, true>(nd, hay + sp, m, sp == 0, cls) : dp_unicode_single_chunk_t<SWL, SWL / 2, false>(nd, hay + sp, m, sp == 0, cls);
                } else {
                    if (half) score = dp_unicode_single_chunk<SWL, HR>(nd, hay + sp, m, sp == 0, cls);
                    else if (!HALFONLY) score = dp_unicode_single_chunk<SWL>(nd, hay + sp, m, sp == 0, cls);
                }
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
        q0_c = q0_n; q1_c = q1_n;
        li_n = li_m; ws_n = ws_m; we_n = we_m; s_n = s_m; L_n = L_m;
        li_m = li_f; ws_m = ws_f; we_m = we_f;
        warm = warm_n;
    }
}

// When the window is the whole haystack (max_typos: None scores everything, wmode 2) its width is known from the end offsets alone: the
// wide windows are queued by this kernel BEFORE the single-chunk scorer runs (which then skips them: multi_front == 2), so that the queue's
// scorer can run beside it on a second stream instead of behind it - both are one-wave-per-SIMD kernels whose last round leaves most of the
// chip idle.  Queue entries as k2u_body writes them: (output position, window start, window end, haystack).
// (pushes are collected per workgroup in LDS and take ONE atomic on the queue's counter per 2048 items: an atomic per wave - 4.4 k of them on
// one address for the Arabic-shaped list - is served one at a time, 8 ns each, and made this kernel 53 us long)
#define FZB_SPLIT_CHUNK 2048u
__global__ __launch_bounds__(256) void k2u_split_wide(const EndsAny ends, u64 first, const u32* __restrict__ items, const u32* __restrict__ n_items_ptr, u32 ulen, u32 swl,
                                                      u32 capacity, u32* __restrict__ overflow, u32 qcap, u32* __restrict__ counters) {
    __shared__ uint4 s_buf[FZB_SPLIT_CHUNK];
    __shared__ u32 s_n, s_base;
    const u32 M = min(*n_items_ptr, capacity);
    const u32 nchunks = (M + FZB_SPLIT_CHUNK - 1) / FZB_SPLIT_CHUNK;
    for (u32 ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < FZB_SPLIT_CHUNK / 256; k++) {
            const u32 j = ch * FZB_SPLIT_CHUNK + k * 256 + threadIdx.x;
            if (j >= M) continue;
            const u32 li = items ? items[j] : j;
            u64 s;
            u32 L;
            haystack_span_u(ends, ulen, first + li, s, L);
            if (L <= swl) continue;
            const uint4 e = make_uint4(j, 0u, L, li);
            if (L > FZB_MAX_HAYSTACK_LEN) {  // the greedy fallback's end of the queue: rare, pushed directly
                const u32 slot = atomicAdd(&counters[4], 1u);
                *(uint4*)(overflow + 4 * (size_t)(qcap - 1 - slot)) = e;
            } else {
                s_buf[atomicAdd(&s_n, 1u)] = e;
            }
        }
        __syncthreads();
        const u32 n = s_n;
        if (threadIdx.x == 0 && n) s_base = atomicAdd(&counters[3], n);
        __syncthreads();
        const u32 base = s_base;
        for (u32 k = threadIdx.x; k < n; k += 256) *(uint4*)(overflow + 4 * (size_t)(base + k)) = s_buf[k];
        __syncthreads();
    }
}

void fzb_launch_unicode_split_wide(const CorpusDev& c, u64 first, const u32* items, const u32* n_items_ptr, int sw_lanes, u32 capacity, u32* overflow, u32 qcap, u32* counters,
                                   int grid, hipStream_t st) {
    hipLaunchKernelGGL(k2u_split_wide, dim3(grid), dim3(256), 0, st, EndsAny{c.ends, c.ends_u64}, first, items, n_items_ptr, c.uniform_len, (u32)sw_lanes, capacity, overflow, qcap, counters);
}

#define FZB_K2U_PARAMS const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset, const u32* __restrict__ items, const u32* __restrict__ win, \
    const u32* __restrict__ n_items_ptr, const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ overflow, \
    u32 qcap, u32* __restrict__ counters, u32 ulen, u32 multi_front
#define FZB_K2U_ARGS bytes, ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, qcap, counters, ulen, multi_front
template <int SWL, bool TF>
__global__ __launch_bounds__(128) void k2u_dp_unicode(FZB_K2U_PARAMS) { k2u_body<SWL, false, TF>(FZB_K2U_ARGS); }
// every haystack of the list fits the low half of a chunk (host-known: corpus max_len <= SWL / 2): the general form is compiled out,
// which lets the kernel fit 168 VGPRs (a few spills outside the row loop) and run three waves per SIMD
template <int SWL, bool TF>
__global__ __launch_bounds__(128, 3) void k2u_dp_unicode_half(FZB_K2U_PARAMS) { k2u_body<SWL, true, TF>(FZB_K2U_ARGS); }
// the same at two waves per SIMD (256 VGPRs: no spills in the row loop of the biased-throughout form); FZB_K2U_WAVES=2
template <int SWL, bool TF>
__global__ __launch_bounds__(128, 2) void k2u_dp_unicode_half_w2(FZB_K2U_PARAMS) { k2u_body<SWL, true, TF>(FZB_K2U_ARGS); }

void fzb_launch_dp_unicode(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                           int sw_lanes, int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters,
                           int grid, hipStream_t st, int tform, int multi_front, int one_round_wgs) {
    // one_round_wgs > 0: that many workgroups instead of the resident ones (one item per thread) - the caller runs another one-wave-per-SIMD
    // kernel beside this one, and short-lived workgroups take over the SIMDs that one frees; a persistent grid would hold what it got first
    // `grid` = number of CUs: the kernel is persistent, launch exactly the resident workgroups
    const bool half_only = sw_lanes >= 16 && c.max_len != 0 && c.max_len <= (u32)sw_lanes / 2;
    // the biased-throughout form wants ~250 registers: at two waves per SIMD it runs without spills (C5: 0.152 ms; capped at 168 registers /
    // three waves it spills inside the row loop: 0.199 ms; the first form at three waves: 0.161 ms)
    // (the half-only kernels: the biased form at two waves per SIMD, the first form at three - each instantiated for its own form only)
#define FZB_K2U_LAUNCH(KERNEL, DFLT)                                                                                                   \
    do {                                                                                                                               \
        static int per_cu = 0;                                                                                                         \
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 128, 0) != hipSuccess || per_cu < 1)) per_cu = DFLT; \
        hipLaunchKernelGGL(KERNEL, dim3(one_round_wgs > 0 ? one_round_wgs : grid * per_cu), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, qcap, counters, c.uniform_len, (u32)multi_front); \
    } while (0)
#define FZB_K2U_SW(SWL)                                                                                                                \
    do {                                                                                                                               \
        if (half_only && tform) FZB_K2U_LAUNCH((k2u_dp_unicode_half_w2<SWL, true>), 4);                                            \
        else if (half_only) FZB_K2U_LAUNCH((k2u_dp_unicode_half<SWL, false>), 4);                                                  \
        else if (tform) FZB_K2U_LAUNCH((k2u_dp_unicode<SWL, true>), 2);                                                            \
        else FZB_K2U_LAUNCH((k2u_dp_unicode<SWL, false>), 2);                                                                      \
    } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2U_SW(64); break;
        case 32: FZB_K2U_SW(32); break;
        case 16: FZB_K2U_SW(16); break;
        default: FZB_K2U_SW(8); break;
    }
#undef FZB_K2U_SW
#undef FZB_K2U_LAUNCH
}

// ---- windows of SWL < m <= 1024 bytes: one thread per queued window, chunk by chunk (dp_unicode_multi_chunk) -------------------------------
// The function keeps a chunk's row, previous row, pending and up masks, prefix counts and bonuses in registers (~ 300 live values at 64
// lanes): one wave per SIMD, the accumulation registers as spill space - still 64 haystacks per wavefront where the generic kernel takes one.
template <int SWL, bool TF>
__global__ __launch_bounds__(128) void k2u_dp_unicode_multi(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset, const u32* __restrict__ list,
                                                            const u32* __restrict__ n_list_ptr, const NeedleDev nd, fzb_match_rec* __restrict__ out, u32 capacity,
                                                            u32* __restrict__ scratch, u32 only_from, u32* __restrict__ counters, u32* __restrict__ back_end, u32 fwd_cap) {
    // STRAGGLERS: a window of ten chunks keeps its thread - and the kernel - for ten times the latency of a chunk (Arabic-shaped list, All
    // Scores: 113 us for 45 k windows of which a few dozen are that long), while the wave-per-haystack kernel walks the same window in
    // ~ 15 us.  Windows beyond four chunks are therefore handed on - appended to the BACK of the queue (counters[4]; the generic kernel
    // behind this one scores whatever is there: by DP up to 1024 bytes, greedily beyond) - as long as fewer than fwd_cap have been
    // (counters[7] counts the claims: a list whose windows are all long keeps them here, where the throughput is)
    const u32 nlist = *n_list_ptr;
    if (nlist < only_from) return;  // a short queue is the wave-per-haystack kernel's (launched behind this one with the complementary test)
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 nthreads = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    for (u32 q = gtid; q < nlist; q += nthreads) {
        const u32 opos = list[4 * q], ws = list[4 * q + 1], we = list[4 * q + 2], li = list[4 * q + 3];
        if (opos >= capacity) continue;
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        const u8* hay = bytes + s;
        const u32 sp = ws ? ws - 1 : 0;
        const bool include_exact = sp == 0 && we == L;
        const u32 m = we - sp;
        if (m > 4u * (u32)SWL && fwd_cap) {
            if (atomicAdd(&counters[7], 1u) < fwd_cap) {
                u32* qe = back_end - 4 * (size_t)(atomicAdd(&counters[4], 1u) + 1u);
                qe[0] = opos;
                qe[1] = ws;
                qe[2] = we;
                qe[3] = li;
                continue;
            }
        }
        u32 score = 0;
        if (nd.rows > 0) {
            if (TF) {  // the biased-throughout form (LaunchCfg::cfu_ok); its UTF-8 shortcut when no window of the wave has four continuation bytes in a row
                const bool utf8 = __all((int)!unicode_window_has_cont_run4(hay + sp, m)) != 0;
                score = utf8 ? dp_unicode_multi_chunk_t<SWL, true>(nd, hay + sp, m, sp == 0, cls, scratch, nthreads, gtid)
                             : dp_unicode_multi_chunk_t<SWL, false>(nd, hay + sp, m, sp == 0, cls, scratch, nthreads, gtid);
            } else {
                score = dp_unicode_multi_chunk<SWL>(nd, hay + sp, m, sp == 0, cls, scratch, nthreads, gtid);
            }
        }
        bool exact = include_exact && m == (u32)nd.nbytes;
        if (exact)
            for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
        if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = exact ? 1 : 0;
        rec.valid = 0;
        out[opos] = rec;
    }
}

void fzb_launch_dp_unicode_multi(const CorpusDev& c, u64 first, u32 index_offset, const u32* list, const u32* n_list_ptr, const NeedleDev& nd, int sw_lanes,
                                 fzb_match_rec* out, u32 capacity, u32* scratch, int grid, hipStream_t st, u32 only_from, int tform, u32* counters, u32* back_end, u32 fwd_cap) {
#define FZB_K2UM(SWL, TF) hipLaunchKernelGGL((k2u_dp_unicode_multi<SWL, TF>), dim3(grid), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, list, n_list_ptr, nd, out, capacity, scratch, only_from, counters, back_end, fwd_cap)
#define FZB_K2UM_ET(SWL) do { if (tform) FZB_K2UM(SWL, true); else FZB_K2UM(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2UM_ET(64); break;
        case 32: FZB_K2UM_ET(32); break;
        case 16: FZB_K2UM_ET(16); break;
        default: FZB_K2UM_ET(8); break;
    }
#undef FZB_K2UM_ET
#undef FZB_K2UM_TF
#undef FZB_K2UM
}
