// gfx950 kernels, stage 2b: Smith-Waterman scoring of the survivors whose trimmed window fits ONE chunk of
// the emulated CPU backend (<= SWL bytes; every len-32 haystack).  One THREAD per haystack: the SWL-lane
// score row lives in SWL/2 VGPRs as packed u16 pairs and is updated with v_pk_{add,sub(clamp),max,mul_lo}_u16;
// the reference's `shift_right_padded::<L>` for L >= 2 lanes is a compile-time register renaming, L = 1 is
// one v_alignbit_b32.  No MFMA: this is saturating integer DP, not a contraction.
//
// Restates, bit-exactly for a single chunk (adjacent chunk = the zero column):
//   score_haystack                src/smith_waterman/algo/ascii.rs:10-158
//   propagate_horizontal_gaps     src/smith_waterman/algo/ascii_gap.rs:11-105   (log-step scan, steps 1..SWL/2)
//   trim_haystack / smith_waterman_one / exact bonus   src/matcher/algo.rs:230-263, 332-338
//   0-typo ASCII window (first occurrence of needle[0], last occurrence of needle[n-1])  src/prefilter/algo/ascii.rs:6-72
// Windows wider than one chunk (or > 1024 bytes) are handed to the generic scorer (kernels_generic.hip).
//
// Gap propagation runs in a biased domain r'[L] = r[L] + L*gex, where "shift by k and pay k*gex" becomes a
// plain shift; since every row value is >= 0 the saturating subtract of the reference is preserved exactly
// (see DESIGN.md).  BIAS=false keeps the literal 3-op form for scorings where the bias could overflow u16.
#include "dp_cfm.h"
#include "dp_quad.h"
#include "compact1.h"
#include <cstdlib>

// MODE 0: literal gap scan (the bias could overflow u16), 1: biased scan (dp_body.h).  (dp_cf.h's form of this per-wave kernel - rounds 2-5 -
// only ever ran behind FZB_NO_DP_CLASSES once the classified path existed and left with round 6's prune.)
template <int SWL, int MODE, bool UPPER>
__global__ __launch_bounds__(128, 2) void k2b_dp(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                              const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                              const NeedleDev nd, int wmode, int pad_ok, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count,
                                              u32* __restrict__ overflow, u32 qcap, u32* __restrict__ counters) {
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 M = __builtin_amdgcn_readfirstlane(*n_items_ptr);  // wave-uniform: keeps the loop control on the scalar unit
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = M < capacity ? M : capacity; dev_count[1] = M; }  // [1] = the untruncated total
    // Persistent threads with a three-deep software pipeline over the dependent loads of one item
    // (survivor index / window -> end offsets -> haystack vectors): each stage is requested one iteration before it is
    // needed, so the ~7 us of integer DP of the current item cover the latency and only the prologue waits on memory.
    // (With ~200 VGPRs only two waves fit a SIMD, and they run in phase: without this every iteration began with both
    // of them stalled on a three-deep dependent load chain.)
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
        if (j < M) haystack_span(ends, first + li, s, L);
    };
    auto load_vecs = [&](u64 s, u32 L, uint4& q0, uint4& q1) {
        q0 = make_uint4(0, 0, 0, 0);
        q1 = make_uint4(0, 0, 0, 0);
        const uint4* vp = (const uint4*)(bytes + s);
        if (L > 0) q0 = vp[0];
        if (L > 16) q1 = vp[1];
    };
    u32 li_c, ws_c, we_c, L_c, li_n, ws_n, we_n, L_n, li_m, ws_m, we_m;
    u64 s_c, s_n;
    uint4 q0_c, q1_c;
    load_item(j0, li_c, ws_c, we_c);
    load_item(j0 + stride, li_n, ws_n, we_n);
    load_item(j0 + 2 * stride, li_m, ws_m, we_m);
    load_span(j0, li_c, s_c, L_c);
    load_span(j0 + stride, li_n, s_n, L_n);
    load_vecs(s_c, L_c, q0_c, q1_c);
    for (u64 j = j0; j < M; j += stride) {
        // requests for the following iterations
        uint4 q0_n, q1_n;
        load_vecs(s_n, L_n, q0_n, q1_n);
        u64 s_m;
        u32 L_m;
        load_span(j + 2 * stride, li_m, s_m, L_m);
        u32 li_f, ws_f, we_f;
        load_item(j + 3 * stride, li_f, ws_f, we_f);
        // ---- this iteration's item ----------------------------------------------------------------------
        do {
            if (j >= capacity) break;
            const u32 li = li_c, L = L_c;
            const u8* hay = bytes + s_c;
            const bool inreg = __all((int)(L <= 32));  // wave-uniform: the two prefetched vectors hold every haystack of the wave
            u32 ws = ws_c, we = we_c;
            if (wmode == 2) { ws = 0; we = L; }
            else if (wmode == 1) {
                if (inreg) window_first_last_regs(nd, q0_c, q1_c, L, ws, we);
                else window_first_last(nd, hay, L, ws, we);
            }
            // ---- trim_haystack (matcher/algo.rs:332-338) --------------------------------------------------
            const u32 sp = ws ? ws - 1 : 0;
            const bool include_exact = sp == 0 && we == L;
            const u32 m = we - sp;
            // wider than one chunk: queued for the multi-chunk kernel (counters[3], front of `overflow`), or - beyond the
            // reference's 1024-byte matrix limit - for the generic kernel's greedy scorer (counters[4], back of `overflow`)
            const bool wide = m > (u32)SWL, greedy = m > FZB_MAX_HAYSTACK_LEN;
            const u32 slot_multi = wave_alloc(&counters[3], wide && !greedy);
            const u32 slot_greedy = wave_alloc(&counters[4], greedy);
            if (wide) {
                u32* qe = greedy ? overflow + 4 * (size_t)(qcap - 1 - slot_greedy) : overflow + 4 * (size_t)slot_multi;
                qe[0] = (u32)j;  // (output position, window start, window end, local haystack index)
                qe[1] = ws;
                qe[2] = we;
                qe[3] = li;
                break;
            }
            u32 score = 0;
            u32 hb[SWL / 4];
#pragma unroll
            for (int k = 0; k < SWL / 4; k++) hb[k] = 0;
            if (m > 0) {
                if (inreg) load_window_regs<SWL / 4>(q0_c, q1_c, sp, m, hb);
                else load_window_mem<SWL / 4>(hay + sp, m, hb);
                // wave-uniform choice: if every window in this wave fits the low half of the chunk, the upper half is pure padding
                const bool half = pad_ok && SWL >= 16 && __all((int)(m <= (u32)SWL / 2));
                if (half) score = dp_single_chunk<SWL, MODE == 1, UPPER, (SWL >= 16 ? SWL / 4 : SWL / 2)>(nd, m, sp == 0, cls, hb);
                else score = dp_single_chunk<SWL, MODE == 1, UPPER>(nd, m, sp == 0, cls, hb);
            }
            const bool exact = exact_match<SWL / 4>(nd, include_exact, m, hb);
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li;
            rec.score = (u16)score;
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            out[j] = rec;
        } while (0);
        // ---- rotate the pipeline ------------------------------------------------------------------------
        li_c = li_n; ws_c = ws_n; we_c = we_n; s_c = s_n; L_c = L_n; q0_c = q0_n; q1_c = q1_n;
        li_n = li_m; ws_n = ws_m; we_n = we_m; s_n = s_m; L_n = L_m;
        li_m = li_f; ws_m = ws_f; we_m = we_f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k2b_dp_short: the same scorer for a corpus whose longest haystack fits half a chunk (and two 16-byte vectors): every
// window is scored in dp_cf.h's form with SWL/2 computed lanes, the haystack never leaves the two prefetched vectors, the
// 0-typo window comes from one merged flag word per needle byte and the per-lane bonuses from two small LDS tables.  No
// queues (nothing can be wider than a chunk) and a third of the registers of the general kernel: four waves per SIMD.
// Chosen by fzb_launch_dp when LaunchCfg::cf_ok and CorpusDev::max_len allow; otherwise k2b_dp runs.
// ---------------------------------------------------------------------------------------------------------------
#ifdef FZB_DP_TIMING
// Debug build only (make TIMING=1 -> libfrizbee_hip_timing.so, tools/exp_dp_timing.py): per wave, the constant 100 MHz counter
// (s_memrealtime) and the shader-clock counter (s_memtime) at entry and exit, to split the kernel's duration into dispatch ramp,
// resident time, tail and effective clock.
__device__ unsigned long long fzb_dp_timing[6 * 8192];
extern "C" int fzb_debug_dp_timing(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fzb_dp_timing), sizeof(fzb_dp_timing)); }
#define FZB_TIMING_BEGIN const unsigned long long t_rt0 = wall_clock64(), t_ck0 = clock64();
#define FZB_TIMING_END                                                                                  \
    if ((threadIdx.x & 63) == 0) {                                                                      \
        const u32 w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                     \
        if (w < 8192) { fzb_dp_timing[6 * w] = t_rt0; fzb_dp_timing[6 * w + 1] = wall_clock64(); fzb_dp_timing[6 * w + 2] = t_ck0; fzb_dp_timing[6 * w + 3] = clock64(); \
                        fzb_dp_timing[6 * w + 4] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); fzb_dp_timing[6 * w + 5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); } /* HW_ID, XCC_ID */ \
    }
#else
#define FZB_TIMING_BEGIN
#define FZB_TIMING_END
#endif
template <int SWL, bool UPPER>
__global__ __launch_bounds__(128, 4) void k2b_dp_short(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                                    const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                                    const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count,
                                                    RejectOut rej, u32* __restrict__ kept_out, u32 ulen) {
    FZB_TIMING_BEGIN
    __shared__ CfTables tab;
    __shared__ u8 fl[256];
    cf_build_tables<UPPER>(nd, tab);
    if (wmode == 3) cf_build_typo_table(nd, fl);
    __syncthreads();
    const u32 M = __builtin_amdgcn_readfirstlane(*n_items_ptr);  // wave-uniform: keeps the loop control on the scalar unit
    // wmode 3 (typos): survivors that the decide pass rejected (rare) are skipped and the records behind them move up
    const u32 nrej = wmode == 3 ? __builtin_amdgcn_readfirstlane(*rej.count) : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u32 kept = M - nrej;
        if (dev_count) { dev_count[0] = kept < capacity ? kept : capacity; dev_count[1] = kept; }
        if (kept_out) *kept_out = kept;
    }
    // persistent threads, three-deep software pipeline over the dependent loads of one item (see k2b_dp)
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
    auto load_vecs = [&](u64 s, u32 L, uint4& q0, uint4& q1) {
        q0 = make_uint4(0, 0, 0, 0);
        q1 = make_uint4(0, 0, 0, 0);
        const uint4* vp = (const uint4*)(bytes + s);
        if (L > 0) q0 = vp[0];
        if (SWL > 32 && L > 16) q1 = vp[1];
    };
    u32 li_c, ws_c, we_c, L_c, li_n, ws_n, we_n, L_n, li_m, ws_m, we_m;
    u64 s_c, s_n;
    uint4 q0_c, q1_c;
    load_item(j0, li_c, ws_c, we_c);
    load_item(j0 + stride, li_n, ws_n, we_n);
    load_item(j0 + 2 * stride, li_m, ws_m, we_m);
    load_span(j0, li_c, s_c, L_c);
    load_span(j0 + stride, li_n, s_n, L_n);
    load_vecs(s_c, L_c, q0_c, q1_c);
    // issue priority follows the wave's progress through its share (kernels_common.h, FzbProgressPrio): wave-uniform scalars
    const u32 jw = __builtin_amdgcn_readfirstlane((u32)j0);  // the wave's first item (lane 0's)
    const u32 n_iter = __builtin_amdgcn_readfirstlane(jw < M ? (u32)(((u64)M - jw + stride - 1) / stride) : 1u);
    struct MidItem {
        FzbProgressPrio* p;
        __device__ __forceinline__ void operator()() const { p->k2 += 1; p->apply(); }
    };
    FzbProgressPrio prio{0u, 2 * n_iter};
    for (u64 j = j0; j < M; j += stride) {
        prio.apply();
        uint4 q0_n, q1_n;
        load_vecs(s_n, L_n, q0_n, q1_n);
        u64 s_m;
        u32 L_m;
        load_span(j + 2 * stride, li_m, s_m, L_m);
        u32 li_f, ws_f, we_f;
        load_item(j + 3 * stride, li_f, ws_f, we_f);
        u64 jo = j;  // output position
        bool live = true;
        if (nrej) {  // rare: some marginal survivor was rejected at the exact lane width
            const u32 li = li_c, tile = li / FZB_TILE;
            const u64 word = rej.bits[li >> 6];
            live = !((word >> (li & 63)) & 1);
            u32 before = rej.rej_prefix[tile] + (u32)__popcll(word & ((1ull << (li & 63)) - 1));
            for (u32 wq = tile * (FZB_TILE / 64); wq < (li >> 6); wq++) before += (u32)__popcll(rej.bits[wq]);
            jo = j - before;
        }
        if (live && jo < capacity) {
            const u32 L = L_c;
            u32 ws = ws_c, we = we_c;
            if (wmode == 2) { ws = 0; we = L; }
            else if (wmode == 1) cf_window_first_last_regs(nd, q0_c, q1_c, ws, we);
            else if (wmode == 3) cf_window_typos_regs(fl, q0_c, q1_c, L, ws, we);
            // ---- trim_haystack (matcher/algo.rs:332-338) --------------------------------------------------
            const u32 sp = ws ? ws - 1 : 0;
            const bool include_exact = sp == 0 && we == L;
            const u32 m = we - sp;
            u32 score = 0;
            u32 hb[SWL / 4];
#pragma unroll
            for (int k = 0; k < SWL / 4; k++) hb[k] = 0;
            if (m > 0) {
                load_window_regs<SWL / 4>(q0_c, q1_c, sp, m, hb);
                score = dp_single_chunk_cf_tab<SWL, UPPER, SWL / 4, MidItem>(nd, sp == 0, tab, hb, MidItem{&prio});
            }
            const bool exact = exact_match<SWL / 4>(nd, include_exact, m, hb);
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li_c;
            rec.score = (u16)score;
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            out[jo] = rec;
        }
        li_c = li_n; ws_c = ws_n; we_c = we_n; s_c = s_n; L_c = L_n; q0_c = q0_n; q1_c = q1_n;
        li_n = li_m; ws_n = ws_m; we_n = we_m; s_n = s_m; L_n = L_m;
        li_m = li_f; ws_m = ws_f; we_m = we_f;
        prio.k2 = (prio.k2 | 1u) + 1;  // next item
    }
    FZB_TIMING_END
}

// ---------------------------------------------------------------------------------------------------------------
// Corpora with longer haystacks (dp_cf.h form, LaunchCfg::cf_ok): windows are found and CLASSIFIED first, then every class is scored
// by its own launch, so that a wave never mixes window widths (k2b_dp decides per wave: on a ragged list nearly every wave holds one
// window of the widest class and runs all 64 lanes at 2 waves per SIMD).
//   k2w_classify: one thread per survivor - window (0 typos: first / last occurrence; typos / unicode: from the prefilter kernel;
//                 no prefilter: the whole haystack), trimmed length m, class:
//                   0: m <= SWL/2   1: m <= 3 SWL/4   2: m <= SWL   (single chunk, SWL/2 | 3SWL/4 | SWL computed lanes)
//                   multi-chunk (<= 1024 bytes) and greedy (> 1024) go to the `overflow` queue exactly as k2b_dp queues them.
//                 A workgroup appends its members of a class with ONE global atomic per class (order inside a class is free:
//                 every entry carries its output position).
//   k2b_dp_class: persistent thread-per-survivor scorer of one class: window bytes from memory (requested one item ahead), bonuses
//                 from the LDS tables, dp_cf.h rows with the class's number of computed lanes; registers - and therefore waves per
//                 SIMD (4 / 3 / 2) - follow the class.
//                 It writes, per survivor, one 16-byte record (window start, window end | bit 31 = "the window is the whole haystack", the
//                 64-bit address of the haystack's first byte).  The scorers read that record and nothing else: no end offsets.
// ---------------------------------------------------------------------------------------------------------------
// One tile of the classifier: the workgroup's 256 threads take the survivors j0 .. j0 + 256 * PER - 1 (those below j_end and capacity), PER per
// thread.  Called by every thread of the workgroup (barriers inside); s_cnt / s_base: nine words of LDS each.
template <int PER>
__device__ __forceinline__ void classify_tile(u32 j0, u32 j_end, u32* s_cnt, u32* s_base, const u8* __restrict__ bytes, const EndsAny& ends, u64 first, const u32* __restrict__ items,
                                              const u32* __restrict__ win_in, const NeedleDev& nd, int wmode, u32 swl, uint4* __restrict__ meta, u32* __restrict__ lists, u32 list_stride,
                                              u32* __restrict__ overflow, u32 qcap, u32* __restrict__ counters, u32 capacity, u32 split_multi) {
    if (threadIdx.x < 9) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    u32 cls[PER], rank[PER], li[PER], ws[PER], we[PER];
    const u8* src[PER];
#pragma unroll
    for (int p = 0; p < PER; p++) {
        const u32 j = j0 + p * 256 + threadIdx.x;
        cls[p] = 9; rank[p] = 0; li[p] = 0; ws[p] = 0; we[p] = 0; src[p] = bytes;
        if (j < j_end && j < capacity) {
            li[p] = items ? items[j] : j;
            u32 L = 0;
            u64 s;
            haystack_span(ends, first + li[p], s, L);
            src[p] = bytes + s;
            if (wmode == 0) { ws[p] = win_in[2 * j]; we[p] = win_in[2 * j + 1]; }
            else if (wmode == 2) { ws[p] = 0; we[p] = L; }
            else window_first_last(nd, src[p], L, ws[p], we[p]);
            const u32 sp = ws[p] ? ws[p] - 1 : 0;
            const u32 m = we[p] - sp;
            if (sp == 0 && we[p] == L) we[p] |= 0x80000000u;  // include_exact (src/matcher/algo.rs:238-249): the window is the whole haystack
            cls[p] = m <= swl / 2 ? 0u : m <= 3 * swl / 4 ? 1u : m <= swl ? 2u : m <= FZB_MAX_HAYSTACK_LEN ? 3u : 4u;
            if (split_multi && cls[p] == 3) cls[p] = 5 + ((m - 1) % swl) / (swl / 4);  // tail of 1 ..= swl bytes -> 0 ..= 3
            rank[p] = atomicAdd(&s_cnt[cls[p]], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 9 && s_cnt[threadIdx.x]) {
        const u32 c = threadIdx.x;
        s_base[c] = atomicAdd(c < 3 ? &counters[8 + c] : c == 3 ? &counters[3] : c == 4 ? &counters[4] : &counters[12 + (c - 5)], s_cnt[c]);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PER; p++) {
        const u32 j = j0 + p * 256 + threadIdx.x;
        if (cls[p] < 3 || (cls[p] >= 5 && cls[p] < 9)) {
            const u32 l = cls[p] < 3 ? cls[p] : cls[p] - 2;
            lists[(size_t)l * list_stride + s_base[cls[p]] + rank[p]] = j;
            meta[j] = make_uint4(ws[p], we[p], (u32)(uintptr_t)src[p], (u32)((uintptr_t)src[p] >> 32));
        } else if (cls[p] < 5) {
            const u32 slot = s_base[cls[p]] + rank[p];
            u32* qe = cls[p] == 4 ? overflow + 4 * (size_t)(qcap - 1 - slot) : overflow + 4 * (size_t)slot;
            qe[0] = j;  // (output position, window start, window end, local haystack index)
            qe[1] = ws[p];
            qe[2] = we[p] & 0x7FFFFFFFu;
            qe[3] = li[p];
        }
    }
    __syncthreads();
}

template <int PER>
__global__ __launch_bounds__(256) void k2w_classify(const u8* __restrict__ bytes, const EndsAny ends, u64 first, const u32* __restrict__ items,
                                                    const u32* __restrict__ win_in, const u32* __restrict__ n_items_ptr, const NeedleDev nd, int wmode, u32 swl,
                                                    uint4* __restrict__ meta, u32* __restrict__ lists, u32 list_stride, u32* __restrict__ overflow, u32 qcap,
                                                    u32* __restrict__ counters, u32 capacity, u32* __restrict__ dev_count, u32 split_multi) {
    // classes: 0-2 single chunk, 3 multi-chunk (queue), 4 greedy (queue, from the back), 5-8 multi-chunk by the width of the LAST chunk's
    // tail (split_multi: lists 3-6, counts in counters[12..15]; k2d_dp_multi_tc computes only that many lanes of the last chunk)
    __shared__ u32 s_cnt[9], s_base[9];
    const u32 M = __builtin_amdgcn_readfirstlane(*n_items_ptr);
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = M < capacity ? M : capacity; dev_count[1] = M; }  // [1] = the untruncated total
    // a workgroup takes 256 * PER survivors at a time (PER per thread) and appends its members of a class with ONE global atomic per class
    // and tile: atomics that return a value to the same address serialise in L2 (one per 256 survivors cost ~40 us on the 0.6 M
    // survivors of the ragged list)
    for (u32 j0 = blockIdx.x * (256 * PER); j0 < M; j0 += gridDim.x * (256 * PER))  // uniform trip count per workgroup
        classify_tile<PER>(j0, M, s_cnt, s_base, bytes, ends, first, items, win_in, nd, wmode, swl, meta, lists, list_stride, overflow, qcap, counters, capacity, split_multi);
}

// Compaction and classification of a ragged ASCII list in ONE launch (round 6): k_compact1's workgroups - each owns a contiguous run of tiles and
// derives its survivors' output positions itself (compact1.h) - classify the survivors they have just listed, a batch of tiles at a time, instead
// of a second, dependent launch finding them again through the list.  Same lists, records and counters as k_compact1 + k2w_classify.
template <int PER>
__global__ __launch_bounds__(256) void k_compact1_classify(const u64* __restrict__ bitmap, const u32* __restrict__ tile_counts, u32 n_items, u32* __restrict__ out_idx,
                                                           u32* __restrict__ total_out, const u8* __restrict__ bytes, const EndsAny ends, u64 first, const NeedleDev nd, int wmode,
                                                           u32 swl, uint4* __restrict__ meta, u32* __restrict__ lists, u32 list_stride, u32* __restrict__ overflow, u32 qcap,
                                                           u32* __restrict__ counters, u32 capacity, u32* __restrict__ dev_count, u32 split_multi) {
    __shared__ u32 s_cnt[9], s_base[9];
    compact1_body(
        bitmap, tile_counts, n_items, nullptr, nullptr, out_idx, total_out, nullptr,
        [&](u32 pos0, u32 nsurv) {
            __threadfence_block();  // the batch's indices, written by other waves of this workgroup, are read below
            __syncthreads();
            // (one survivor per thread when the batch has no more than that: a second survivor is a second, dependent walk through its haystack)
            if (nsurv <= 256)
                classify_tile<1>(pos0, pos0 + nsurv, s_cnt, s_base, bytes, ends, first, out_idx, nullptr, nd, wmode, swl, meta, lists, list_stride, overflow, qcap, counters, capacity, split_multi);
            else
                for (u32 j0 = pos0; j0 < pos0 + nsurv; j0 += 256 * PER)
                    classify_tile<PER>(j0, pos0 + nsurv, s_cnt, s_base, bytes, ends, first, out_idx, nullptr, nd, wmode, swl, meta, lists, list_stride, overflow, qcap, counters, capacity, split_multi);
        },
        [&](u32 total) {
            if (dev_count) { dev_count[0] = total < capacity ? total : capacity; dev_count[1] = total; }
        });
}

// (vblock of vgrid: the workgroup's index among those that walk this list - blockIdx / gridDim in the class's own launch, a slice of the grid
// in k2_classes_all)
template <int SWL, bool UPPER, int REAL>
__device__ __forceinline__ void dp_class_body(const CfTables& tab, u32 vblock, u32 vgrid, u32 index_offset,
                                              const u32* __restrict__ items, const uint4* __restrict__ meta, const u32* __restrict__ list, const u32* __restrict__ n_list_ptr,
                                              const NeedleDev& nd, fzb_match_rec* __restrict__ out) {
    constexpr int NB = SWL / 4;           // window dwords of a full chunk
    constexpr int NBR = (REAL + 1) / 2;   // dwords that can hold window bytes of this class
    const u32 M = __builtin_amdgcn_readfirstlane(*n_list_ptr);
    const u32 stride = vgrid * blockDim.x;
    const u32 q0 = vblock * blockDim.x + threadIdx.x;
    // two-deep pipeline: (list entry -> haystack index, the classifier's record) one item ahead of (window bytes), which are one item ahead of the DP
    auto load_meta = [&](u32 q, u32& j, u32& li, u32& sp, u32& m, u32& L, u64& s) {
        j = 0; li = 0; sp = 0; m = 0; L = 0; s = (u64)(uintptr_t)meta;  // (an address that can be read: the slot's bytes are requested unconditionally only when m > 0)
        if (q < M) {
            j = list[q];
            li = items ? items[j] : j;
            const uint4 w = meta[j];
            sp = w.x ? w.x - 1 : 0;
            m = (w.y & 0x7FFFFFFFu) - sp;
            L = w.y >> 31;  // include_exact
            s = (u64)w.z | ((u64)w.w << 32);
        }
    };
    auto load_bytes = [&](u64 s, u32 sp, u32 m, u32 (&hb)[NBR]) {
        const u8* th = (const u8*)(uintptr_t)s + sp;
#pragma unroll
        for (int k = 0; k < NBR; k++) {
            const u32 p = 4 * k;
            u32 v = 0;
            if (p < m) {
                v = load_u32_unaligned(th, p);
                const u32 rem = m - p;
                if (rem < 4) v &= (1u << (8 * rem)) - 1;
            }
            hb[k] = v;
        }
    };
    u32 j_c, li_c, sp_c, m_c, ex_c, j_n, li_n, sp_n, m_n, ex_n;
    u64 s_c, s_n;
    u32 hb_c[NBR];
    load_meta(q0, j_c, li_c, sp_c, m_c, ex_c, s_c);
    load_meta(q0 + stride, j_n, li_n, sp_n, m_n, ex_n, s_n);
    load_bytes(s_c, sp_c, m_c, hb_c);
    const u32 qw = __builtin_amdgcn_readfirstlane(q0);
    const u32 n_iter = __builtin_amdgcn_readfirstlane(qw < M ? (M - qw + stride - 1) / stride : 1u);
    struct MidItem {
        FzbProgressPrio* p;
        __device__ __forceinline__ void operator()() const { p->k2 += 1; p->apply(); }
    };
    FzbProgressPrio prio{0u, 2 * n_iter};
    for (u32 q = q0; q < M; q += stride) {
        prio.apply();
        u32 hb_n[NBR];
        load_bytes(s_n, sp_n, m_n, hb_n);
        u32 j_f, li_f, sp_f, m_f, ex_f;
        u64 s_f;
        load_meta(q + 2 * stride, j_f, li_f, sp_f, m_f, ex_f, s_f);
        {
            u32 hb[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) hb[k] = k < NBR ? hb_c[k < NBR ? k : 0] : 0u;
            u32 score = 0;
            if (m_c > 0) score = dp_single_chunk_cf_tab<SWL, UPPER, REAL, MidItem>(nd, sp_c == 0, tab, hb, MidItem{&prio});
            const bool exact = exact_match<NB>(nd, ex_c != 0, m_c, hb);
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li_c;
            rec.score = (u16)score;
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            out[j_c] = rec;
        }
        j_c = j_n; li_c = li_n; sp_c = sp_n; m_c = m_n; ex_c = ex_n; s_c = s_n;
#pragma unroll
        for (int k = 0; k < NBR; k++) hb_c[k] = hb_n[k];
        j_n = j_f; li_n = li_f; sp_n = sp_f; m_n = m_f; ex_n = ex_f; s_n = s_f;
        prio.k2 = (prio.k2 | 1u) + 1;  // next item
    }
}

template <int SWL, bool UPPER, int REAL>
__global__ __launch_bounds__(128, (REAL * 4 <= SWL ? 4 : REAL * 8 <= 3 * SWL ? 3 : 2)) void k2b_dp_class(
    u32 index_offset, const u32* __restrict__ items, const uint4* __restrict__ meta, const u32* __restrict__ list, const u32* __restrict__ n_list_ptr, const NeedleDev nd,
    fzb_match_rec* __restrict__ out) {
    __shared__ CfTables tab;
    cf_build_tables<UPPER>(nd, tab);
    __syncthreads();
    dp_class_body<SWL, UPPER, REAL>(tab, blockIdx.x, gridDim.x, index_offset, items, meta, list, n_list_ptr, nd, out);
}

// k_compact1 + k2w_classify in one launch (k_compact1_classify): the bitmap and tile counts of the streaming filter -> the survivor list, its length
// (total_out), and the classifier's lists / records / queue entries / counters / dev_count.  0-typo and whole-haystack windows only (wmode 1 / 2).
void fzb_launch_compact1_classify(const CorpusDev& c, u64 first, const u64* bitmap, const u32* tile_counts, u32 n_items, u32* out_idx, u32* total_out, const NeedleDev& nd, int sw_lanes,
                                  int wmode, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, u32* win_out, u32* lists, u32 list_stride, int grid, hipStream_t st,
                                  int split_multi) {
    hipLaunchKernelGGL((k_compact1_classify<3>), dim3(grid), dim3(256), 0, st, bitmap, tile_counts, n_items, out_idx, total_out, c.bytes, EndsAny{c.ends, c.ends_u64}, first, nd, wmode,
                       (u32)sw_lanes, (uint4*)win_out, lists, list_stride, overflow, qcap, counters, capacity, dev_count, (u32)split_multi);
}

void fzb_launch_dp_classes(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win_in, const u32* n_items_ptr, const NeedleDev& nd, int sw_lanes,
                           int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, u32* win_out, u32* lists, u32 list_stride,
                           int num_cus, hipStream_t st, int part, int split_multi) {
    // part: 0 = classify + the three class launches, 1 = classify only, 2 = the class launches only (host.hip runs the multi-chunk scorer on a
    // second stream between the two)
    bool upper = false;
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    if (part != 2) {
        // (two survivors per thread: 1 / 2 / 4 gave 0.458 / 0.454 / 0.464 ms for the C4 shard's step)
#define FZB_K2W(ET) hipLaunchKernelGGL((k2w_classify<2>), dim3(num_cus * 8), dim3(256), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, items, win_in, n_items_ptr, nd, wmode, (u32)sw_lanes, (uint4*)win_out, lists, list_stride, overflow, qcap, counters, capacity, dev_count, (u32)split_multi)
        FZB_K2W(u32);
#undef FZB_K2W
    }
    if (part == 1) return;
#define FZB_K2C(SWL, U, REAL, CLS)                                                                                                        \
    do {                                                                                                                                  \
        static int per_cu = 0;                                                                                                            \
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2b_dp_class<SWL, U, REAL>, 128, 0) != hipSuccess || per_cu < 1)) per_cu = 4; \
        hipLaunchKernelGGL((k2b_dp_class<SWL, U, REAL>), dim3(num_cus * per_cu), dim3(128), 0, st, index_offset, items, (const uint4*)win_out, lists + (size_t)CLS * list_stride, &counters[8 + CLS], nd, out); \
    } while (0)
#define FZB_K2C_ALL(SWL, U) do { FZB_K2C(SWL, U, SWL / 4, 0); FZB_K2C(SWL, U, 3 * SWL / 8, 1); FZB_K2C(SWL, U, SWL / 2, 2); } while (0)
#define FZB_K2C_U(SWL) do { if (upper) FZB_K2C_ALL(SWL, true); else FZB_K2C_ALL(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2C_U(64); break;
        case 32: FZB_K2C_U(32); break;
        case 16: FZB_K2C_U(16); break;
        default: FZB_K2C_U(8); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k2d: the queued windows of SWL < m <= 1024 bytes, one thread each, chunk by chunk (dp_multi_chunk).
// ---------------------------------------------------------------------------------------------------------------
template <int SWL, bool BIAS>
__global__ __launch_bounds__(128, 2) void k2d_dp_multi(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                                    const u32* __restrict__ list, const u32* __restrict__ n_list_ptr, const NeedleDev nd,
                                                    fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ scratch) {
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 nlist = *n_list_ptr;
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
        u32 score = dp_multi_chunk<SWL, BIAS>(nd, hay + sp, m, sp == 0, cls, scratch, nthreads, gtid);
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

// ---------------------------------------------------------------------------------------------------------------
// k2d_dp_long: LONG needles (NeedleLongDev: beyond 64 bytes / 63 rows, any length the reference's guard accepts), ASCII.  Round 3-4 scored
// them with the wave-per-haystack kernel only (a lane per DP column, half the lanes idle at the u16 class' 32-lane chunks, a shuffle per
// gap step): 9.0 ms for the 50 k windows of the bench's 80-byte needle, 37 x the per-cell cost of the thread-per-haystack scorers.
// dp_multi_chunk never needed the needle by value - it reads one row's bytes per iteration - so the same body runs here with the rows read
// from the matcher's device blob: one THREAD per window of up to 1024 bytes (every window width: a single-chunk window is a one-chunk
// walk), the parked rows in a global slab [row][dword][thread].  Windows beyond 1024 bytes (match_greedy, src/smith_waterman/greedy.rs)
// are queued - (output position, window, haystack) entries, counters[3] - for the wave-per-haystack kernel behind this one.
// Records are written at the items' list positions (index order), the two counters by the first thread.
// ---------------------------------------------------------------------------------------------------------------
template <int SWL, bool BIAS>
__global__ __launch_bounds__(128) void k2d_dp_long(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset, const u32* __restrict__ items,
                                                   const u32* __restrict__ win, int wmode, const u32* __restrict__ n_items_ptr, const NeedleLongDev nd,
                                                   fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ scratch, u32* __restrict__ queue,
                                                   u32* __restrict__ counters) {
    __shared__ u8 cls[256];
    build_cls_table(cls);
    __syncthreads();
    const u32 n = *n_items_ptr;
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = n < capacity ? n : capacity; dev_count[1] = n; }
    const u32 nthreads = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    for (u32 q = gtid; q < n; q += nthreads) {
        if (q >= capacity) continue;
        const u32 li = items ? items[q] : q;
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        const u8* hay = bytes + s;
        u32 ws = 0, we = L;
        if (wmode == 1) {
            // the 0-typo window in its lane-free form (src/prefilter/algo/ascii.rs:6-72): first occurrence of the first needle byte, one past
            // the last occurrence of the last one, either case; the haystack passed the exact filter, so both exist (haystacks are 16-byte
            // aligned in the padded layout: whole dwords, bytes beyond L masked)
            const u32 c0 = (u32)nd.c[0] * 0x01010101u, f0 = (u32)nd.f[0] * 0x01010101u;
            const u32 cl = (u32)nd.c[nd.rows - 1] * 0x01010101u, fl = (u32)nd.f[nd.rows - 1] * 0x01010101u;
            ws = 0xFFFFFFFFu;
            we = 0;
            for (u32 p = 0; p < L; p += 4) {
                const u32 w = *(const u32*)(hay + p);
                const u32 vm = L - p >= 4 ? 0xFu : ((1u << (L - p)) - 1u);
                const u32 mf = (zero_bytes4_dp(w ^ c0) | zero_bytes4_dp(w ^ f0)) & vm;
                const u32 ml = (zero_bytes4_dp(w ^ cl) | zero_bytes4_dp(w ^ fl)) & vm;
                if (mf && ws == 0xFFFFFFFFu) ws = p + (u32)__builtin_ctz(mf);
                if (ml) we = p + 32u - (u32)__builtin_clz(ml);
            }
            if (ws == 0xFFFFFFFFu) ws = 0;
        } else if (wmode != 2) { ws = win[2 * q]; we = win[2 * q + 1]; }
        const u32 sp = ws ? ws - 1 : 0;
        const bool include_exact = sp == 0 && we == L;
        const u32 m = we - sp;
        if (m > FZB_MAX_HAYSTACK_LEN) {  // the greedy fallback: the wave-per-haystack kernel's
            u32* qe = queue + 4 * (size_t)atomicAdd(&counters[3], 1u);
            qe[0] = q; qe[1] = ws; qe[2] = we; qe[3] = li;
            continue;
        }
        u32 score = dp_multi_chunk<SWL, BIAS, NeedleLongDev>(nd, hay + sp, m, sp == 0, cls, scratch, nthreads, gtid);
        bool exact = include_exact && m == (u32)nd.nbytes;
        if (exact)
            for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
        if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = exact ? 1 : 0;
        rec.valid = 0;
        out[q] = rec;
    }
}

// The same windows with FOUR lanes each (dp_quad.h; LaunchCfg::cfm_ok, 64- or 32-lane score chunks): sixteen windows per wavefront, the needle's rows
// staged in LDS, the parked rows in the global slab (a block per window slot of the grid), requested one row ahead.
template <int SWL, bool UPPER>
__global__ __launch_bounds__(256) void k2d_dp_long_quad(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset, const u32* __restrict__ items,
                                                        const u32* __restrict__ win, int wmode, const u32* __restrict__ n_items_ptr, const NeedleLongDev nd,
                                                        fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ scratch, u32* __restrict__ queue,
                                                        u32* __restrict__ counters) {
    __shared__ CfTables tab;
    __shared__ u16 s_cf[FZB_LONG_LDS_ROWS];
    NeedleLongRows nr;
    static_cast<NeedleLongDev&>(nr) = nd;
    nr.cf = s_cf;
    cf_build_tables<UPPER, NeedleLongDev>(nd, tab);
    for (u32 r = threadIdx.x; r < (u32)nd.rows; r += blockDim.x) s_cf[r] = (u16)((u32)nd.c[r] | ((u32)nd.f[r] << 8));
    __syncthreads();
    const u32 n = *n_items_ptr;
    if (dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = n < capacity ? n : capacity; dev_count[1] = n; }
    // A workgroup takes 64 windows at a time: one thread per window finds it, the 64 are ordered by their number of chunks (a counting sort in
    // LDS) and dealt to the four wavefronts longest first - a wave lasts as long as its widest window, and sixteen windows drawn at random hold
    // the list's widest almost every time (100..200-byte haystacks: 6.7 chunk walks per wave; by quartiles 5.6).
    __shared__ uint4 s_rec[64];  // q | window start | window end (bit 31: the window is the whole haystack) | local haystack index
    __shared__ u64 s_hay[64];
    __shared__ u32 s_hist[34];
    const u32 tid = threadIdx.x;
    const u32 wslot = (tid >> 6) * 16 + ((tid >> 4) & 3u) * 4 + (tid & 3u);
    const u32 nslots = gridDim.x * 64, slot = blockIdx.x * 64 + wslot;
    const bool lane0 = ((tid >> 2) & 3u) == 0;
    for (u32 base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {  // (the same trip count for every thread of the workgroup)
        if (tid < 34) s_hist[tid] = 0;
        __syncthreads();
        u32 bin = 33, rank = 0;
        uint4 rec = make_uint4(0xFFFFFFFFu, 0, 0, 0);
        u64 hptr = 0;
        if (tid < 64) {
            const u32 q = base + tid;
            if (q < n && q < capacity) {
                const u32 li = items ? items[q] : q;
                u64 s;
                u32 L;
                haystack_span(ends, first + li, s, L);
                const u8* hay = bytes + s;
                u32 ws = 0, we = L;
                if (wmode == 1) {  // the lane-free 0-typo window (as k2d_dp_long)
                    const u32 c0 = (u32)nd.c[0] * 0x01010101u, f0 = (u32)nd.f[0] * 0x01010101u;
                    const u32 cl = (u32)nd.c[nd.rows - 1] * 0x01010101u, fl = (u32)nd.f[nd.rows - 1] * 0x01010101u;
                    ws = 0xFFFFFFFFu;
                    we = 0;
                    for (u32 p = 0; p < L; p += 4) {
                        const u32 w = *(const u32*)(hay + p);
                        const u32 vm = L - p >= 4 ? 0xFu : ((1u << (L - p)) - 1u);
                        const u32 mf = (zero_bytes4_dp(w ^ c0) | zero_bytes4_dp(w ^ f0)) & vm;
                        const u32 ml = (zero_bytes4_dp(w ^ cl) | zero_bytes4_dp(w ^ fl)) & vm;
                        if (mf && ws == 0xFFFFFFFFu) ws = p + (u32)__builtin_ctz(mf);
                        if (ml) we = p + 32u - (u32)__builtin_clz(ml);
                    }
                    if (ws == 0xFFFFFFFFu) ws = 0;
                } else if (wmode != 2) { ws = win[2 * q]; we = win[2 * q + 1]; }
                const u32 sp = ws ? ws - 1 : 0;
                const u32 m = we - sp;
                if (m > FZB_MAX_HAYSTACK_LEN) {  // the greedy fallback: the wave-per-haystack kernel's
                    u32* qe = queue + 4 * (size_t)atomicAdd(&counters[3], 1u);
                    qe[0] = q; qe[1] = ws; qe[2] = we; qe[3] = li;
                } else {
                    rec = make_uint4(q, ws, we | ((sp == 0 && we == L) ? 0x80000000u : 0u), li);
                    hptr = (u64)(uintptr_t)hay;
                    bin = 32 - (m + SWL - 1) / SWL;  // widest first (m = 0: bin 32)
                }
            }
            rank = atomicAdd(&s_hist[bin], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            u32 off = 0;
            for (u32 k = 0; k < bin; k++) off += s_hist[k];
            s_rec[off + rank] = rec;
            s_hay[off + rank] = hptr;
        }
        __syncthreads();
        const uint4 w = s_rec[wslot];
        if (w.x != 0xFFFFFFFFu) {
            const u8* hay = (const u8*)(uintptr_t)s_hay[wslot];
            const u32 sp = w.y ? w.y - 1 : 0;
            const bool include_exact = (w.z >> 31) != 0;
            const u32 m = (w.z & 0x7FFFFFFFu) - sp;
            constexpr int MAXC = SWL == 32 ? 8 : 4;  // windows of up to 256 bytes: row by row, in registers
            u32 score;
            if (m <= (u32)MAXC * SWL) score = dp_quad_rows<SWL, UPPER, MAXC>(nr, hay + sp, m, sp == 0, tab);
            else score = dp_quad_window<SWL, UPPER, true>(nr, hay + sp, m, sp == 0, tab, scratch + slot, nslots);
            if (lane0) {
                bool exact = include_exact && m == (u32)nd.nbytes;
                if (exact)
                    for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
                if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
                fzb_match_rec r;
                r.index = index_offset + w.w;
                r.score = (u16)score;
                r.exact = exact ? 1 : 0;
                r.valid = 0;
                out[w.x] = r;
            }
        }
        __syncthreads();  // (s_rec is rewritten by the next 64)
    }
}

// dwords of the slab per 256-thread WORKGROUP of k2d_dp_long_quad ([row][QuadPark::WORDS][64 windows]); 0 = that lane width has no quad form
size_t fzb_dp_long_quad_words_per_block(const NeedleLongDev& nd, int sw_lanes) {
    return (sw_lanes == 64 || sw_lanes == 32) ? (size_t)nd.rows * (size_t)(sw_lanes / 4 + 1) * 64 : 0;
}
void fzb_launch_dp_long_quad(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd, int sw_lanes,
                             int upper, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* scratch, u32* queue, u32* counters, int grid, hipStream_t st) {
#define FZB_K2Q(SWL, U) hipLaunchKernelGGL((k2d_dp_long_quad<SWL, U>), dim3(grid), dim3(256), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, wmode, n_items_ptr, nd, out, capacity, dev_count, scratch, queue, counters)
    if (sw_lanes == 64) { if (upper) FZB_K2Q(64, true); else FZB_K2Q(64, false); }
    else { if (upper) FZB_K2Q(32, true); else FZB_K2Q(32, false); }
#undef FZB_K2Q
}

// dwords of the parked-row slab per THREAD of k2d_dp_long (dp_multi_chunk's layout: [row][SWL / 2 dwords][thread])
size_t fzb_dp_long_scratch_words_per_thread(const NeedleLongDev& nd, int sw_lanes) { return (size_t)nd.rows * (size_t)(sw_lanes / 2); }

void fzb_launch_dp_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd, int sw_lanes,
                        int bias_ok, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* scratch, u32* queue, u32* counters, int grid, hipStream_t st) {
#define FZB_K2L(SWL, B) hipLaunchKernelGGL((k2d_dp_long<SWL, B>), dim3(grid), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, wmode, n_items_ptr, nd, out, capacity, dev_count, scratch, queue, counters)
#define FZB_K2L_B(SWL) do { if (bias_ok) FZB_K2L(SWL, true); else FZB_K2L(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2L_B(64); break;
        case 32: FZB_K2L_B(32); break;
        case 16: FZB_K2L_B(16); break;
        default: FZB_K2L_B(8); break;
    }
#undef FZB_K2L_B
#undef FZB_K2L
}

// Where a thread of the dp_cfm.h kernels parks its rows between two chunks: its workgroup's LDS when the launch gave it room (`park_dw` =
// dwords per parked row, fzb_park_lds_dwords: needles of a few rows), else its column of the global slab.  In LDS a row's round trip costs
// an LDS access instead of an L2 one - with one or two waves per SIMD and a dependent load per needle row and chunk that latency is the
// kernel (paths-shaped list: the multi-chunk slice alone 44.6 us for 27 k windows of two or three chunks).
struct ParkAt { u32* base; u32 sstride, sidx, rpitch; };
__device__ __forceinline__ ParkAt park_at(u32* scratch, u32 nthreads, u32 gtid, u32 park_dw, u32 nw) {
    extern __shared__ __attribute__((aligned(16))) u32 s_park[];
    if (park_dw) return ParkAt{s_park, blockDim.x, threadIdx.x, park_dw * blockDim.x};
    return ParkAt{scratch, nthreads, gtid, nw * nthreads};
}

// the same list through dp_cfm.h's form (LaunchCfg::cfm_ok); bonuses from the LDS tables as in the other dp_cf.h kernels
template <int SWL, bool UPPER>
__global__ __launch_bounds__(128, 2) void k2d_dp_multi_t(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                                      const u32* __restrict__ list, const u32* __restrict__ n_list_ptr, const NeedleDev nd,
                                                      fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ scratch, u32 park_dw) {
    __shared__ CfTables tab;
    cf_build_tables<UPPER>(nd, tab);
    __syncthreads();
    const u32 nlist = *n_list_ptr;
    const u32 nthreads = gridDim.x * blockDim.x, gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const ParkAt pk = park_at(scratch, nthreads, gtid, park_dw, SWL / 2);
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
        u32 score = dp_multi_chunk_t<SWL, UPPER>(nd, hay + sp, m, sp == 0, tab, pk.base, pk.sstride, pk.sidx, pk.rpitch);
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

// the multi-chunk windows as k2w_classify's four lists by the width of the last chunk's tail (lists 3-6 of `lists`, counts in counters[12..15]):
// one persistent walk over the concatenation, widest class first (a slot's later items are its cheaper ones); a wave computes the last chunk
// with the class of its first lane - the widest among its 64 (only the three waves that straddle a list boundary compute more than needed)
template <int SWL, bool UPPER>
__device__ __forceinline__ void dp_multi_tc_body(const CfTables& tab, u32 vblock, u32 vgrid, u32 index_offset,
                                                 const u32* __restrict__ items, const uint4* __restrict__ meta, const u32* __restrict__ lists, u32 list_stride,
                                                 const u32* __restrict__ counts, const NeedleDev& nd, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ scratch,
                                                 u32 park_dw) {
    // positions [0, e3) class 3 (the whole last chunk), [e3, e2) class 2, [e2, e1) class 1, [e1, e0) class 0
    const u32 e3 = __builtin_amdgcn_readfirstlane(counts[3]), e2 = e3 + __builtin_amdgcn_readfirstlane(counts[2]), e1 = e2 + __builtin_amdgcn_readfirstlane(counts[1]),
              e0 = e1 + __builtin_amdgcn_readfirstlane(counts[0]);
    const u32 nthreads = vgrid * blockDim.x, gtid = vblock * blockDim.x + threadIdx.x;
    const ParkAt pk = park_at(scratch, nthreads, gtid, park_dw, SWL / 2);
    for (u32 q = gtid; q < e0; q += nthreads) {
        const u32 cls = q < e3 ? 3u : q < e2 ? 2u : q < e1 ? 1u : 0u;
        const u32 base = cls == 3 ? 0u : cls == 2 ? e3 : cls == 1 ? e2 : e1;
        const u32 j = lists[(size_t)(3 + cls) * list_stride + (q - base)];
        if (j >= capacity) continue;
        const u32 li = items ? items[j] : j;
        const uint4 w = meta[j];  // the classifier's record: window, "whole haystack" flag, where the bytes are
        const u8* hay = (const u8*)(uintptr_t)((u64)w.z | ((u64)w.w << 32));
        const u32 sp = w.x ? w.x - 1 : 0;
        const bool include_exact = (w.y >> 31) != 0;
        const u32 m = (w.y & 0x7FFFFFFFu) - sp;
        const u32 wcls = __builtin_amdgcn_readfirstlane(cls);  // lanes are in position order: the first active lane holds the widest class
        u32 score = dp_multi_chunk_tc<SWL, UPPER>(nd, hay + sp, m, sp == 0, tab, pk.base, pk.sstride, pk.sidx, pk.rpitch, wcls);
        bool exact = include_exact && m == (u32)nd.nbytes;
        if (exact)
            for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
        if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = exact ? 1 : 0;
        rec.valid = 0;
        out[j] = rec;
    }
}

// The same four lists with FOUR lanes per window (dp_quad.h): sixteen windows per wavefront, dp_cfm.h's arithmetic, the parked rows in LDS
// ([row][word][window of the workgroup]).  `singles` (counters[8..10], or null): the three single-chunk lists behind them, widest class first - on
// a short list EVERY window takes four lanes (one chunk in registers, all of its lanes computed) and every workgroup of the launch walks the one
// concatenation: the thread-per-window class bodies would otherwise queue up behind this slice at two waves per SIMD.
template <int SWL, bool UPPER>
__device__ __forceinline__ void dp_quad_tc_body(const CfTables& tab, u32 vblock, u32 vgrid, u32 index_offset, const u32* __restrict__ items, const uint4* __restrict__ meta,
                                                const u32* __restrict__ lists, u32 list_stride, const u32* __restrict__ counts, const u32* __restrict__ singles,
                                                const NeedleDev& nd, fzb_match_rec* __restrict__ out, u32 capacity) {
    extern __shared__ __attribute__((aligned(16))) u32 s_park[];
    if constexpr (SWL == 64 || SWL == 32) {
        const u32 e3 = __builtin_amdgcn_readfirstlane(counts[3]), e2 = e3 + __builtin_amdgcn_readfirstlane(counts[2]), e1 = e2 + __builtin_amdgcn_readfirstlane(counts[1]),
                  e0 = e1 + __builtin_amdgcn_readfirstlane(counts[0]);
        // window of the wave: DPP row x its four windows (row-lanes w, w + 4, w + 8, w + 12)
        const u32 wpb = blockDim.x / 4, wslot = (threadIdx.x >> 6) * 16 + ((threadIdx.x >> 4) & 3u) * 4 + (threadIdx.x & 3u);
        const u32 ngroups = vgrid * wpb, gid = vblock * wpb + wslot;
        u32* const park = s_park + wslot;
        const u32 s2 = e0 + (singles ? __builtin_amdgcn_readfirstlane(singles[2]) : 0u), s1 = s2 + (singles ? __builtin_amdgcn_readfirstlane(singles[1]) : 0u),
                  s0 = s1 + (singles ? __builtin_amdgcn_readfirstlane(singles[0]) : 0u);
        for (u32 q = gid; q < s0; q += ngroups) {
            u32 l, base;  // list, first position of it in the concatenation
            if (q < e0) {
                const u32 cls = q < e3 ? 3u : q < e2 ? 2u : q < e1 ? 1u : 0u;
                l = 3 + cls;
                base = cls == 3 ? 0u : cls == 2 ? e3 : cls == 1 ? e2 : e1;
            } else {
                l = q < s2 ? 2u : q < s1 ? 1u : 0u;
                base = l == 2 ? e0 : l == 1 ? s2 : s1;
            }
            const u32 j = lists[(size_t)l * list_stride + (q - base)];
            if (j >= capacity) continue;
            const u32 li = items ? items[j] : j;
            const uint4 w = meta[j];
            const u8* hay = (const u8*)(uintptr_t)((u64)w.z | ((u64)w.w << 32));
            const u32 sp = w.x ? w.x - 1 : 0;
            const bool include_exact = (w.y >> 31) != 0;
            const u32 m = (w.y & 0x7FFFFFFFu) - sp;
            constexpr int MAXC = SWL == 32 ? 4 : 3;  // windows of up to 128 / 192 bytes: row by row, in registers
            u32 score;
            if (m <= (u32)MAXC * SWL) score = dp_quad_rows<SWL, UPPER, MAXC>(nd, hay + sp, m, sp == 0, tab);
            else score = dp_quad_window<SWL, UPPER, false>(nd, hay + sp, m, sp == 0, tab, park, 0u);
            if (((threadIdx.x >> 2) & 3u) == 0) {
                bool exact = include_exact && m == (u32)nd.nbytes;
                if (exact)
                    for (u32 k = 0; k < m; k++) exact = exact && hay[sp + k] == nd.raw[k];
                if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
                fzb_match_rec rec;
                rec.index = index_offset + li;
                rec.score = (u16)score;
                rec.exact = exact ? 1 : 0;
                rec.valid = 0;
                out[j] = rec;
            }
        }
    }
}

// The three single-chunk classes and the multi-chunk tail classes in ONE launch - the grid is cut into four slices, a workgroup runs the body of
// its slice (widest work first).  On a list of a million items every one of four separate launches is a single round of single items, so their
// latencies (and launch boundaries) add up along the stream (paths-shaped list 128 -> 102 us); here they run side by side.  Registers follow the
// widest body (two waves per SIMD); on the 12.5 M-item shard, where everything is bound by instruction issue, one launch and four are equal.
template <int SWL, bool UPPER>
__global__ __launch_bounds__(128, 2) void k2_classes_all(u32 index_offset, const u32* __restrict__ items, const uint4* __restrict__ meta, const u32* __restrict__ lists,
                                                      u32 list_stride, const u32* __restrict__ counters, const NeedleDev nd, fzb_match_rec* __restrict__ out, u32 capacity,
                                                      u32* __restrict__ scratch, u32 gm, u32 gc, u32 park_dw, u32 coop_below, u32 quad_all_below) {
    __shared__ CfTables tab;
    cf_build_tables<UPPER>(nd, tab);
    __syncthreads();
    u32 b = blockIdx.x;
    // fewer multi-chunk windows than `coop_below` (0: the needle's parked rows do not fit, or another lane width): four lanes per window; fewer
    // windows of any kind than `quad_all_below`: four lanes for all of them, the whole grid on one list
    const u32 total = __builtin_amdgcn_readfirstlane(counters[12] + counters[13] + counters[14] + counters[15]);
    const u32 total_all = total + __builtin_amdgcn_readfirstlane(counters[8] + counters[9] + counters[10]);
    // which list this workgroup walks (0: the multi-chunk ones, 1-3: the single-chunk classes, widest first) and as which of its workgroups.
    // Thread form: the grid in slices, widest work first.  Four lanes for the multi-chunk slice: its workgroups are short-lived now and the class
    // bodies' are the long ones, so the two kinds are dealt out in turns (4 : 3, the classes in turns too) - both are among the first workgroups the
    // dispatcher places and finish side by side, instead of the classes queueing behind a slice that holds every slot for its ten microseconds
    // (1.4 M paths 90.9 -> 88.4 us, 8..128-byte list of 2 M items 109.7 -> 102.8).
    const bool quad = total < coop_below, all = quad && total_all < quad_all_below;
    u32 kind, vb;
    if (all) kind = 0, vb = b;
    else if (quad && gm == 4 * gc) {
        const u32 grp = b / 7, pos = b % 7;
        if (pos & 1u) {
            const u32 cidx = 3 * grp + (pos >> 1);
            kind = 1 + cidx % 3, vb = cidx / 3;
        } else kind = 0, vb = 4 * grp + (pos >> 1);
    } else if (b < gm) kind = 0, vb = b;
    else kind = 1 + (b - gm) / gc, vb = (b - gm) % gc;
    if (kind == 0) {
        if (quad) dp_quad_tc_body<SWL, UPPER>(tab, vb, all ? gridDim.x : gm, index_offset, items, meta, lists, list_stride, &counters[12], all ? &counters[8] : nullptr, nd, out, capacity);
        else dp_multi_tc_body<SWL, UPPER>(tab, vb, gm, index_offset, items, meta, lists, list_stride, &counters[12], nd, out, capacity, scratch, park_dw);
    } else if (kind == 1) dp_class_body<SWL, UPPER, SWL / 2>(tab, vb, gc, index_offset, items, meta, lists + 2 * (size_t)list_stride, &counters[10], nd, out);
    else if (kind == 2) dp_class_body<SWL, UPPER, 3 * SWL / 8>(tab, vb, gc, index_offset, items, meta, lists + (size_t)list_stride, &counters[9], nd, out);
    else dp_class_body<SWL, UPPER, SWL / 4>(tab, vb, gc, index_offset, items, meta, lists, &counters[8], nd, out);
}

// dwords per parked row when the dp_cfm.h kernels (128 threads) keep a needle's parked rows in LDS, 0 when they go to the global slab: a row
// is the top half of a chunk's final vector (SWL/4 dwords, or half of that packed to bytes when the scores fit a byte) and one word of
// flags; the budget (FZB_PARK_LDS_KB, default 37: four workgroups per CU beside their tables) holds 8 rows of byte scores at 64 lanes.
static u32 fzb_park_lds_dwords(const NeedleDev& nd, int sw_lanes) {
    const u32 ht = (u32)sw_lanes / 4;
    const u32 dw = (nd.lane_mask == 0xFF ? ht / 2 : ht) + 1;
    const size_t bytes = (size_t)nd.rows * dw * 128 * 4;
    return bytes <= (size_t)fzb_knobs().park_lds_kb * 1024 ? dw : 0u;
}

// gm workgroups for the multi-chunk lists (the scratch slab is sized for them: ensure_dp_scratch), gc for each single-chunk class
void fzb_launch_classes_all(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* lists, u32 list_stride, const u32* counters,
                            const NeedleDev& nd, int sw_lanes, fzb_match_rec* out, u32 capacity, u32* scratch, int gm, int gc, hipStream_t st) {
    bool upper = false;
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    const u32 park_dw = fzb_park_lds_dwords(nd, sw_lanes);
    // four lanes per window for the multi-chunk slice (dp_quad.h): 64- and 32-lane backends; taken ON THE DEVICE below `coop_below` queued windows.
    // A thread-per-window wave walks ~ 1 000 instructions per (row, chunk) whatever the queue's length - 44 us for a window of three chunks and
    // five rows even when the queue holds one wavefront's worth; four lanes per window make that chain a quarter as long at the same total, so
    // the form pays while the thread form would leave SIMDs with at most one wavefront.  Measured on 8..128-byte lists (windows = 2.1 % of the
    // items; profiles/r06_quad.txt): 10.7 k windows 81.6 -> 61.9 us per step, 21 k 89.2 -> 69.9, 43 k 110.9 -> 104.0, 64 k 128.5 -> 139.5 (the
    // thread form again); paths-shaped 1.4 M items (27 k windows) 99.8 -> 90.5.  Windows of up to 3 (4) chunks run row by row in registers; wider
    // ones park their rows in LDS, which is what the 48 KB bound is for (a needle too long for it keeps the thread form for every window).
    const size_t coop_lds = (sw_lanes == 64 || sw_lanes == 32) ? (size_t)nd.rows * (sw_lanes / 4 + 1) * (128 / 4) * 4 : 0;  // [row][QuadPark::WORDS][window of the workgroup]
    const u32 coop_below = (coop_lds != 0 && coop_lds <= 48 * 1024 && fzb_knobs().coop_below != 0) ? (fzb_knobs().coop_below > 0 ? (u32)fzb_knobs().coop_below : (u32)gm * 48u) : 0u;
    // ... and below 32 768 windows of ANY kind every window takes four lanes (paths-shaped lists of 100 k / 300 k items, 7.9 k / 23.9 k windows:
    // 46.4 -> 38.7 / 50.0 -> 43.7 us; 8..128-byte lists: 25 k windows 57.7 -> 54.2, 50 k 66.3 -> 69.3, 112 k (1.4 M paths) 90.9 -> 92.1)
    const u32 quad_all_below = std::min<u32>(coop_below, (u32)gm * 32u);
    const size_t dyn_lds = std::max((size_t)nd.rows * park_dw * 128 * 4, coop_below ? coop_lds : (size_t)0);
#define FZB_K2A(SWL, U) hipLaunchKernelGGL((k2_classes_all<SWL, U>), dim3(gm + 3 * gc), dim3(128), dyn_lds, st, index_offset, items, (const uint4*)win, lists, list_stride, counters, nd, out, capacity, scratch, (u32)gm, (u32)gc, park_dw, coop_below, quad_all_below)
#define FZB_K2A_U(SWL) do { if (upper) FZB_K2A(SWL, true); else FZB_K2A(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2A_U(64); break;
        case 32: FZB_K2A_U(32); break;
        case 16: FZB_K2A_U(16); break;
        default: FZB_K2A_U(8); break;
    }
}

// mode: 0 = unbiased scan, 1 = dp_body.h's biased scan (bias_ok), 2 = dp_cfm.h (cfm_ok)
void fzb_launch_dp_multi(const CorpusDev& c, u64 first, u32 index_offset, const u32* list, const u32* n_list_ptr, const NeedleDev& nd, int sw_lanes, int mode,
                         fzb_match_rec* out, u32 capacity, u32* scratch, int grid, hipStream_t st) {
    const int bias_ok = mode >= 1;
    if (mode == 2) {
        bool upper = false;
        for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
        const u32 park_dw = fzb_park_lds_dwords(nd, sw_lanes);
#define FZB_K2T(SWL, U, ET) hipLaunchKernelGGL((k2d_dp_multi_t<SWL, U>), dim3(grid), dim3(128), (size_t)nd.rows * park_dw * 128 * 4, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, list, n_list_ptr, nd, out, capacity, scratch, park_dw)
#define FZB_K2T_ET(SWL, U) FZB_K2T(SWL, U, u32)
#define FZB_K2T_U(SWL) do { if (upper) FZB_K2T_ET(SWL, true); else FZB_K2T_ET(SWL, false); } while (0)
        switch (sw_lanes) {
            case 64: FZB_K2T_U(64); break;
            case 32: FZB_K2T_U(32); break;
            case 16: FZB_K2T_U(16); break;
            default: FZB_K2T_U(8); break;
        }
        return;
    }
#define FZB_K2D(SWL, B, ET) hipLaunchKernelGGL((k2d_dp_multi<SWL, B>), dim3(grid), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, list, n_list_ptr, nd, out, capacity, scratch)
#define FZB_K2D_ET(SWL, B) FZB_K2D(SWL, B, u32)
#define FZB_K2D_B(SWL) do { if (bias_ok) FZB_K2D_ET(SWL, true); else FZB_K2D_ET(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2D_B(64); break;
        case 32: FZB_K2D_B(32); break;
        case 16: FZB_K2D_B(16); break;
        default: FZB_K2D_B(8); break;
    }
}

bool fzb_dp_short_applies(const CorpusDev& c, int sw_lanes, int mode) {
    return mode == 2 && (sw_lanes == 64 || sw_lanes == 32) && c.max_len != 0 && c.max_len <= (u32)sw_lanes / 2;
}

void fzb_launch_dp(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                   int sw_lanes, int mode, int wmode, int pad_ok, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, int num_cus, hipStream_t st,
                   const RejectOut* rejects) {
    bool upper = false;  // an uppercase letter among the needle bytes as they are compared
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    if (fzb_dp_short_applies(c, sw_lanes, mode)) {
        const RejectOut rj = rejects ? *rejects : RejectOut{};
        u32* kept_out = rejects ? &counters[1] : nullptr;
        // every haystack fits half a chunk (and the two prefetched vectors): the short-haystack kernel
#define FZB_K2S(SWL, U, ET)                                                                                                             \
    do {                                                                                                                                \
        static int per_cu = 0;                                                                                                          \
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2b_dp_short<SWL, U>, 128, 0) != hipSuccess || per_cu < 1)) per_cu = 4; \
        hipLaunchKernelGGL((k2b_dp_short<SWL, U>), dim3(num_cus * per_cu), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, rj, kept_out, c.uniform_len); \
    } while (0)
#define FZB_K2S_ET(SWL, U) FZB_K2S(SWL, U, u32)
#define FZB_K2S_U(SWL) do { if (upper) FZB_K2S_ET(SWL, true); else FZB_K2S_ET(SWL, false); } while (0)
        if (sw_lanes == 64) FZB_K2S_U(64); else FZB_K2S_U(32);
        return;
    }
    // the kernel is persistent: launch exactly the workgroups that are resident at once
#define FZB_K2B(SWL, B, U, ET)                                                                                                          \
    do {                                                                                                                                \
        static int per_cu = 0;                                                                                                          \
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2b_dp<SWL, B, U>, 128, 0) != hipSuccess || per_cu < 1)) per_cu = 4; \
        hipLaunchKernelGGL((k2b_dp<SWL, B, U>), dim3(num_cus * per_cu), dim3(128), 0, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, n_items_ptr, nd, wmode, pad_ok, out, capacity, dev_count, overflow, qcap, counters); \
    } while (0)
#define FZB_K2B_ET(SWL, B, U) FZB_K2B(SWL, B, U, u32)
#define FZB_K2B_U(SWL, B) do { if (upper) FZB_K2B_ET(SWL, B, true); else FZB_K2B_ET(SWL, B, false); } while (0)
    // (mode 2 = dp_cf.h's preconditions hold: lists that qualify took k2b_dp_short above, every other one is scored by the classified path - the
    // per-wave kernel then only runs under FZB_NO_DP_CLASSES, in its biased first form, which those preconditions include)
#define FZB_K2B_B(SWL) do { if (mode >= 1) FZB_K2B_U(SWL, 1); else FZB_K2B_U(SWL, 0); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2B_B(64); break;
        case 32: FZB_K2B_B(32); break;
        case 16: FZB_K2B_B(16); break;
        default: FZB_K2B_B(8); break;
    }
}
