// Internal structures shared by the host side (host.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define FZB_MAX_ROWS 63        // needle rows of the by-value NeedleDev (bytes on the ASCII path, scalars on the unicode path); longer needles: NeedleLongDev
#define FZB_MAX_NEEDLE_BYTES 64
#define FZB_MAX_HAYSTACK_LEN 1024  // reference: src/smith_waterman/algo/mod.rs:18 (beyond this the greedy fallback scores)
#define FZB_UNICODE_FWD_CAP 4096u  // windows beyond four chunks the thread-per-haystack unicode scorer hands on per query (and the room the queue's back keeps for them)
#define FZB_TILE 1024          // haystacks per filter tile (one bitmap group + one count)

// Needle + scoring constants, passed BY VALUE as a kernel argument (wave-uniform -> SGPR loads).
// Mirrors what `Prefilter::new` / `SmithWaterman::new` precompute (src/prefilter/algo/mod.rs:30-42,
// src/smith_waterman/algo/mod.rs:21-42) and the constants block of `score_haystack`
// (src/smith_waterman/algo/ascii.rs:33-46).
struct NeedleDev {
    int32_t rows;          // DP rows: needle bytes (ASCII path) or needle scalars (unicode path)
    int32_t nbytes;        // needle byte length (exact-match compare)
    int32_t max_typos;     // -1 = None
    int32_t min_haystack_len;  // chars - max_typos (src/matcher/algo.rs:62-65)
    int32_t unicode;       // 1 = unicode path
    int32_t lane_mask;     // 0xFF (u8 score class) or 0xFFFF (u16 class): lane type of the emulated CPU backend
    // scoring constants, already combined as the reference combines them
    u16 match_plus_mismatch;  // match_score.saturating_add(mismatch_penalty)
    u16 mismatch;
    u16 gex;               // gap_extend_penalty
    u16 gopm;              // gap_open_penalty.saturating_sub(gap_extend_penalty)
    u16 prefix, capitalization, matching_case, exact_bonus, delimiter;
    u16 match_score, gap_open;  // raw values (greedy fallback, src/smith_waterman/greedy.rs)
    u16 _pad;
    alignas(4) u8 raw[FZB_MAX_NEEDLE_BYTES];  // needle bytes as given
    alignas(4) u8 c[FZB_MAX_NEEDLE_BYTES];    // ASCII rows: byte            (case_needle, src/prefilter/mod.rs:49-65)
    alignas(4) u8 f[FZB_MAX_NEEDLE_BYTES];    //            its case flip
    u8 uc[FZB_MAX_ROWS + 1][4];    // unicode rows: scalar bytes  (case_needle_unicode, src/prefilter/mod.rs:71-96)
    u8 uf[FZB_MAX_ROWS + 1][4];    //               flipped scalar bytes
    u8 ulen[FZB_MAX_ROWS + 1];     //               UTF-8 length
    static constexpr bool kLong = false;
};

// A needle beyond NeedleDev's by-value arrays (> 64 bytes or > 63 rows; the reference takes needles up to `Scoring::max_needle_len()`,
// src/lib.rs:480-503 - 10 922 rows with the default scoring): the same scalars, the arrays in device memory.  Member NAMES are those of
// NeedleDev: the kernels that serve such needles (lane-exact prefilter kernels_window.hip, wave-per-haystack scorer
// kernels_generic.hip, literal modes kernels_literal.hip) are templates over the needle type.
struct NeedleLongDev {
    int32_t rows, nbytes, max_typos, min_haystack_len, unicode, lane_mask;
    u16 match_plus_mismatch, mismatch, gex, gopm;
    u16 prefix, capitalization, matching_case, exact_bonus, delimiter;
    u16 match_score, gap_open, _pad;
    const u8* raw;          // [nbytes]
    const u8* c;            // [nbytes] ASCII rows
    const u8* f;            // [nbytes]
    const u8 (*uc)[4];      // [rows] unicode rows
    const u8 (*uf)[4];      // [rows]
    const u8* ulen;         // [rows]
    static constexpr bool kLong = true;
};

// One haystack list resident in HBM.  Layout ("padded-16"): every haystack starts on a 16-byte boundary of
// `bytes`, gaps are zero, `ends[i]` is the exclusive byte end of haystack i inside `bytes`;
// start(i) = i ? roundup16(ends[i-1]) : 0.  >= 80 zero bytes follow the last haystack.
struct CorpusDev {
    const u8* bytes;
    const void* ends;  // u32[n] or u64[n]
    u64 n;
    u64 total_bytes;   // padded size
    int ends_u64;
    u32 max_len;       // longest haystack in bytes, 0 = unknown (lets the pipeline skip the multi-chunk scorer launch)
    u32 uniform_len;   // every haystack has exactly this many bytes (0 = not known): start(i) = i * roundup16(len), no end offsets read
    // The streaming filter's VIEW of a ragged list (fzb_corpus_upload builds it for lists whose haystacks are 33..256 bytes; nullptr
    // otherwise; every other stage reads the canonical layout).  What bounds the thread-per-haystack filter on such a list is the access
    // pattern itself - 64 lanes x 16 bytes from 64 different lines per load instruction: 215 us for the 0.9 GB of the C4 shard with the
    // automaton switched off (round 3's measurement) - so the view stores the bytes the way the lanes read them:
    //   * every 1024-haystack tile sorted by DESCENDING length (round 5; rounds 3-4: by number of 16-byte vectors) (vperm[p] = position inside its tile the haystack at
    //     sorted position p came from, vlen[p] = its length);
    //   * every GROUP of 64 consecutive sorted haystacks (one wave's worth) interleaved by vector: vector v of the group's member j lives
    //     at vbytes + 16 * vgofs[group] + 1024 * v + 16 * j, for v < vgnv[group] = the group's longest member (zero vectors behind a
    //     shorter one) - so a wave's load of "vector v of my haystack" is ONE fully coalesced 1 KiB access, like a copy kernel's.
    const u8* vbytes;
    const u32* vgofs;  // per group: offset of its block in 16-byte units
    const u8* vgnv;    // per group: vectors per member (<= 16) | (bytes per member in the LAST vector's row / 4 - 1) << 5: rows 0 .. nv-2 are 1 KiB (16 B per
                       // member), the last row holds 4 / 8 / 12 / 16 bytes per member - as narrow as the group's longest member's tail allows (round 5)
    const u16* vlen;   // per sorted haystack
    const u16* vperm;  // per sorted haystack
    u32 view_nv;       // the view's widest member in 16-byte vectors (<= 16): read from the lengths when the view is built, not a caller's hint
    // OUTLIERS: the (few) haystacks beyond 256 bytes are not in the view (vlen = 0xFFFF, no vectors); the filter decides them from the
    // canonical layout in a small follow-up launch over this list of their indices (k1_cdfa_outliers).  Without it one 300-byte path
    // would cost a million-item list its view.
    const u32* vlong;
    u32 n_long;
};

struct fzb_match_rec {  // == fzb_match; `_pad` carries the valid flag between kernels (0 in final output)
    u32 index;
    u16 score;
    u8 exact;
    u8 valid;
};

// Per-call device workspace (owned by the matcher, grown on demand)
struct Workspace {
    u64* bitmap;        // count/64 words: filter decisions
    u32* tile_counts;   // ntiles
    u32* surv_idx;      // local haystack index of survivor j
    u32* win;           // 2 * survivors: (start, end) windows from the lane-exact prefilter; start=0xFFFFFFFF => rejected
    u32* overflow;      // queue of (output position, window start, window end, haystack): multi-chunk windows from the front, > 1024-byte windows from the back
    u32* dp_scratch;    // multi-chunk DP: parked row/gap vectors, [row][dword][thread]
    size_t dp_scratch_words;
    fzb_match_rec* sort_tmp;  // device radix sort: ping-pong buffer + digit-major tile histogram
    u32* sort_hist;
    size_t sort_cap;
    u64* bitmap2;       // second-level keep bits (after the lane-exact prefilter)
    u32* tile_counts2;
    u32* items2;        // local haystack index of kept survivor
    u32* win2;          // its window
    u32* counters;      // [0]=filter survivors [1]=kept by the lane-exact prefilter [2]=output base of the NEXT chunk [3]=multi-chunk queue length [4]=greedy (>1024 B) queue length
                        // [5]=marginal survivors (LCS == need) [6]=of those, rejected by the lane-exact decision
    u64* bitmap_m;      // typo fast path: "accepted with nothing to spare" bits, their tile counts, the list of those haystacks,
    u32* tile_counts_m; //   the reject bits / per-tile counts / prefix the decide pass produces
    u32* marg_list;
    u64* reject_bits;
    u32* tile_rejects;
    u32* rej_prefix;
    size_t cap_marg;    // capacity (in haystacks) of the six arrays above (0 = not allocated)
    u32* cls_win;       // classified scoring (corpora with longer haystacks): per survivor a 16-byte record (window start, window end |
                        //   "the window is the whole haystack" flag, 64-bit address of the haystack's bytes), three class lists
    u32* cls_lists;     //   [7][cap]; class counts in counters[8..10], multi-chunk tail classes in counters[12..15]
    size_t cap_cls;
    u64* table;         // 256 x u64 filter table (device)
    u8* dfa;            // (rows + 1) x 256 next-state table of the ordered-subsequence DFA (device)
    u8* uni_dfa;        // unicode path, 0 typos: states x 256 table of the exact prefilter's byte-level DFA (device)
    u8* lcs_dfa;        // typo configurations: states x 256 table of the LCS automaton (device)
    u8* cdfa;           // class-composite form of the matcher's streaming automaton: [256 byte -> class][states x K^G next state] (device)
    size_t cap_items;   // capacity (in haystacks) of the first-level arrays
    size_t cap_level2;  // capacity of the second-level arrays (0 = not allocated)
    bool tables_stale;  // the matcher's needle / config changed since `table` and `dfa` were uploaded
    // the five tables above are ONE device allocation (fixed offsets) behind ONE pinned staging buffer: a needle change is one asynchronous
    // copy on the query's stream instead of up to five synchronous ones (45 -> ~ 8 us of a re-query after fzb_matcher_set_pattern)
    u8* tables_blob;
    u8* tables_host;
    hipEvent_t tables_ev;   // recorded behind the last copy out of `tables_host` (waited for before it is refilled)
    bool tables_ev_pending;
    // matched-indices path (fzb_match_list_indices)
    u32* trace_cells;   // per-wave score / match matrices of the traced scorer
    size_t trace_cells_words;
};

struct LaunchCfg {
    int pf_lanes, sw_lanes;
    int filter_mode;    // 0 = none (all pass), 1 = ordered subsequence (exact for ASCII 0 typos), 2 = LCS >= rows-k (superset)
    int filter_exact;   // 1 if the filter decision is exactly the reference's accept decision
    int window_mode;    // 0 = from the lane-exact prefilter kernel, 1 = inline first/last occurrence (ASCII 0 typos), 2 = full haystack
    int bias_ok;        // DP gap propagation may run in the biased domain (no u16 overflow possible)
    int pad_ok;         // needle has no NUL byte: zero-padding lanes can never match (enables the padded-half DP form)
    int cfu_ok;  // dp_unicode.h's biased-throughout form: bias_ok and 2 * gap_extend <= mismatch_penalty
    int cf_ok;
    int cfm_ok;  // dp_cfm.h preconditions (multi-chunk windows in the biased domain)          // single-chunk scorer in its second form (dp_cf.h): pad_ok, bias_ok and 2 * gap_extend <= mismatch_penalty
    int num_cus;
    u32 dead_byte;      // a byte value no needle row can match (used to neutralise bytes past a haystack's end in the DFA filter)
};

// Index-ordered record runs of contiguous shards, all readable from the current device (k_concat_runs); count[g] points at the pair
// fzb_match_list_device writes (records written, matches found), cap[g] is the run's buffer size in records
#define FZB_MAX_RUNS 64
struct RunSet {
    const fzb_match_rec* run[FZB_MAX_RUNS];
    const u32* count[FZB_MAX_RUNS];
    u32 cap[FZB_MAX_RUNS];
    int n;
};

// Rejections of the decide pass (typo configurations on the short-haystack path): one bit per haystack of the range, a count per
// 1024-haystack tile and its exclusive prefix (filled only when anything was rejected), the total in counters[6]
struct RejectOut {
    u64* bits;
    u32* tile_rejects;
    u32* rej_prefix;
    u32* count;
};

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
// kernels_filter.hip
void fzb_launch_filter(const CorpusDev& c, u64 first, u32 count, const u64* table, const u8* dfa, u32 dead, int rows, int mode, int need, u32 min_len,
                       u64* bitmap, u32* tile_counts, u32* reset_counters, int grid, hipStream_t st, u64* bitmap_m = nullptr, u32* tile_counts_m = nullptr,
                       u64* reject_bits = nullptr, u32* tile_rejects = nullptr, int nul_safe = 0, int acc_lo = -1, const u8* cdfa = nullptr, u32 cdfa_bytes = 0,
                       int cdfa_K = 0, int cdfa_G = 0);
void fzb_launch_scan_rejects(const u32* tile_rejects, u32 ntiles, const u32* reject_count, u32* rej_prefix, hipStream_t st);
void fzb_launch_init_counters(u32* counters, u32 n0, hipStream_t st);  // the 16-word counter block: [0] = n0, the rest 0
void fzb_launch_compact1(const u64* bitmap, const u32* counts, u32 n_items, const u32* n_items_ptr, const u32* src, u32* out_idx, u32* total_out, int grid, hipStream_t st,
                         u32* total_out2 = nullptr);
void fzb_launch_filter_items(const CorpusDev& c, u64 first, const u32* items, const u32* n_items_ptr, const u64* table, int rows, int mode, int need, u32 min_len,
                             u64* bitmap, u32* tile_counts, int grid, hipStream_t st);
void fzb_launch_compact2(const u64* bitmap, const u32* counts, const u32* n_items_ptr, const u32* in_idx, const u32* in_win, u32* out_idx, u32* out_win, u32* total_out,
                         int grid, hipStream_t st);
// kernels_window.hip
void fzb_launch_window(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleDev& nd, int pf_lanes,
                       u32* win, u64* bitmap2, u32* tile_counts2, u32* counters, int grid, hipStream_t st, const RejectOut* decide = nullptr, u32 max_items = 0);
// kernels_dp.hip
void fzb_launch_dp(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                   int sw_lanes, int mode, int wmode, int pad_ok, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, int grid, hipStream_t st,
                   const RejectOut* rejects = nullptr);
bool fzb_dp_short_applies(const CorpusDev& c, int sw_lanes, int mode);
void fzb_launch_compact1_classify(const CorpusDev& c, u64 first, const u64* bitmap, const u32* tile_counts, u32 n_items, u32* out_idx, u32* total_out, const NeedleDev& nd, int sw_lanes,
                                  int wmode, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, u32* win_out, u32* lists, u32 list_stride, int grid, hipStream_t st,
                                  int split_multi);
void fzb_launch_dp_classes(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win_in, const u32* n_items_ptr, const NeedleDev& nd, int sw_lanes,
                           int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters, u32* win_out, u32* lists, u32 list_stride,
                           int num_cus, hipStream_t st, int part = 0, int split_multi = 0);
void fzb_launch_classes_all(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* lists, u32 list_stride, const u32* counters,
                            const NeedleDev& nd, int sw_lanes, fzb_match_rec* out, u32 capacity, u32* scratch, int gm, int gc, hipStream_t st);
void fzb_launch_dp_multi(const CorpusDev& c, u64 first, u32 index_offset, const u32* list, const u32* n_list_ptr, const NeedleDev& nd, int sw_lanes, int mode,
                         fzb_match_rec* out, u32 capacity, u32* scratch, int grid, hipStream_t st);
// kernels_unicode.hip
// wmode 2 (the window is the whole haystack): queue the windows wider than a chunk ahead of the single-chunk scorer (which then gets multi_front = 2)
void fzb_launch_unicode_split_wide(const CorpusDev& c, u64 first, const u32* items, const u32* n_items_ptr, int sw_lanes, u32 capacity, u32* overflow, u32 qcap, u32* counters,
                                   int grid, hipStream_t st);
void fzb_launch_dp_unicode(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                           int sw_lanes, int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32 qcap, u32* counters,
                           int grid, hipStream_t st, int tform = 0, int multi_front = 0, int one_round_wgs = 0);
// windows of sw_lanes < m <= 1024 bytes queued by fzb_launch_dp_unicode (front of `overflow`, count in counters[3]); `scratch` as the ASCII multi-chunk scorer's
void fzb_launch_dp_unicode_multi(const CorpusDev& c, u64 first, u32 index_offset, const u32* list, const u32* n_list_ptr, const NeedleDev& nd, int sw_lanes,
                                 fzb_match_rec* out, u32 capacity, u32* scratch, int grid, hipStream_t st, u32 only_from = 0, int tform = 0, u32* counters = nullptr,
                                 u32* back_end = nullptr, u32 fwd_cap = 0);  // fwd_cap > 0: windows beyond four chunks handed on to the queue's back (counters[4], [7])  // runs only when *n_list_ptr >= only_from; tform: dp_unicode_multi_chunk_t
// kernels_sort.hip
void fzb_launch_sort(fzb_match_rec* buf, fzb_match_rec* tmp, const u32* n_ptr, u32* hist, u32 ntiles_cap, int reverse_first, int by_score, int grid, hipStream_t st, int passes = 2);
void fzb_launch_concat_runs(const RunSet& rs, const u32* base_in, u32* total_out, fzb_match_rec* out, u32 capacity, int grid, u32* cut_flag, hipStream_t st);
// kernels_multi.hip
void fzb_launch_records_to_items(const fzb_match_rec* cand, const u32* n_ptr, u32 index_offset, u32* items, int grid, hipStream_t st);
void fzb_launch_identity_records(fzb_match_rec* out, u32 n, u32 index_offset, u32* count_out, int grid, hipStream_t st);
void fzb_launch_join_add(fzb_match_rec* hits, const u32* n_hits_ptr, const fzb_match_rec* cand, const u32* n_cand_ptr, int grid, hipStream_t st);
void fzb_launch_remove_hits(const fzb_match_rec* cand, const u32* n_cand_ptr, const fzb_match_rec* hits, const u32* n_hits_ptr, u64* bitmap, u32* tile_counts,
                            fzb_match_rec* out, u32* total_out, int grid, hipStream_t st);
void fzb_launch_copy_records(const fzb_match_rec* in, const u32* n_ptr, fzb_match_rec* out, u32 capacity, u32* count_out, int grid, hipStream_t st);
// kernels_literal.hip
void fzb_launch_literal_filter(const CorpusDev& c, u64 first, u32 count, const u32* items, const u32* n_items_ptr, const NeedleDev& nd, int mode, u64* bitmap, u32* tile_counts,
                               int grid, hipStream_t st);
void fzb_launch_literal_score(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* n_items_ptr, const NeedleDev& nd, int mode, fzb_match_rec* out,
                              u32 capacity, u32* dev_count, u32* tpos, u32* tnpos, u32 tstride, int grid, hipStream_t st);
// kernels_generic.hip
void fzb_launch_generic(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* list, const u32* n_list_ptr,
                        const NeedleDev& nd, int sw_lanes, int unicode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, int grid, hipStream_t st,
                        int list_forward = 0, u32 only_below = 0, const u32* alt_list = nullptr, const u32* alt_count = nullptr);  // only_below + alt_list: from only_below entries on the launch walks the alternative list (downwards) instead of returning  // list_forward: entry q at list + 4 q; only_below: run only while *n_list_ptr < only_below
size_t fzb_trace_scratch_words(const NeedleDev& nd, int grid);
// long needles (NeedleLongDev): kernels_window.hip / kernels_generic.hip / kernels_literal.hip
size_t fzb_window_long_scratch_bytes(const NeedleLongDev& nd, int grid);
void fzb_launch_window_long(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleLongDev& nd, int pf_lanes, u32* win, u64* bitmap2,
                            u32* tile_counts2, void* scratch, int grid, hipStream_t st);
size_t fzb_generic_long_adj_bytes(const NeedleLongDev& nd, int sw_lanes, int grid);
size_t fzb_trace_scratch_words_long(const NeedleLongDev& nd, int grid);
void fzb_launch_generic_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd,
                             int sw_lanes, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, u16* adj, const u32* cells, u32* pos, u32* npos, u32 stride, int grid,
                             hipStream_t st, const u32* list = nullptr);
#define FZB_LONG_LDS_ROWS 1024  // rows of a long needle k2d_dp_long_quad stages in LDS (longer needles take k2d_dp_long)
// long needles, ASCII: one thread per window of up to 1024 bytes (kernels_dp.hip, k2d_dp_long); wider windows are queued for the launch above
size_t fzb_dp_long_scratch_words_per_thread(const NeedleLongDev& nd, int sw_lanes);
void fzb_launch_dp_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd, int sw_lanes,
                        int bias_ok, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* scratch, u32* queue, u32* counters, int grid, hipStream_t st);
size_t fzb_dp_long_quad_words_per_block(const NeedleLongDev& nd, int sw_lanes);
void fzb_launch_dp_long_quad(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd, int sw_lanes,
                             int upper, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* scratch, u32* queue, u32* counters, int grid, hipStream_t st);
void fzb_launch_literal_filter_long(const CorpusDev& c, u64 first, u32 count, const u32* items, const u32* n_items_ptr, const NeedleLongDev& nd, int mode, u64* bitmap,
                                    u32* tile_counts, int grid, hipStream_t st);
void fzb_launch_literal_score_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* n_items_ptr, const NeedleLongDev& nd, int mode, fzb_match_rec* out,
                                   u32 capacity, u32* dev_count, u32* tpos, u32* tnpos, u32 tstride, int grid, hipStream_t st);
void fzb_launch_generic_trace(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleDev& nd,
                              int sw_lanes, int unicode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, u32* cells, u32* pos, u32* npos, u32 stride,
                              int grid, hipStream_t st);
#endif
