// Every environment switch of the library, in ONE place.  They exist for comparison and debugging only - the defaults are what is measured
// and shipped - and are parsed once, on first use (fzb_knobs()); no launch path calls getenv.  tests/test_gpu_knobs.py runs every
// alternative path against the oracle on the GPU through fzb_debug_reload_knobs() (test hook: re-parses the environment).
#pragma once
#include <stdint.h>

struct FzbKnobs {
    // --- which form of a stage runs (the alternative is always the older / literal form of the same arithmetic) ---
    bool no_lcs_dfa = false;         // FZB_NO_LCS_DFA=1       typo filter: the bit-vector kernel k1_filter instead of the LCS automaton in k1_dfa
    bool no_dp_cfu = false;          // FZB_NO_DP_CFU=1        unicode scorer in its first form
    bool typo_exact_window = false;  // FZB_TYPO_EXACT_WINDOW=1 no typo fast path: every survivor re-decided at the exact lane width
    bool no_dp_classes = false;      // FZB_NO_DP_CLASSES=1    per-wave choice of computed lanes (k2b_dp) instead of classified scoring
    bool no_overlap = false;         // FZB_NO_OVERLAP=1       multi-chunk scorer on the caller's stream (only matters with FZB_SMALL_LIST)
    bool no_dp_cfm = false;          // FZB_NO_DP_CFM=1        multi-chunk scorer in its first form (dp_body.h)
    bool no_tail_classes = false;    // FZB_NO_TAIL_CLASSES=1  every last chunk of a multi-chunk window computed in full
    bool no_cdfa = false;            // FZB_NO_CDFA=1          ragged filter: the byte automaton instead of the class-composite one
    bool no_filter_view = false;     // FZB_FILTER_VIEW=0      no interleaved filter view (not built at upload, not used by the filter)
    bool cdfa_nodfa = false;         // FZB_CDFA_NODFA=1       MEASUREMENT ONLY: the ragged filter's loads without the automaton (results meaningless)
    bool ragged_burst = true;        // FZB_RAGGED_BURST=0     rolling form of the canonical-layout ragged filter
    bool debug_sync = false;         // FZB_DEBUG_SYNC=1       synchronise and report after every stage of the pipeline
    bool window_four_pass = false;   // FZB_WINDOW_FOUR_PASS=1 lane-exact window kernel on small lists: 256-thread workgroups, four passes per tile (instead of 1024 threads, one pass)
    bool window_no_mask_cache = false;  // FZB_WINDOW_NO_MASK_CACHE=1 lane-exact window kernel: a needle row's occurrence mask recomputed at every request (no LDS cache)
    bool long_generic_only = false;  // FZB_LONG_GENERIC_ONLY=1 long needles scored by the wave-per-haystack kernel alone (rounds 3-4) instead of one thread per window (k2d_dp_long)
    bool window_whole_tiles = false; // FZB_WINDOW_WHOLE_TILES=1 the PRE form as one 1024-thread workgroup per tile instead of four 256-thread workgroups per tile
    bool window_no_pre = false;      // FZB_WINDOW_NO_PRE=1    lane-exact window kernel, one-pass form: every thread computes its own haystack's occurrence masks chunk by chunk (round 4's form) instead of the workgroup laying them out ahead
    bool no_unicode_fwd = false;     // FZB_UNICODE_FWD=0      the thread-per-haystack unicode multi-chunk scorer keeps its windows beyond four chunks (default: hands up to 4096 on to the wave-per-haystack kernel)
    bool no_handoff = true;          // FZB_HANDOFF=1 (or naming FZB_HANDOFF_MIN_TILES) turns the filter -> scorer handoff ON; default since round 5 and FZB_NO_HANDOFF=1: classifier and scorers gather the survivors' bytes from the corpus (no staging)
    bool shard_gather_copy = false;  // FZB_SHARD_GATHER=copy  multi-device query: counts to the host + hipMemcpyPeerAsync even when every shard shares the root device
    bool dfa_general = true;         // FZB_DFA_UNI32=1        k1_dfa's instantiation without per-lane lengths on a uniform list of 32-byte haystacks (round 5 experiment: same time); default: the general form
    bool dfa_stride256 = false;      // FZB_DFA_STRIDE256=1    k1_dfa's table at a 256-byte row pitch: the v_perm result is the address (one VALU instruction per byte), more LDS bank conflicts
    bool view_read_len = false;      // FZB_VIEW_READ_LEN=1    the view filter reads the haystacks' lengths even when nothing needs them (round 4's form)
    bool view_plain_loads = false;   // FZB_VIEW_PLAIN_LOADS=1 the view filter's loads without the non-temporal hint
    bool verify_promises = true;     // FZB_VERIFY_PROMISES=0  fzb_corpus_set_uniform_len / _set_max_len on BORROWED memory accepted without the device pass over the end offsets
    int shard_inline = -1;           // FZB_SHARD_INLINE=0|1   multi-device query, shards on the root device: 0 = through the worker threads, 1 = enqueued by the caller
    int handoff_min_tiles = 4096;    // FZB_HANDOFF_MIN_TILES  the handoff only for lists of at least this many 1024-haystack tiles (0: always)
    int unicode_multi = -1;          // FZB_UNICODE_MULTI=0|1  unicode windows of 65..1024 bytes: never / always thread per haystack (k2u_dp_unicode_multi); default: by the queue's length
    int generic_wgs = 12;            // FZB_GENERIC_WGS        workgroups per CU of the wave-per-haystack kernel over a queue of wide unicode windows
    int park_lds_kb = 37;            // FZB_PARK_LDS_KB        multi-chunk scorer: parked rows in LDS when they fit this many KB per workgroup (0: always the global slab)
    int window_dbg = 0;              // FZB_WINDOW_DBG=bits    MEASUREMENT ONLY: phases of the window kernel's PRE form switched off (results meaningless)
    int stage_dbg = 0;               // FZB_STAGE_DBG=bits     MEASUREMENT ONLY: parts of the view kernel's staging switched off (results meaningless)
    int k2u_waves = 0;               // FZB_K2U_WAVES=3        unicode scorer's biased form capped at three waves per SIMD (spills)
    uint32_t small_list = 0xFFFFFFFFu;  // FZB_SMALL_LIST=n   lists of n haystacks and more: four scorer launches on two streams instead of k2_classes_all
    // --- tuning (grid shapes) ---
    int compact_grid_mul = 4;        // FZB_COMPACT_GRID_MUL   workgroups per CU of k_compact1
    int classify_per = 2;            // FZB_CLASSIFY_PER=1|4   survivors per thread of k2w_classify
    int dp_wgs_per_cu = 0;           // FZB_DP_WGS_PER_CU      fewer resident workgroups of the short scorer (0 = occupancy)
    int dfa_wgs = 8;                 // FZB_DFA_WGS            workgroups per CU of the streaming filter on short / uniform lists (k1_dfa; 8 = every wave slot)
    int cdfa_wgs = 5;                // FZB_CDFA_WGS           workgroups per CU, class-composite filter on the canonical layout
    int view_wgs = 6;                // FZB_VIEW_WGS           workgroups per CU, filter over the view
    int ragged_wgs = 8;              // FZB_RAGGED_WGS         workgroups per CU, burst filter
    // --- upload ---
    int upload_mode = 2;             // FZB_UPLOAD_MODE=register|staged (default direct = 2; register = 1; staged = 0)
    int upload_threads = 0;          // FZB_UPLOAD_THREADS     worker threads of the staged mode
};

const FzbKnobs& fzb_knobs();
