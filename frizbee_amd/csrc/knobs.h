// Every environment switch of the library, in ONE place, parsed once on first use (fzb_knobs(); no launch path calls getenv).  What is left
// after round 6's prune are debugging aids and switches that force a SHIPPING alternative - the older form of a stage that still serves the
// scorings, needles or corpora outside the fast form's preconditions - so that tests can run it against the oracle on ordinary inputs
// (tests/test_gpu_knobs.py, through fzb_debug_reload_knobs()).  Measured-and-rejected variants are not kept behind switches: profiles/HISTORY.md.
#pragma once
#include <stdint.h>

struct FzbKnobs {
    // --- debugging aids ---
    bool debug_sync = false;         // FZB_DEBUG_SYNC=1        synchronise and report after every stage of the pipeline
    bool typo_exact_window = false;  // FZB_TYPO_EXACT_WINDOW=1 no typo fast path: every survivor of a typo query re-decided at the exact lane width (nothing rests on DESIGN.md "Typo configurations")
    bool no_filter_view = false;     // FZB_FILTER_VIEW=0       no interleaved filter view (not built at upload - saves a second copy of the bytes in HBM -, not used by the filter)
    bool verify_promises = true;     // FZB_VERIFY_PROMISES=0   fzb_corpus_set_uniform_len / _set_max_len on BORROWED memory accepted without the device pass over the end offsets
    // --- force the form that serves inputs outside the fast form's preconditions ---
    bool no_lcs_dfa = false;         // FZB_NO_LCS_DFA=1        typo filter: the bit-vector kernel k1_filter (needles whose LCS automaton has more than 226 states) instead of the automaton in k1_dfa
    bool no_cdfa = false;            // FZB_NO_CDFA=1           ragged filter: the byte automaton (automata whose class-composite table exceeds 16 KB) instead of the composite one
    bool no_dp_classes = false;      // FZB_NO_DP_CLASSES=1     ASCII scorer: per-wave choice of computed lanes (k2b_dp: scorings outside dp_cf.h) instead of classified scoring
    bool no_fused_classify = false;  // FZB_NO_FUSED_CLASSIFY=1 ragged ASCII lists: k_compact1 + k2w_classify as two launches (the form typo queries and item lists take) instead of k_compact1_classify
    bool no_dp_cfm = false;          // FZB_NO_DP_CFM=1         multi-chunk scorer in its first form (dp_body.h: scorings outside dp_cfm.h)
    bool no_dp_cfu = false;          // FZB_NO_DP_CFU=1         unicode scorer in its first form (scorings outside dp_unicode.h's biased form)
    int unicode_multi = -1;          // FZB_UNICODE_MULTI=0|1   unicode windows of 65..1024 bytes: never / always thread per haystack (k2u_dp_unicode_multi); default: by the queue's length, on the device
    int coop_below = -1;             // FZB_COOP_BELOW=n        multi-chunk ASCII windows of a ragged list: four lanes per window (dp_quad.h) when fewer than n are queued (0: never; default: 48 per workgroup of the slice = 49 152 on 256 CUs; below 32 768 windows of any kind the single-chunk ones too)
    int park_lds_kb = 37;            // FZB_PARK_LDS_KB         multi-chunk scorer: parked rows in LDS when they fit this many KB per workgroup (0: always the global slab, as for needles of many rows)
    int spin_wait_us = 1000;         // FZB_SPIN_WAIT_US=n      synchronous entry points poll the stream for up to n us before they block (0: block at once, as hipStreamSynchronize does)
    // --- multi-device form ---
    bool shard_gather_copy = false;  // FZB_SHARD_GATHER=copy   counts to the host + hipMemcpyPeerAsync even when every shard shares the root device (the form shards on other devices take)
    int shard_inline = -1;           // FZB_SHARD_INLINE=0|1    shards on the root device: 0 = through the worker threads, 1 = enqueued by the caller
};

const FzbKnobs& fzb_knobs();
