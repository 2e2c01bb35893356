// Host-side internals shared by the translation units of the C ABI (host.hip: matcher + pipeline; host_upload.hip: corpus upload;
// host_shard.hip: the multi-device form).  Not part of the boundary: include/frizbee_hip.h is.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <string>
#include <vector>

#include "../../include/frizbee_hip.h"
#include "fzb_internal.h"
#include "knobs.h"

// error state of the calling thread (fzb_last_error) + the usual early return
int fzb_fail(int code, const std::string& msg);
void fzb_clear_error();
#define HIPCHK(expr)                                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fzb_fail(FZB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline hipError_t fzb_dev_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16); }

struct fzb_corpus {
    CorpusDev dev{};
    void* own_bytes = nullptr;
    void* own_ends = nullptr;
    void* own_view[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // the filter's view: vbytes, vgofs, vgnv, vlen, vperm, vlong (CorpusDev)
};

// what the synchronous entry points remember between two results (fetch_records, host.hip)
struct FetchHint {
    size_t last = 0;            // records of the previous result: the next one's records are copied speculatively, behind the count
    u32* count_host = nullptr;  // page-locked landing place of the two counters
};

struct fzb_matcher {
    fzb_config config{};
    std::string needle;
    bool empty = false, case_sensitive = false, unicode = false, use_u8 = false;
    int literal_mode = 0;  // 0 = fuzzy; else FZB_MATCH_EXACT / PREFIX / SUFFIX / SUBSTRING (src/literal)
    int rows = 0;
    NeedleDev nd{};
    LaunchCfg lc{};
    std::vector<u64> table;  // host copy of the filter table
    std::vector<u8> dfa;     // host copy of the subsequence DFA
    std::vector<u8> uni_dfa; // unicode path, 0 typos: byte-level DFA of the exact prefilter (empty if it needs more than 255 states)
    int uni_dfa_states = 0;
    // typo configurations: the bit-vector LCS test of the streaming filter as a DFA over its REACHABLE states (empty when there are more
    // than 226): state 0 = nothing matched, states >= lcs_acc_lo accept (LCS >= rows - max_typos)
    std::vector<u8> lcs_dfa;
    int lcs_states = 0, lcs_acc_lo = 0;
    bool lcs_scalar = false;  // unicode typo configurations: `lcs_dfa` is the SCALAR-level automaton (exact criterion), not the byte-level one over the table's masks
    // The matcher's streaming automaton (cdfa_src: 1 = `dfa` (subsequence / KMP), 2 = `uni_dfa`, 3 = `lcs_dfa`) with G transitions composed
    // over its K byte classes: [256 bytes: byte -> class][states x K^G: next state], for the ragged filter (kernels_filter.hip,
    // k1_cdfa_ragged).  Empty when states x K^G does not fit 16 KB even for G = 2.
    std::vector<u8> cdfa;
    int cdfa_src = 0, cdfa_K = 0, cdfa_G = 0;
    Workspace ws{};
    int device = -1;
    bool profiling = false;
    static constexpr int PROF_SLOTS = 32;  // ring of per-call events: [0]=pipeline start [1]=pipeline end [2],[3]=around the filter kernel [4]=before the scorers
    hipEvent_t evring[PROF_SLOTS][5] = {};
    hipStream_t aux_stream = nullptr;  // second stream of a query (multi-chunk scorer beside the class launches) and its fork / join events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int ev_filter[PROF_SLOTS] = {};
    u64 prof_calls = 0;

    u32 last_counters[4] = {0, 0, 0, 0};
    // staging for the synchronous API
    fzb_match_rec* out_dev = nullptr;
    size_t out_cap = 0;
    u32* count_dev = nullptr;
    FetchHint fetch;
    // fzb_match_list_indices: the selection (+ its length), the positions (`stride` per record) and their counts
    u32* trace_sel = nullptr;
    u32* trace_pos = nullptr;
    u32* trace_npos = nullptr;
    size_t trace_cap = 0, trace_pos_words = 0;
    // a needle beyond NeedleDev's arrays (> 64 bytes or > 63 rows): scalars in `ndl`, the arrays in one device blob (uploaded on first
    // use), and the global scratch of its kernels (N-typo path state, per-row previous-chunk vectors, traced cells)
    bool long_needle = false;
    bool long_dfa = false;  // a long ASCII needle of up to 200 rows, 0 typos: `dfa` holds its ordered-subsequence automaton and the streaming filter is the first stage
    bool long_upper = false;  // a long ASCII needle with an uppercase letter among its rows (k2d_dp_long's form)
    NeedleLongDev ndl{};
    std::vector<u8> long_blob_host;
    size_t long_off_c = 0, long_off_f = 0, long_off_uc = 0, long_off_uf = 0, long_off_ulen = 0;
    void* long_blob_dev = nullptr;
    void* long_scratch = nullptr;
    size_t long_scratch_bytes = 0;
    // multi-device form (host_shard.hip): the per-shard clones of this matcher (their device state lives on the shard's device);
    // on a clone: its stream and the device it is bound to
    std::vector<fzb_matcher*> shard_clones;
    void* shard_workers = nullptr;     // on the parent: the persistent worker threads, one per shard (host_shard.hip)
    hipStream_t shard_stream = nullptr;  // on a clone: the shard's stream; on the parent: the root's stream (ordering + copy to the host)
    hipEvent_t shard_event = nullptr;    // on a clone: "the run has arrived on the root"
    u32* shard_count_host = nullptr;     // on a clone: page-locked landing place of its record count
    int shard_device = -1;
    // on the parent: peer access between the root and every other device a shard lived on, decided once per (root, device) pair
    // (1 = enabled in both directions: runs travel device to device over xGMI; 0 = refused by the runtime: hipMemcpyPeerAsync stages
    // them through host memory), and the last query's report of how every shard's run reached the root (fzb_matcher_shard_report)
    std::vector<std::array<int, 3>> shard_peers;  // {root, device, state}
    std::string shard_report;
};


// pooled page-locked host buffers (result lists, upload staging): get() returns nullptr on failure; put() returns false for a pointer
// that is not the pool's
void* fzb_pinned_get(size_t bytes);
bool fzb_pinned_put(void* p);

// host.hip internals used by the other translation units
int fzb_bind_device(fzb_matcher* m);
// count + records of a result in device memory -> a pooled pinned host buffer, with ONE synchronisation when the previous result's size
// was a good guess (host.hip); dev_words = 8 u32, the record count is word n_word, the others stay readable in h.count_host
// the wait of a synchronous entry point: polls the stream for up to FZB_SPIN_WAIT_US (default 1 ms), then blocks (host.hip)
hipError_t fzb_stream_wait(hipStream_t st);
int fzb_fetch_records(FetchHint& h, const void* dev_records, const u32* dev_words, int n_word, size_t capacity, hipStream_t st, fzb_match** out, size_t* out_len);
int fzb_ensure_out_staging(fzb_matcher* m, size_t count);
// the ordering post-step of `match_list` on the device (host.hip, next to fzb_sorted_range_device)
struct OrderPlan {
    bool reversed, by_score, one_pass, via_tmp;
    fzb_match_rec* in;  // where the index-ordered records have to be written before fzb_order_finish
};
int fzb_order_begin(fzb_matcher* m, size_t cap, fzb_match_rec* dev_out, OrderPlan* p);
int fzb_order_finish(fzb_matcher* m, const OrderPlan& p, fzb_match_rec* dev_out, const u32* dev_count, hipStream_t stream);
// `Matcher::match_list` over the sub-range [first, first + count) of a corpus, records numbered from index_offset, ordered per
// config.sort on the device (fzb_match_list_sorted_device = the whole corpus from 0)
// k_merge_matches_by_* (src/k_merge.rs:56-132) over runs given by pointer
int fzb_k_merge_runs(int32_t sort, const fzb_match* const* runs, const size_t* run_lens, size_t nruns, fzb_match* out);
void fzb_shard_workers_free(void* workers);  // host_shard.hip
int fzb_build_filter_view(fzb_corpus* c);     // host_upload.hip
int fzb_sorted_range_device(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match* dev_out, size_t capacity, uint32_t* dev_count,
                            void* stream);
