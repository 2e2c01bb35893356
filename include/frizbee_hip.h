/*
 * frizbee_hip.h - C ABI of the MI355X (gfx950) backend for saghen/frizbee's batched fuzzy-scoring path.
 *
 * This is the drop-in boundary: a Rust `MatcherBackend::Hip` variant binds exactly
 * these entry points.  Each one names the reference interface it replaces (paths relative to the
 * reference crate root).  Plain pointers and sizes only; nothing here depends on torch or HIP types
 * (`void* stream` is a `hipStream_t`, NULL = the default stream).
 *
 * All functions return FZB_OK (0) or an error code; fzb_last_error() returns the message for the
 * calling thread.  Where the reference panics, the message text is the reference's panic text.
 * No function ever unwinds across the boundary.
 */
#ifndef FRIZBEE_HIP_H
#define FRIZBEE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FZB_OK = 0,
    FZB_ERR_INVALID = 1,      /* bad argument (NULL, bad enum, invalid UTF-8 needle)                      */
    FZB_ERR_PANIC = 2,        /* the reference would panic (message = its panic text)                     */
    FZB_ERR_UNSUPPORTED = 3,  /* reserved: nothing returns it since round 3 (needles of every accepted length are handled) */
    FZB_ERR_HIP = 4,          /* HIP runtime error / no device                                            */
    FZB_ERR_CAPACITY = 5      /* caller-provided device buffer too small                                  */
};

/* src/lib.rs:357-368 CaseMatching, :379-392 UnicodeMatching, :311-326 SortStrategy */
enum { FZB_CASE_IGNORE = 0, FZB_CASE_SMART = 1, FZB_CASE_RESPECT = 2 };
enum { FZB_UNICODE_IGNORE = 0, FZB_UNICODE_SMART = 1, FZB_UNICODE_ALWAYS = 2 };
enum { FZB_SORT_SCORE_THEN_INDEX_ASC = 0, FZB_SORT_SCORE_THEN_INDEX_DESC = 1, FZB_SORT_INDEX_ASC = 2, FZB_SORT_INDEX_DESC = 3 };
/* src/lib.rs:414-427 Matching: fuzzy (Smith-Waterman) or one of the literal modes of src/literal (contiguous occurrence) */
enum { FZB_MATCH_FUZZY = 0, FZB_MATCH_EXACT = 1, FZB_MATCH_PREFIX = 2, FZB_MATCH_SUFFIX = 3, FZB_MATCH_SUBSTRING = 4 };

/* src/lib.rs:439-478 `Scoring` (same field order as the Rust struct declaration) */
typedef struct fzb_scoring {
    uint16_t match_score, mismatch_penalty, gap_open_penalty, gap_extend_penalty;
    uint16_t prefix_bonus, capitalization_bonus, matching_case_bonus, exact_match_bonus, delimiter_bonus;
} fzb_scoring;

/* src/lib.rs:236-258 `Config` (+ per-pattern overrides already resolved, src/pattern.rs:250-262).
 * `matching`: the literal modes (src/literal/algo.rs) ignore max_typos and do not depend on pf_lanes / sw_lanes.
 * pf_lanes / sw_lanes select WHICH reference CPU backend the results are bit-exact against, because
 * frizbee's scores, windows and typo-prefilter decisions depend on the SIMD lane count: both 0 = pick the
 * pair the reference's `Matcher::get_backend` (src/matcher/mod.rs:448-498) would pick on THIS host CPU
 * (`is_x86_feature_detected!` predicates) for the needle's score class: AVX-512(+VBMI for the u8 class):
 * prefilter 64, score 64 (u8) / 32 (u16); AVX2: 32, 32 / 16; SSE4.1 or scalar: 16, 16 / 8.
 * Non-zero values force a pair (pf_lanes in {16,32,64}, sw_lanes in {8,16,32,64}); pf_lanes set with sw_lanes 0 = that ISA
 * family's score width for the needle's class (what a multi-pattern matcher needs: its patterns may differ in class). */
typedef struct fzb_config {
    int32_t max_typos; /* Option<u16>: -1 = None (no prefilter) */
    int32_t casing;    /* FZB_CASE_*    */
    int32_t unicode;   /* FZB_UNICODE_* */
    int32_t sort;      /* FZB_SORT_*    */
    fzb_scoring scoring;
    uint16_t pf_lanes, sw_lanes;
    int32_t matching;  /* FZB_MATCH_* */
} fzb_config;

/* src/lib.rs:141-153 `Match` with an explicit layout (Rust's is unspecified; the shim copies field-wise) */
typedef struct fzb_match {
    uint32_t index;
    uint16_t score;
    uint8_t exact;
    uint8_t _pad;
} fzb_match;

typedef struct fzb_matcher fzb_matcher; /* replaces `Matcher` / `MatcherImpl<P,S>` (src/matcher/mod.rs:77-82, algo.rs:47-54) */
typedef struct fzb_corpus fzb_corpus;   /* the `&[S: AsRef<str>]` haystack list, packed and resident in HBM           */

const char* fzb_last_error(void);

/* `Config::default()` (src/lib.rs:260-271) / `Scoring::default()` (src/lib.rs:463-478), lanes = 0 (auto) */
void fzb_config_default(fzb_config* out);

/* `Matcher::new(needle, &config)` (src/matcher/mod.rs:90-111, 178-204) -> `MatcherImpl::new` (algo.rs:57-71):
 * resolves Smart casing/unicode, picks the u8/u16 score class (smith_waterman/mod.rs:92-116), runs
 * `guard_against_score_overflow` (lib.rs:506-537; FZB_ERR_PANIC with the reference's text), builds the
 * device-side needle tables.  An empty needle is valid (matches everything with score 0, mod.rs:381-384). */
int fzb_matcher_create(const fzb_config* config, const uint8_t* needle_utf8, size_t needle_len, fzb_matcher** out);
int fzb_matcher_clone(const fzb_matcher* m, fzb_matcher** out); /* `impl Clone for Matcher` (parallel.rs:46) */
/* `Matcher::set_pattern` / `Matcher::set_config` (src/matcher/mod.rs:154-176): rebuild for a new needle / config; no-ops when nothing
 * changed.  The device workspace (sized by the corpus) is kept: re-querying a resident corpus after every keystroke allocates nothing.
 * On error the matcher is left unchanged. */
int fzb_matcher_set_pattern(fzb_matcher* m, const uint8_t* needle_utf8, size_t needle_len);
int fzb_matcher_set_config(fzb_matcher* m, const fzb_config* config);
void fzb_matcher_free(fzb_matcher* m);
/* introspection: out[0]=pf_lanes out[1]=sw_lanes out[2]=u8 class? out[3]=case_sensitive out[4]=unicode path? out[5]=rows */
int fzb_matcher_info(const fzb_matcher* m, int32_t out[6]);

/* Packs what `match_list(&haystacks)` borrows (src/matcher/mod.rs:212) into device memory:
 * `bytes` = all haystacks concatenated, `end_offsets[i]` = exclusive end of haystack i (n entries).
 * The copy is owned by the library and outlives calls, so repeated queries amortise the upload. */
int fzb_corpus_upload(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, fzb_corpus** out);
/* Same, but the data already lives in HBM (e.g. a torch tensor's data_ptr()) in the library's device layout
 * ("padded-16"): haystack i starts at start(i) = i ? roundup16(dev_ends[i-1]) : 0 and ends (exclusive) at
 * dev_ends[i]; bytes between haystacks are zero; dev_bytes is 16-byte aligned and has >= 80 readable zero
 * bytes after the last haystack.  dev_ends has n entries (uint32 if ends_are_u64 == 0).  Borrowed, not copied.
 * (A list of 32-byte haystacks stored back to back already has this layout.) */
int fzb_corpus_from_device(const void* dev_bytes, const void* dev_ends, int ends_are_u64, size_t n, uint64_t total_bytes, fzb_corpus** out);
/* Optional hint for borrowed corpora: the longest haystack in bytes.  Must be an upper bound; 0 = unknown.  fzb_corpus_upload measures
 * it: on a corpus it uploaded a looser value (or 0) is ignored and a value below the measured one is refused (FZB_ERR_INVALID).  On
 * borrowed memory the bound is CHECKED when it is given: one device pass over the end offsets (a set-up call: it synchronises the
 * device); a haystack longer than max_len, or offsets that decrease / leave the buffer, make the call fail with FZB_ERR_INVALID and name
 * the first offending index.  FZB_VERIFY_PROMISES=0 in the environment skips the pass. */
int fzb_corpus_set_max_len(fzb_corpus* c, uint32_t max_len);
/* Optional accelerator for borrowed RAGGED corpora (fzb_corpus_upload builds it itself): the streaming filter's view of the list - a
 * second copy of the bytes, every 1024-haystack tile sorted by length and stored interleaved in groups of 64, so that a wavefront's
 * loads are contiguous.  The lengths are read from the end offsets (no hint is trusted).  A list that does not
 * call for a view (nothing beyond 32 bytes, more than one haystack in 256 beyond 256 bytes, uniform length) or a device without room leaves the corpus as it
 * is: *out_built (optional) = 1 when the corpus has a view when the call returns.  The corpus' bytes must not change afterwards. */
int fzb_corpus_build_view(fzb_corpus* c, int* out_built);
/* Optional promise for borrowed corpora: EVERY haystack has exactly `len` bytes (so haystack i starts at i * roundup16(len)); the hot
 * kernels then compute the spans instead of reading the end offsets (a tenth of the filter's traffic on 32-byte records and one
 * dependent load less per survivor).  fzb_corpus_upload detects it by itself, and on a corpus it uploaded only the detected value is
 * accepted (FZB_ERR_INVALID otherwise).  A non-zero `len` also becomes the corpus' max_len (overwriting fzb_corpus_set_max_len);
 * 0 clears the promise and the bound it implied.  On borrowed memory the promise is CHECKED against the end offsets when it is made (one
 * device pass, as for fzb_corpus_set_max_len): a wrong promise would mis-span every haystack, so it is refused with FZB_ERR_INVALID. */
int fzb_corpus_set_uniform_len(fzb_corpus* c, uint32_t len);
void fzb_corpus_free(fzb_corpus* c);
size_t fzb_corpus_len(const fzb_corpus* c);

/* `Matcher::match_list(&haystacks)` (src/matcher/mod.rs:212-222) = `match_list_into(.., offset 0)` ->
* `Specialized::match_list::<TYPOS,UNICODE,_>` (src/matcher/algo.rs:78-103) and the reverse / `radix_sort_matches`
 * post-step (src/sort.rs:6-40), all on the GPU.  `*out` is malloc'd by the library
 * (free with fzb_matches_free) and holds exactly the Vec<Match> the reference returns, in its order. */
int fzb_match_list(fzb_matcher* m, const fzb_corpus* c, fzb_match** out, size_t* out_len);

/* `Specialized::match_list(&mut self, haystacks, haystack_index_offset, &mut matches)`
 * (src/matcher/algo.rs:17-22, backend.rs:95-107): appends, in input order, one Match per prefilter-passing
 * haystack of the sub-range [first, first+count) with `index = index_offset + (i - first)`; no sorting.
 * This is the seam `MatcherBackend::Hip` implements and what `match_list_parallel`'s per-shard workers call. */
int fzb_match_list_into(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match** out, size_t* out_len);

/* Allocates, once, every device buffer the queries of this matcher over corpus `c` can need (the matcher keeps them across
 * fzb_matcher_set_pattern / fzb_matcher_set_config): after it no query allocates - the keystroke-latency use (SURVEY 8f rank 2,
 * callers `Matcher::set_pattern`, src/matcher/mod.rs:154-176).  Optional: without it the buffers grow on first use. */
int fzb_matcher_reserve(fzb_matcher* m, const fzb_corpus* c);

/* Device-resident form of the above for callers that keep results in HBM (benchmarks, multi-GPU gather):
 * writes the index-ordered records to dev_out (capacity records) and the counts to dev_count - TWO uint32 in device memory:
 * dev_count[0] = records written = min(matches, capacity), dev_count[1] = matches found (not clamped) - asynchronously on
 * `stream`.  No host synchronisation.
 * CAPACITY: nothing on the host knows the number of matches when the call returns, so a buffer that is too small cannot be
 * refused: the records at positions >= capacity are NOT written (for the sorted form the sort then orders that truncated prefix -
 * its head is not the head of the full list), and dev_count[1] > capacity is how the reader of the buffer sees it
 * (frizbee_amd.distributed.ShardExchange raises on it).  capacity >= count (one record per haystack of the range) can never
 * truncate.  FZB_ERR_CAPACITY is returned where the host does know: an empty pattern list. */
int fzb_match_list_device(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset,
                          fzb_match* dev_out, size_t capacity, uint32_t* dev_count, void* stream);

/* `Matcher::match_list` entirely in HBM: the records of the WHOLE corpus in the order `config.sort` asks for
 * (reverse for the *Desc strategies, then the stable descending radix sort of src/sort.rs:6-40 for the Score* strategies,
 * both as device kernels).  fzb_match_list is this call + one device-to-host copy. */
int fzb_match_list_sorted_device(fzb_matcher* m, const fzb_corpus* c, fzb_match* dev_out, size_t capacity, uint32_t* dev_count, void* stream);

/* `Matcher::match_list_parallel(&haystacks, threads)` (src/matcher/parallel.rs:18-89).  The GPU processes the
 * whole list in one pass, so `threads` only keeps the reference's contract: 0 => FZB_ERR_PANIC
 * "threads must be positive"; the result equals fzb_match_list for every thread count (parallel.rs:104-130). */
int fzb_match_list_parallel(fzb_matcher* m, const fzb_corpus* c, size_t threads, fzb_match** out, size_t* out_len);

void fzb_matches_free(fzb_match* p);

/* ---- the multi-device form: `Matcher::match_list_parallel` with the GPUs of one node as its workers ---------------------------
 * The reference's match_list_parallel (src/matcher/parallel.rs:18-89) cuts the list into contiguous chunks, hands every worker thread
 * a chunk with its global index offset (:55-63), sorts each worker's run (:66-76) and k-way merges the runs (:78-87).  Here a worker
 * is a DEVICE: the list is cut into `ndev` contiguous shards, shard g resident on device g; a query runs one persistent host thread
 * per shard (hipSetDevice + a per-shard clone of the matcher: the pipeline on that GPU, records in index order) and each run is
 * copied device to device to its place in ONE list on the root device (the caller's current device).  Shard order is ascending index
 * order, so that list is the one `match_list` orders: the root runs the reverse / stable radix sort of src/sort.rs:6-40 once and makes
 * one copy to the host.  The result equals fzb_match_list on the unsharded list for every sort strategy - what the reference's per-run
 * sort + k-way merge produces, without a host-side merge.  (`threads` of fzb_match_list_parallel keeps the reference's contract on ONE
 * device; the number of devices is a property of the sharded corpus, never inferred from `threads`.) */
typedef struct fzb_sharded_corpus fzb_sharded_corpus;
enum {
    FZB_SHARD_BY_COUNT = 0,       /* shard g = [g * ceil(n / ndev), ...): equal haystack counts                           */
    FZB_SHARD_BY_BYTES = 1,       /* shard g starts at the first haystack that starts at or after g/ndev of the bytes (ragged lists) */
    FZB_SHARD_OVERSUBSCRIBE = 2   /* more shards than visible devices is allowed: shard g lives on device g % devices (testing)     */
};
int fzb_device_count(int* out);
/* the shard boundaries (haystack indices) the upload uses: out_bounds has nshards + 1 entries; host arithmetic only */
int fzb_shard_ranges(const uint64_t* end_offsets, size_t n, int nshards, int by_bytes, uint64_t* out_bounds);
/* fzb_corpus_upload of every shard onto its device, the shards in parallel.  FZB_ERR_HIP when fewer than ndev devices are visible
 * (unless FZB_SHARD_OVERSUBSCRIBE). */
int fzb_corpus_upload_sharded(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, int ndev, int flags, fzb_sharded_corpus** out);
void fzb_sharded_corpus_free(fzb_sharded_corpus* sc);
size_t fzb_sharded_corpus_len(const fzb_sharded_corpus* sc);
int fzb_sharded_corpus_shards(const fzb_sharded_corpus* sc);
int fzb_sharded_corpus_shard(const fzb_sharded_corpus* sc, int g, uint64_t* lo, uint64_t* hi, int* device);
/* `Matcher::match_list_parallel(&haystacks, threads)` over the sharded list, one worker per shard.  The matcher keeps one clone of
 * itself per shard (device workspaces on the shards' devices; they follow fzb_matcher_set_pattern / fzb_matcher_set_config, and a
 * clone whose shard lives on another device in a later corpus is rebuilt there) and its worker threads.  The caller's current device
 * is the root (the matcher binds to it like on any first query).  fzb_last_counters on `m` afterwards = the sum over the shards.
 * Free the result with fzb_matches_free. */
int fzb_match_list_parallel_sharded(fzb_matcher* m, const fzb_sharded_corpus* sc, fzb_match** out, size_t* out_len);
/* How the runs of the last fzb_match_list_parallel_sharded on `m` reached the root, as text: the gather form and, per shard, "same
 * device", "peer access enabled (device to device over xGMI)" or "peer access REFUSED ..." (hipDeviceCanAccessPeer /
 * hipDeviceEnablePeerAccess are asked once per (root, device) pair; without peer access the runtime stages hipMemcpyPeerAsync through
 * host memory - correct, slower, and said here rather than silently).  The reference has no counterpart: its workers share one address
 * space (src/matcher/parallel.rs:43-64).  The string lives until the next sharded query on `m` or fzb_matcher_free. */
const char* fzb_matcher_shard_report(const fzb_matcher* m);
/* The combine step alone, for callers that moved the per-shard runs themselves (one process per GPU: the root rank after an RCCL
 * gather - frizbee_amd.distributed): run g = dev_runs[g], index-ordered records of shard g as fzb_match_list_device wrote them,
 * dev_counts[g] -> the two uint32 that call wrote in DEVICE memory (records written, matches found), run_caps[g] = the run's buffer size
 * in records (a run with matches found > run_caps[g] was truncated by its producer: FZB_ERR_CAPACITY, nothing merged); runs in ascending
 * shard order, all readable from the current device.  Concatenation + `match_list`'s ordering (src/matcher/mod.rs:215-221) on the device, on `stream`, then one copy to the
 * host = the list `match_list_parallel` returns (src/matcher/parallel.rs:66-87).  Free the result with fzb_matches_free. */
int fzb_merge_shard_runs(fzb_matcher* m, const void* const* dev_runs, const uint32_t* const* dev_counts, const size_t* run_caps, size_t nruns, void* stream,
                         fzb_match** out, size_t* out_len);

/* `Matcher::match_list_parallel` (src/matcher/parallel.rs:18-89) with ONE PROCESS PER GPU: the workers of parallel.rs:43-64 are the
 * ranks, each holding its contiguous share of the list as its own corpus (fzb_shard_ranges gives the boundaries), and what the
 * reference's workers hand back through shared memory (parallel.rs:66-87) travels by RCCL over xGMI below this boundary - nothing above
 * it (Rust, C++, Python) links a collective library.  RCCL is opened at run time (librccl.so.1): FZB_ERR_HIP with the loader's message
 * when the process cannot get one.
 *   fzb_rccl_unique_id       rank 0 draws the communicator's id (ncclGetUniqueId); the HOST moves these 128 bytes to the other ranks by
 *                            whatever channel started them (environment, file, socket, MPI): the one out-of-band step
 *   fzb_shard_comm_create    COLLECTIVE (every rank of `world` calls it, ncclCommInitRank) on the rank's current device; owns a stream
 *                            and the exchange buffers, which grow on demand and are reused by later queries
 *   fzb_match_list_parallel_rccl   COLLECTIVE: rank g scores `shard` (its share, records numbered from index_offset = the share's first
 *                            global index), the runs' lengths are all-gathered (8 bytes per rank, the call's one host synchronisation
 *                            before the result), then ONE RCCL group moves every run - exactly its records, not a capacity - to rank 0
 *                            (flags = 0), or to every rank (FZB_GATHER_ALL: the all-gather of variable-length runs); a receiver
 *                            concatenates in rank order (= ascending index order), applies `match_list`'s ordering once on its device
 *                            (src/matcher/mod.rs:215-221) and copies the list to its host: *out / *out_len = the list
 *                            `match_list_parallel` returns for the WHOLE list, to be freed with fzb_matches_free; on a rank that does
 *                            not receive, *out = NULL and *out_len = 0.  Every rank must pass a matcher of the same needle and config.
 *   fzb_shard_comm_last_exchange   out_bytes[0] / [1] = record bytes this rank sent / received in its last query.
 * As with any collective: a rank that fails BEFORE the exchange (a bad argument, no memory) leaves the others waiting inside RCCL - check arguments
 * that can differ per rank before the call, and treat an error of a collective call as the end of that communicator.  One communicator per thread;
 * a communicator and the matchers used with it belong to the device that was current in fzb_shard_comm_create. */
#define FZB_RCCL_ID_BYTES 128
typedef struct fzb_shard_comm fzb_shard_comm;
enum { FZB_GATHER_ROOT = 0, FZB_GATHER_ALL = 1 };
int fzb_rccl_unique_id(uint8_t out_id[FZB_RCCL_ID_BYTES]);
int fzb_shard_comm_create(const uint8_t id[FZB_RCCL_ID_BYTES], int rank, int world, fzb_shard_comm** out);
void fzb_shard_comm_free(fzb_shard_comm* comm);
int fzb_shard_comm_rank(const fzb_shard_comm* comm);
int fzb_shard_comm_world(const fzb_shard_comm* comm);
int fzb_match_list_parallel_rccl(fzb_matcher* m, const fzb_corpus* shard, uint32_t index_offset, fzb_shard_comm* comm, int flags, fzb_match** out, size_t* out_len);
int fzb_shard_comm_last_exchange(const fzb_shard_comm* comm, uint64_t out_bytes[2]);

/* `MatchIndices` (src/lib.rs:189-199): a Match plus the haystack byte positions that matched the needle, in reverse
 * order.  positions[positions_begin .. positions_begin + positions_len) of the array returned next to the records. */
typedef struct fzb_match_indices {
    uint32_t index;
    uint16_t score;
    uint8_t exact;
    uint8_t _pad;
    uint32_t positions_begin;
    uint32_t positions_len;
} fzb_match_indices;

/* `Matcher::match_list_indices(&haystacks)` (src/matcher/mod.rs:234-275 -> match_list_indices_impl src/matcher/algo.rs:196-227,
 * smith_waterman_indices_one :264-292, score_haystack[_unicode]_indices src/smith_waterman/algo/mod.rs:49-152, the traceback
 * src/smith_waterman/alignment_iter.rs:35-181; literal modes src/literal/algo.rs:129-155), on the GPU: the scorer keeps its
 * score / match matrices in HBM and walks the alignment back on the device.  The haystack list is the `selection`
 * (n_selection corpus indices, host memory, any order, repeats allowed - typically the top of a fzb_match_list result) or,
 * with selection == NULL, the whole corpus; `index` numbers that list like the reference numbers `haystacks`
 * (selection[index] is the corpus index).  Order: as the reference, list order, reversed for the *Desc strategies, then a
 * stable sort by descending score for the Score* strategies.  Free with fzb_match_indices_free. */
int fzb_match_list_indices(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t n_selection,
                           fzb_match_indices** out, size_t* out_len, uint32_t** out_positions);
/* The unordered form: `Specialized::match_list_indices` (src/matcher/algo.rs:24-33) and what `Matcher::match_iter_indices` /
 * `match_one_indices` (src/matcher/mod.rs:321-334, 357-371) yield - list order whatever `config.sort` says,
 * `index = index_offset + position in the list`. */
int fzb_match_list_indices_into(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, uint32_t index_offset,
                                fzb_match_indices** out, size_t* out_len, uint32_t** out_positions);
void fzb_match_indices_free(fzb_match_indices* matches, uint32_t* positions);

/* `radix_sort_matches(&mut [Match])` (src/sort.rs:6-40): stable, descending score, host side */
void fzb_radix_sort_matches(fzb_match* matches, size_t n);
/* `k_merge_matches_by_*` (src/k_merge.rs:56-132): merges per-shard runs (each sorted per `sort`) - the
 * host-side combine after the multi-GPU gather.  runs = concatenated runs, run_lens[k] records each. */
int fzb_k_merge_matches(int32_t sort, const fzb_match* runs, const size_t* run_lens, size_t nruns, fzb_match* out);

/* ---- multi-pattern composition ----------------------------------------------------------------------
 * `Matcher::from_patterns(&[Pattern], &Config)` (src/matcher/mod.rs:95-111): a haystack matches when every non-negated
 * pattern matches and no negated one does; score = saturating sum of the non-negated patterns' scores, exact = OR
 * (src/matcher/multi.rs:84-152).  One `fzb_pattern` = reference `Pattern{needle, negated, config: PatternConfig}`
 * (src/pattern.rs:9-18, 230-262); the per-pattern overrides are resolved against the matcher's config exactly like
 * `PatternConfig::resolve`, including the matching mode (`Pattern::parse`: `^foo` prefix, `foo$` suffix, `^foo$` exact,
 * `'foo` substring, `!foo` negated substring). */
typedef struct fzb_pattern {
    const uint8_t* needle_utf8;
    size_t needle_len;      /* 0: the pattern is dropped (Matcher::compile, src/matcher/mod.rs:193-195) */
    int32_t negated;
    int32_t has_max_typos;  /* PatternConfig::max_typos = Some(max_typos); 0 = None = inherit the config's */
    int32_t max_typos;
    int32_t casing;         /* FZB_CASE_*, or -1 = inherit */
    int32_t unicode;        /* FZB_UNICODE_*, or -1 = inherit */
    int32_t has_scoring;    /* PatternConfig::scoring = Some(scoring) */
    fzb_scoring scoring;
    int32_t matching;       /* FZB_MATCH_*, or -1 = inherit Config::matching (what Pattern::parse leaves for a plain atom) */
} fzb_pattern;
typedef struct fzb_multi_matcher fzb_multi_matcher;

/* `Pattern::parse_query(query)` (src/pattern.rs:186-222; atoms per `Pattern::parse`, :87-167): whitespace separated atoms, `\\` escapes,
 * atoms with an empty needle dropped.  The array (and the needle strings it points to) is released with fzb_patterns_free. */
int fzb_parse_query(const uint8_t* query_utf8, size_t query_len, fzb_pattern** out_patterns, size_t* out_n);
void fzb_patterns_free(fzb_pattern* patterns, size_t n);

int fzb_multi_matcher_create(const fzb_config* config, const fzb_pattern* patterns, size_t n_patterns, fzb_multi_matcher** out);
void fzb_multi_matcher_free(fzb_multi_matcher* mm);
size_t fzb_multi_matcher_len(const fzb_multi_matcher* mm); /* compiled (non-empty) patterns */
/* `Matcher::match_list` over CompiledPatterns::{Empty, Single, Multi} (src/matcher/mod.rs:212-222, 373-392): ordered per config.sort */
int fzb_multi_match_list(fzb_multi_matcher* mm, const fzb_corpus* c, fzb_match** out, size_t* out_len);
/* `Matcher::match_list_indices` for a `from_patterns` matcher (see fzb_match_list_indices): CompiledPatterns::Multi runs
 * `match_one_indices_multi` (src/matcher/multi.rs:56-82) - a negated pattern that matches drops the haystack, every other pattern
 * must match, scores add with saturation, exact flags OR, and the patterns' positions are merged (descending, de-duplicated). */
int fzb_multi_match_list_indices(fzb_multi_matcher* mm, const fzb_corpus* c, const uint32_t* selection, size_t n_selection,
                                 fzb_match_indices** out, size_t* out_len, uint32_t** out_positions);
/* list-order forms (see fzb_match_list_into / fzb_match_list_indices_into): `Matcher::match_list_into` over CompiledPatterns
 * (src/matcher/mod.rs:373-392) with the result on the host, and what `match_iter` / `match_one` / `match_iter_indices` yield
 * (`match_one_multi`, `match_one_indices_multi`, src/matcher/multi.rs:29-82) */
int fzb_multi_match_list_into(fzb_multi_matcher* mm, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match** out, size_t* out_len);
int fzb_multi_match_list_indices_into(fzb_multi_matcher* mm, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, uint32_t index_offset,
                                      fzb_match_indices** out, size_t* out_len, uint32_t** out_positions);
/* `match_list_multi_into(patterns, haystacks, haystack_index_offset, matches)` (src/matcher/multi.rs:84-152): index order, result in HBM */
int fzb_multi_match_list_device(fzb_multi_matcher* mm, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset,
                                fzb_match* dev_out, size_t capacity, uint32_t* dev_count, void* stream);

/* Measurement hooks (bench.py): device time of the fzb_match_list_device calls made on this matcher since
 * fzb_set_profiling(m, 1), measured with HIP events recorded on the launch stream (event records only, no
 * synchronisation until read).  fzb_last_timings averages over those calls (at most the last 32):
 * out_ms[0]=filter kernel (mean over the calls that ran one), [1]=whole pipeline, [2]=calls averaged, [3]=how many of them ran a filter kernel. */
int fzb_set_profiling(fzb_matcher* m, int enabled);
int fzb_last_timings(fzb_matcher* m, float out_ms[4]);
/* the same calls by stage: out_ms[0]=streaming filter kernel, [1]=compaction (+ lane-exact prefilter and its compaction), [2]=scorers,
 * [3]=whole pipeline, [4]=calls averaged, [5]=how many of them ran a streaming filter kernel (the filter figure is their mean) */
int fzb_last_stage_timings(fzb_matcher* m, float out_ms[6]);
/* counters of the last call: out[0]=survivors of the filter stage, [1]=kept by the lane-exact prefilter,
 * [2]=windows queued for the wave-per-haystack kernel's back queue (beyond 1024 bytes: the greedy fallback; unicode scorings outside the
 * thread-per-haystack kernels' preconditions), [3]=multi-chunk windows (one chunk < window <= 1024 bytes) */
int fzb_last_counters(fzb_matcher* m, uint32_t out[4]);

/* test hook, host only: the byte-level DFA of the unicode 0-typo prefilter run over one haystack (1 / 0), -1 if the matcher has none */
int fzb_debug_unicode_dfa_accepts(const fzb_matcher* m, const uint8_t* bytes, size_t len);

/* test hook, host only: the LCS automaton of a typo configuration (the streaming filter's accept test `LCS(needle, haystack) >= rows -
 * max_typos` as a DFA over the reachable bit-vector states) run over one haystack: 1 / 0, -1 if the matcher has none (0 typos, no
 * prefilter, more than 226 states); *out_states (optional) = its number of states */
int fzb_debug_lcs_dfa_accepts(const fzb_matcher* m, const uint8_t* bytes, size_t len, int32_t* out_states);

/* test hook, host only: the class-composite form of the matcher's streaming automaton (G byte transitions composed over the K byte
 * classes; the ragged filter's table) run over one haystack: 1 / 0 = accepts / rejects, -1 if the matcher has none; out_kg[0] = K, [1] = G */
int fzb_debug_cdfa_state(const fzb_matcher* m, const uint8_t* bytes, size_t len, int32_t* out_kg);

/* test hook: the library's environment switches (frizbee_amd/csrc/knobs.h - comparison and debugging only, parsed once on first use) are
 * read again.  Matchers created before the call keep what was decided when they were created. */
void fzb_debug_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif /* FRIZBEE_HIP_H */
