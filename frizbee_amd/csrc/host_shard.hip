// The multi-device form of the boundary: `Matcher::match_list_parallel` (src/matcher/parallel.rs:18-89) with the GPUs of one node in
// the role of its worker threads.  The reference cuts the list into contiguous chunks, gives every worker a global index offset
// (parallel.rs:55-63), sorts each worker's run (:66-76) and k-way merges the runs (:78-87).  Here a shard is one contiguous index range
// resident on one device.  Per query a persistent host thread per shard (hipSetDevice, its own stream and its own clone of the matcher,
// whose workspace lives on that device) runs the pipeline UNSORTED - the shard's records in index order - reads back the record count
// (8 bytes) and copies the run, device to device, to its place in one list on the ROOT device (the caller's current device): shard order
// is ascending index order, so the concatenation is exactly the list `match_list` orders, and the root runs the reverse / stable radix
// sort ONCE (kernels_sort.hip) and makes ONE copy to the host.  Same result as the reference's per-run sort + k-way merge, without a
// host heap (round 3 merged on the host: 17 ns per record on one thread, 67 ms for 8 x 500 k records).
// A Rust host binds exactly this - no Python, no torch.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "host_internal.h"

int fzb_corpus_upload_impl(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, uint64_t ends_base, fzb_corpus** out);

struct fzb_sharded_corpus {
    size_t n = 0;
    std::vector<u64> bounds;          // nshards + 1 haystack indices
    std::vector<int> device;          // device of shard g
    std::vector<fzb_corpus*> shard;   // resident on device[g]
};

namespace {
struct ThreadResult {
    int rc = FZB_OK;
    std::string err;
};
// run fn(g) on one host thread per shard; the first failure (lowest shard) becomes the calling thread's error
template <typename F>
int for_shards(size_t nshards, F fn) {
    std::vector<ThreadResult> res(nshards);
    auto body = [&](size_t g) {
        res[g].rc = fn(g);
        if (res[g].rc) res[g].err = fzb_last_error();  // the worker's thread-local message
    };
    std::vector<std::thread> pool;
    for (size_t g = 1; g < nshards; g++) pool.emplace_back(body, g);
    if (nshards) body(0);
    for (auto& th : pool) th.join();
    for (size_t g = 0; g < nshards; g++)
        if (res[g].rc) return fzb_fail(res[g].rc, "shard " + std::to_string(g) + ": " + res[g].err);
    return FZB_OK;
}

// The workers of the multi-device query: one persistent host thread per shard, owned by the matcher (the reference spawns its
// workers per call, src/matcher/parallel.rs:43-64 - for a 0.1 ms query that is most of the time).  A worker that has just finished
// a job polls for the next one for a short while before it sleeps, so a steady stream of queries never pays a wake-up.
class ShardWorkers {
public:
    explicit ShardWorkers(size_t n) : res_(n) {
        const unsigned hw = std::thread::hardware_concurrency();
        spin_ok_ = hw == 0 || n <= (size_t)hw / 2;
        for (size_t g = 1; g < n; g++) th_.emplace_back([this, g] { loop(g); });
    }
    ~ShardWorkers() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    size_t size() const { return res_.size(); }
    // fn(g) for g < n, shard 0 on the calling thread; the first failure (lowest shard) becomes the calling thread's error
    int run(size_t n, const std::function<int(size_t)>& fn) {
        if (!n) return FZB_OK;
        fn_ = &fn;
        n_ = n;
        pending_.store((int)th_.size(), std::memory_order_relaxed);  // every worker wakes; those beyond n have nothing to do
        {
            std::lock_guard<std::mutex> l(mu_);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        body(0);
        for (unsigned spin = 0; pending_.load(std::memory_order_acquire) > 0; spin++)
            if (spin > 4096) std::this_thread::yield();
        for (size_t g = 0; g < n; g++)
            if (res_[g].rc) return fzb_fail(res_[g].rc, "shard " + std::to_string(g) + ": " + res_[g].err);
        return FZB_OK;
    }

private:
    void body(size_t g) {
        res_[g].rc = (*fn_)(g);
        if (res_[g].rc) res_[g].err = fzb_last_error();  // the worker's thread-local message
        else res_[g].err.clear();
    }
    void loop(size_t g) {
        uint64_t seen = 0;
        for (;;) {
            uint64_t now;
            // (polling only pays while every worker has a core to itself: with more shards than half the host's hardware threads the
            // pollers would take the cores of the very workers being waited for - they go to sleep at once then)
            const unsigned spin_max = spin_ok_ ? 20000u : 0u;
            for (unsigned spin = 0; (now = gen_.load(std::memory_order_acquire)) == seen && spin < spin_max; spin++) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((spin & 1023u) == 1023u) std::this_thread::yield();
            }
            if (now == seen) {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                now = gen_.load(std::memory_order_acquire);
            }
            seen = now;
            if (stop_) return;
            if (g < n_) body(g);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::vector<ThreadResult> res_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> pending_{0};
    const std::function<int(size_t)>* fn_ = nullptr;
    size_t n_ = 0;
    bool stop_ = false;
    bool spin_ok_ = true;
};
}  // namespace

// Runs that the current device can read (its own memory; peers' memory once peer access is on) -> the ordered list on the host: one
// concatenation kernel that reads the run lengths on the device, `match_list`'s ordering, the count and then the records to the host.
static int merge_runs_on_device(fzb_matcher* m, const void* const* dev_runs, const uint32_t* const* dev_counts, const size_t* run_caps, size_t nruns, hipStream_t st, fzb_match** out,
                                size_t* out_len) {
    int rc;
    size_t cap = 0;
    for (size_t g = 0; g < nruns; g++) {
        if (run_caps[g] > 0xFFFFFFFFull) return fzb_fail(FZB_ERR_INVALID, "run capacity beyond the u32 index space");
        cap += run_caps[g];
    }
    if (cap > 0xFFFFFFFFull) return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(cap) + " > 4294967295 (index offset: 0)");
    if (!cap) return FZB_OK;
    if ((rc = fzb_ensure_out_staging(m, cap))) return rc;
    OrderPlan plan;
    if ((rc = fzb_order_begin(m, cap, m->out_dev, &plan))) return rc;
    // count_dev: two blocks of four words, alternating between batches of FZB_MAX_RUNS runs - [0] records written so far, [1] matches
    // found, [2] "a run was truncated by its producer" (sticky: every batch writes into the same word)
    u32* words = m->count_dev;  // (every word read below is assigned by the concatenation launches: no clearing fill in the stream)
    const u32* base = nullptr;
    u32* tot = words;
    for (size_t g0 = 0; g0 < nruns; g0 += FZB_MAX_RUNS) {
        RunSet rs{};
        rs.n = (int)std::min<size_t>(FZB_MAX_RUNS, nruns - g0);
        for (int k = 0; k < rs.n; k++) {
            rs.run[k] = (const fzb_match_rec*)dev_runs[g0 + (size_t)k];
            rs.count[k] = dev_counts[g0 + (size_t)k];
            rs.cap[k] = (u32)run_caps[g0 + (size_t)k];
        }
        fzb_launch_concat_runs(rs, base, tot, plan.in, (u32)cap, m->lc.num_cus * 2, words + 2, st);
        base = tot;
        tot = tot == words ? words + 4 : words;
    }
    HIPCHK(hipGetLastError());
    if ((rc = fzb_order_finish(m, plan, m->out_dev, base, st))) return rc;
    // one synchronisation for the counters and (speculatively, sized by the previous result) the records
    const fzb_match_rec* final_dev = (plan.reversed || plan.by_score) ? m->out_dev : plan.in;
    fzb_match* r = nullptr;
    size_t n = 0;
    if ((rc = fzb_fetch_records(m->fetch, final_dev, words, (int)(base - words), cap, st, &r, &n))) return rc;
    if (m->fetch.count_host[2]) {
        fzb_matches_free(r);
        return fzb_fail(FZB_ERR_CAPACITY, "a shard's run was truncated by its producer (more matches than its buffer holds): nothing merged");
    }
    *out = r;
    *out_len = n;
    return FZB_OK;
}

extern "C" {

int fzb_device_count(int* out) {
    if (!out) return fzb_fail(FZB_ERR_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = 0;
        return fzb_fail(FZB_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *out = n;
    return FZB_OK;
}

// Pure host arithmetic (no device): the contiguous index ranges of the shards, out_bounds[g] .. out_bounds[g+1].
//   by count: g * ceil(n / nshards) (what frizbee_amd.distributed.shard_range computes)
//   by bytes: shard g starts at the first haystack that STARTS at or after g/nshards of the total bytes (ragged lists: every device
//             streams about the same number of bytes; frizbee_amd.distributed.shard_ranges_by_bytes)
int fzb_shard_ranges(const uint64_t* end_offsets, size_t n, int nshards, int by_bytes, uint64_t* out_bounds) {
    if (!out_bounds || nshards < 1 || (n && by_bytes && !end_offsets)) return fzb_fail(FZB_ERR_INVALID, "bad argument");
    const u64 total = (n && by_bytes) ? end_offsets[n - 1] : 0;
    out_bounds[0] = 0;
    for (int g = 1; g < nshards; g++) {
        u64 cut;
        if (!by_bytes) {
            const u64 per = ((u64)n + (u64)nshards - 1) / (u64)nshards;
            cut = std::min<u64>((u64)g * per, n);
        } else {
            const u64 target = (u64)(((unsigned __int128)total * (unsigned)g) / (unsigned)nshards);
            // haystack i starts at end_offsets[i-1]: one past the first END that reaches the target
            cut = target ? (u64)(std::lower_bound(end_offsets, end_offsets + n, target) - end_offsets) + 1 : 0;
            cut = std::min<u64>(cut, n);
        }
        out_bounds[g] = std::max(cut, out_bounds[g - 1]);
    }
    out_bounds[nshards] = n;
    return FZB_OK;
}

int fzb_corpus_upload_sharded(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, int ndev, int flags, fzb_sharded_corpus** out) {
    if (!out || (n && (!bytes || !end_offsets))) return fzb_fail(FZB_ERR_INVALID, "null argument");
    if (ndev < 1) return fzb_fail(FZB_ERR_INVALID, "ndev must be positive");
    if (n > 0xFFFFFFFFull)
        return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(n) + " > 4294967295 (index offset: 0)");
    int have = 0;
    int rc = fzb_device_count(&have);
    if (rc) return rc;
    if (have < 1) return fzb_fail(FZB_ERR_HIP, "no HIP device");
    if (have < ndev && !(flags & FZB_SHARD_OVERSUBSCRIBE))
        return fzb_fail(FZB_ERR_HIP, std::to_string(ndev) + " devices asked for, " + std::to_string(have) + " visible (FZB_SHARD_OVERSUBSCRIBE lets shards share a device)");
    for (size_t i = 1; i < n; i++)  // the cut search below needs a sorted array; a decreasing offset is a caller error either way
        if (end_offsets[i] < end_offsets[i - 1]) return fzb_fail(FZB_ERR_INVALID, "end_offsets must be non-decreasing");
    auto sc = new fzb_sharded_corpus();
    sc->n = n;
    sc->bounds.assign((size_t)ndev + 1, 0);
    rc = fzb_shard_ranges(end_offsets, n, ndev, (flags & FZB_SHARD_BY_BYTES) ? 1 : 0, sc->bounds.data());
    if (rc) { delete sc; return rc; }
    sc->device.resize((size_t)ndev);
    sc->shard.assign((size_t)ndev, nullptr);
    for (int g = 0; g < ndev; g++) sc->device[(size_t)g] = g % have;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;  // shard 0 is uploaded on the calling thread: its current device is restored below
    rc = for_shards((size_t)ndev, [&](size_t g) -> int {
        HIPCHK(hipSetDevice(sc->device[g]));
        const u64 lo = sc->bounds[g], hi = sc->bounds[g + 1];
        const u64 base = lo ? end_offsets[lo - 1] : 0;
        return fzb_corpus_upload_impl(bytes ? bytes + base : nullptr, end_offsets ? end_offsets + lo : nullptr, (size_t)(hi - lo), base, &sc->shard[g]);
    });
    if (have_cur) (void)hipSetDevice(cur);
    if (rc) {
        const std::string msg = fzb_last_error();
        fzb_sharded_corpus_free(sc);
        return fzb_fail(rc, msg);
    }
    *out = sc;
    return FZB_OK;
}

void fzb_sharded_corpus_free(fzb_sharded_corpus* sc) {
    if (!sc) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (size_t g = 0; g < sc->shard.size(); g++)
        if (sc->shard[g]) {
            (void)hipSetDevice(sc->device[g]);
            fzb_corpus_free(sc->shard[g]);
        }
    if (have_cur) (void)hipSetDevice(cur);
    delete sc;
}

size_t fzb_sharded_corpus_len(const fzb_sharded_corpus* sc) { return sc ? sc->n : 0; }
int fzb_sharded_corpus_shards(const fzb_sharded_corpus* sc) { return sc ? (int)sc->shard.size() : 0; }
int fzb_sharded_corpus_shard(const fzb_sharded_corpus* sc, int g, uint64_t* lo, uint64_t* hi, int* device) {
    if (!sc || g < 0 || (size_t)g >= sc->shard.size()) return fzb_fail(FZB_ERR_INVALID, "no such shard");
    if (lo) *lo = sc->bounds[(size_t)g];
    if (hi) *hi = sc->bounds[(size_t)g + 1];
    if (device) *device = sc->device[(size_t)g];
    return FZB_OK;
}

int fzb_match_list_parallel_sharded(fzb_matcher* m, const fzb_sharded_corpus* sc, fzb_match** out, size_t* out_len) {
    if (!m || !sc || !out || !out_len) return fzb_fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    *out_len = 0;
    const size_t ns = sc->shard.size();
    const int sort = m->config.sort;
    const bool reversed = sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;
    if (m->empty) {  // CompiledPatterns::Empty: every index, score 0, reversed if the strategy says so, never sorted (mod.rs:215-220, 381-384)
        fzb_match* r = (fzb_match*)malloc(std::max<size_t>(sc->n, 1) * sizeof(fzb_match));
        if (!r) return fzb_fail(FZB_ERR_INVALID, "out of memory");
        for (size_t i = 0; i < sc->n; i++) r[i] = fzb_match{(uint32_t)(reversed ? sc->n - 1 - i : i), 0, 0, 0};
        *out = r;
        *out_len = sc->n;
        return FZB_OK;
    }
    // The ROOT: the device that is current on the calling thread.  It receives every shard's run and orders the whole list once.
    int rc = fzb_bind_device(m);
    if (rc) return rc;
    const int root = m->device;
    if (!m->shard_stream) HIPCHK(hipStreamCreateWithFlags(&m->shard_stream, hipStreamNonBlocking));
    if (sc->n == 0) return FZB_OK;
    if ((rc = fzb_ensure_out_staging(m, sc->n))) return rc;
    OrderPlan plan;
    if ((rc = fzb_order_begin(m, sc->n, m->out_dev, &plan))) return rc;
    fzb_match_rec* const gather = plan.in;  // the concatenation of the runs (the sort's second buffer when one radix pass orders it)
    // one clone of the matcher per shard (host work only; its device state is created by the shard's worker on the shard's device and
    // kept across queries and across fzb_matcher_set_pattern / set_config).  A clone follows its shard to another device: its device
    // state is released where it lives and built again on first use.
    while (m->shard_clones.size() < ns) {
        fzb_matcher* cm = nullptr;
        if ((rc = fzb_matcher_clone(m, &cm))) return rc;
        m->shard_clones.push_back(cm);
    }
    for (size_t g = 0; g < ns; g++) {
        fzb_matcher*& cm = m->shard_clones[g];
        if (cm->shard_device >= 0 && cm->shard_device != sc->device[g]) {
            (void)hipSetDevice(cm->shard_device);
            fzb_matcher_free(cm);
            cm = nullptr;
            (void)hipSetDevice(root);
            if ((rc = fzb_matcher_clone(m, &cm))) {
                m->shard_clones.resize(g);  // the clones behind this one were not touched; they are rebuilt on the next call
                return rc;
            }
        }
    }
    // Peer access, once per (root, device) pair: with it hipMemcpyPeerAsync moves a run device to device over xGMI (SDMA); without it the
    // runtime stages the copy through host memory - still correct, several times slower, and reported (fzb_matcher_shard_report), never
    // silent.  hipDeviceEnablePeerAccess is per direction and per current device: both directions are asked for.
    for (size_t g = 0; g < ns; g++) {
        const int d = sc->device[g];
        if (d == root) continue;
        bool known = false;
        for (const auto& pr : m->shard_peers) known = known || (pr[0] == root && pr[1] == d);
        if (known) continue;
        int can_rd = 0, can_dr = 0;
        if (hipDeviceCanAccessPeer(&can_rd, root, d) != hipSuccess) can_rd = 0;
        if (hipDeviceCanAccessPeer(&can_dr, d, root) != hipSuccess) can_dr = 0;
        int state = 0;
        if (can_rd && can_dr) {
            hipError_t e1 = hipSetDevice(root) == hipSuccess ? hipDeviceEnablePeerAccess(d, 0) : hipErrorInvalidDevice;
            hipError_t e2 = hipSetDevice(d) == hipSuccess ? hipDeviceEnablePeerAccess(root, 0) : hipErrorInvalidDevice;
            state = ((e1 == hipSuccess || e1 == hipErrorPeerAccessAlreadyEnabled) && (e2 == hipSuccess || e2 == hipErrorPeerAccessAlreadyEnabled)) ? 1 : 0;
        }
        (void)hipGetLastError();  // a refusal is a state, not an error of this query
        (void)hipSetDevice(root);
        m->shard_peers.push_back({root, d, state});
    }
    // How the runs reach the root.  PULL (every shard lives on the root device - one GPU holding several shards): the workers only enqueue
    // their pipelines; the root's stream waits for them and ONE kernel concatenates the runs, reading their lengths on the device - no
    // host round trip before the final list.  COPY (shards on other devices): each worker reads its count back (8 bytes) and copies its
    // run to its place with hipMemcpyPeerAsync as soon as the counts below it are known.  FZB_SHARD_GATHER=copy forces the second form
    // (how it is tested on one GPU).
    bool pull = !fzb_knobs().shard_gather_copy;
    for (size_t g = 0; g < ns; g++) pull = pull && sc->device[g] == root;
    const int inline_mode = fzb_knobs().shard_inline;  // -1 = decide here, 0 = always the workers, 1 = always the calling thread (pull form only)
    const bool inline_enqueue = pull && (inline_mode == 1 || inline_mode < 0);
    // A single shard runs on the root's own stream (no event to record and wait for: 0.252 -> 0.240 ms).  SEVERAL shards keep a stream each:
    // their stages are launch-bound kernels of a few microseconds, and one behind the other on ONE stream eight pipelines take 0.40 ms where
    // eight streams take 0.31 (measured; the device overlaps the small kernels' latencies).
    const bool one_stream = inline_enqueue && ns == 1;
    // counts[g]: shard g's number of records, published by its worker as soon as it is known (-1 before); worker g starts its copy
    // when the counts of the shards below it are in - the prefix is where its run starts in the gathered list
    std::vector<std::atomic<int64_t>> counts(ns);
    for (auto& c : counts) c.store(-1, std::memory_order_relaxed);
    std::vector<u8> copied(ns, 0);
    auto shard_job = [&](size_t g) -> int {
        struct Publish {  // whatever happens, the workers above must not wait for this one
            std::atomic<int64_t>& slot;
            ~Publish() { if (slot.load(std::memory_order_relaxed) < 0) slot.store(0, std::memory_order_release); }
        } publish{counts[g]};
        const fzb_corpus* c = sc->shard[g];
        const size_t count = (size_t)c->dev.n;
        HIPCHK(hipSetDevice(sc->device[g]));
        fzb_matcher* cm = m->shard_clones[g];
        if (cm->shard_device < 0) {
            HIPCHK(hipStreamCreateWithFlags(&cm->shard_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&cm->shard_event, hipEventDisableTiming));
            HIPCHK(hipHostMalloc((void**)&cm->shard_count_host, 16, hipHostMallocDefault));
            cm->shard_device = sc->device[g];
        }
        int rc_ = fzb_ensure_out_staging(cm, count);
        if (rc_) return rc_;
        // the shard's records in INDEX order, numbered from the shard's first index (what a worker of match_list_parallel pushes,
        // parallel.rs:55-63) - unsorted: the root orders the whole list
        rc_ = fzb_match_list_device(cm, c, 0, count, (uint32_t)sc->bounds[g], (fzb_match*)cm->out_dev, cm->out_cap, cm->count_dev, one_stream ? m->shard_stream : cm->shard_stream);
        if (rc_) return rc_;
        if (one_stream) return FZB_OK;  // the concatenation follows on the same stream
        if (pull) {
            HIPCHK(hipEventRecord(cm->shard_event, cm->shard_stream));
            copied[g] = 1;
            return FZB_OK;
        }
        if (!count) return FZB_OK;
        HIPCHK(hipMemcpyAsync(cm->shard_count_host, cm->count_dev, 8, hipMemcpyDeviceToHost, cm->shard_stream));
        HIPCHK(hipStreamSynchronize(cm->shard_stream));
        const u32 cnt = cm->shard_count_host[0];
        counts[g].store((int64_t)cnt, std::memory_order_release);
        if (!cnt) return FZB_OK;
        size_t at = 0;
        for (size_t k = 0; k < g; k++) {
            int64_t v;
            for (unsigned spin = 0; (v = counts[k].load(std::memory_order_acquire)) < 0; spin++)
                if (spin > 4096) std::this_thread::yield();
            at += (size_t)v;
        }
        // the run goes to its place in the root's list: device to device over xGMI (or inside the one device the shards share)
        if (sc->device[g] == root) HIPCHK(hipMemcpyAsync(gather + at, cm->out_dev, (size_t)cnt * sizeof(fzb_match_rec), hipMemcpyDeviceToDevice, cm->shard_stream));
        else HIPCHK(hipMemcpyPeerAsync(gather + at, root, cm->out_dev, sc->device[g], (size_t)cnt * sizeof(fzb_match_rec), cm->shard_stream));
        HIPCHK(hipEventRecord(cm->shard_event, cm->shard_stream));
        copied[g] = 1;
        return FZB_OK;
    };
    // Shards that share the root device are enqueued by the CALLING thread, one after the other: launches from several host threads onto
    // one device serialise inside the runtime anyway (measured with 8 shards on one GPU: 0.35 ms through the workers, see bench.py
    // `sharded`), and nothing waits between them in the pull form.  Shards on other devices go through their workers.
    if (inline_enqueue) {
        rc = FZB_OK;
        for (size_t g = 0; g < ns && !rc; g++) {
            rc = shard_job(g);
            if (rc) rc = fzb_fail(rc, "shard " + std::to_string(g) + ": " + fzb_last_error());
        }
    } else {
        ShardWorkers* pool = (ShardWorkers*)m->shard_workers;
        if (!pool || pool->size() < ns) {
            delete pool;
            m->shard_workers = pool = new ShardWorkers(ns);
        }
        rc = pool->run(ns, shard_job);
    }
    (void)hipSetDevice(root);
    // Every failure from here on leaves through `drained`: the copies (and pipelines) the shards have in flight write into this matcher's
    // buffers, so they are waited for before the caller can retry or free anything (the message of the failure is kept)
    auto drained = [&](int code) -> int {
        const std::string msg = fzb_last_error();
        for (size_t g = 0; g < ns; g++)
            if (copied[g]) (void)hipEventSynchronize(m->shard_clones[g]->shard_event);
        (void)hipStreamSynchronize(m->shard_stream);
        (void)hipGetLastError();
        return fzb_fail(code, msg);
    };
    if (rc) return drained(rc);
    {   // how every run travelled (fzb_matcher_shard_report)
        std::string rep = "root device " + std::to_string(root) + "; gather " + (pull ? "pull (one concatenation kernel reads the runs)" : "copy (count to the host, run copied to its place)");
        for (size_t g = 0; g < ns; g++) {
            const int d = sc->device[g];
            rep += "; shard " + std::to_string(g) + " on device " + std::to_string(d) + ": ";
            if (d == root) { rep += "same device"; continue; }
            int state = 0;
            for (const auto& pr : m->shard_peers)
                if (pr[0] == root && pr[1] == d) state = pr[2];
            rep += state ? "peer access enabled (device to device over xGMI)" : "peer access REFUSED by the runtime (hipMemcpyPeerAsync stages the run through host memory)";
        }
        m->shard_report = rep;
    }
    size_t total = 0;
    u32 agg[4] = {0, 0, 0, 0};
    for (size_t g = 0; g < ns; g++) {
        total += (size_t)std::max<int64_t>(counts[g].load(std::memory_order_acquire), 0);
        if (copied[g]) {
            hipError_t e_ = hipStreamWaitEvent(m->shard_stream, m->shard_clones[g]->shard_event, 0);
            if (e_ != hipSuccess) return drained(fzb_fail(FZB_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e_)));
        }
        for (int k = 0; k < 4; k++) agg[k] += m->shard_clones[g]->last_counters[k];
    }
    memcpy(m->last_counters, agg, sizeof(agg));  // fzb_last_counters on the parent = the sum over the shards
    if (pull) {
        std::vector<const void*> runs(ns);
        std::vector<const uint32_t*> cnts(ns);
        std::vector<size_t> caps(ns);
        for (size_t g = 0; g < ns; g++) {
            runs[g] = m->shard_clones[g]->out_dev;
            cnts[g] = m->shard_clones[g]->count_dev;
            caps[g] = (size_t)sc->shard[g]->dev.n;
        }
        rc = merge_runs_on_device(m, runs.data(), cnts.data(), caps.data(), ns, m->shard_stream, out, out_len);
        return rc ? drained(rc) : FZB_OK;
    }
    if (!total) return FZB_OK;
    // the whole list's count -> device memory (the sort reads it there), reverse / stable radix sort ONCE, one copy to the host
    {
        hipError_t e_ = hipMemsetD32Async((hipDeviceptr_t)m->count_dev, (int)(u32)total, 1, m->shard_stream);
        if (e_ != hipSuccess) return drained(fzb_fail(FZB_ERR_HIP, std::string("hipMemsetD32Async: ") + hipGetErrorString(e_)));
    }
    if ((rc = fzb_order_finish(m, plan, m->out_dev, m->count_dev, m->shard_stream))) return drained(rc);
    fzb_match* r = (fzb_match*)fzb_pinned_get(total * sizeof(fzb_match));
    if (!r) return drained(fzb_fail(FZB_ERR_HIP, "hipHostMalloc failed for the result list"));
    const fzb_match_rec* final_dev = (plan.reversed || plan.by_score) ? m->out_dev : gather;
    hipError_t e = hipMemcpyAsync(r, final_dev, total * sizeof(fzb_match), hipMemcpyDeviceToHost, m->shard_stream);
    if (e == hipSuccess) e = fzb_stream_wait(m->shard_stream);
    if (e != hipSuccess) {
        fzb_pinned_put(r);
        return fzb_fail(FZB_ERR_HIP, std::string("device to host: ") + hipGetErrorString(e));
    }
    *out = r;
    *out_len = total;
    return FZB_OK;
}

const char* fzb_matcher_shard_report(const fzb_matcher* m) { return m ? m->shard_report.c_str() : ""; }

// The same combine for runs that are already on ONE device (frizbee_amd.distributed: the root rank after the RCCL gather of the
// per-rank buffers): concatenation in run order -> reverse / stable radix sort -> one copy to the host.  dev_counts[g] points at run
// g's record count in device memory, so nothing is read back before the final list.
int fzb_merge_shard_runs(fzb_matcher* m, const void* const* dev_runs, const uint32_t* const* dev_counts, const size_t* run_caps, size_t nruns, void* stream, fzb_match** out,
                         size_t* out_len) {
    if (!m || !out || !out_len || (nruns && (!dev_runs || !dev_counts || !run_caps))) return fzb_fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    *out_len = 0;
    if (!nruns) return FZB_OK;
    int rc = fzb_bind_device(m);
    if (rc) return rc;
    return merge_runs_on_device(m, dev_runs, dev_counts, run_caps, nruns, (hipStream_t)stream, out, out_len);
}

}  // extern "C"

void fzb_shard_workers_free(void* p) { delete (ShardWorkers*)p; }
