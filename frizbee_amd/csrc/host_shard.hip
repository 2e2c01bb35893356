// The multi-device form of the boundary: `Matcher::match_list_parallel` (src/matcher/parallel.rs:18-89) with the GPUs of one node in
// the role of its worker threads.  The reference cuts the list into contiguous chunks, gives every worker a global index offset
// (parallel.rs:55-63), sorts each worker's run (:66-76) and k-way merges the runs (:78-87).  Here a shard is one contiguous index range
// resident on one device; per query one host thread per shard (hipSetDevice, its own stream and its own clone of the matcher, whose
// workspace lives on that device) runs pipeline + device reverse / radix sort and copies its ordered run to the host; the calling
// thread merges the runs with fzb_k_merge_matches' tournament.  No device-to-device traffic: only the host consumes the result
// (north_star: "(score,index) top-k feeding radix_sort_matches on the host").  A Rust host binds exactly this - no Python, no torch.
#include <algorithm>
#include <cstdlib>
#include <cstring>
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
}  // namespace

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
    rc = for_shards((size_t)ndev, [&](size_t g) -> int {
        HIPCHK(hipSetDevice(sc->device[g]));
        const u64 lo = sc->bounds[g], hi = sc->bounds[g + 1];
        const u64 base = lo ? end_offsets[lo - 1] : 0;
        return fzb_corpus_upload_impl(bytes ? bytes + base : nullptr, end_offsets ? end_offsets + lo : nullptr, (size_t)(hi - lo), base, &sc->shard[g]);
    });
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
    // one clone of the matcher per shard (host work only; its device state is created by the shard's thread on the shard's device and
    // kept across queries and across fzb_matcher_set_pattern / set_config)
    while (m->shard_clones.size() < ns) {
        fzb_matcher* cm = nullptr;
        int rc = fzb_matcher_clone(m, &cm);
        if (rc) return rc;
        m->shard_clones.push_back(cm);
    }
    std::vector<fzb_match*> runs(ns, nullptr);
    std::vector<size_t> lens(ns, 0);
    int cur = 0;
    HIPCHK(hipGetDevice(&cur));
    int rc = for_shards(ns, [&](size_t g) -> int {
        const fzb_corpus* c = sc->shard[g];
        const size_t count = (size_t)c->dev.n;
        if (!count) return FZB_OK;
        HIPCHK(hipSetDevice(sc->device[g]));
        fzb_matcher* cm = m->shard_clones[g];
        if (cm->shard_device != sc->device[g]) {
            if (cm->shard_device >= 0) return fzb_fail(FZB_ERR_INVALID, "matcher was used with another sharded corpus whose shard " + std::to_string(g) + " lives on a different device");
            HIPCHK(hipStreamCreateWithFlags(&cm->shard_stream, hipStreamNonBlocking));
            cm->shard_device = sc->device[g];
        }
        int rc_ = fzb_ensure_out_staging(cm, count);
        if (rc_) return rc_;
        rc_ = fzb_sorted_range_device(cm, c, 0, count, (uint32_t)sc->bounds[g], (fzb_match*)cm->out_dev, cm->out_cap, cm->count_dev, cm->shard_stream);
        if (rc_) return rc_;
        u32 cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, cm->count_dev, 8, hipMemcpyDeviceToHost, cm->shard_stream));
        HIPCHK(hipStreamSynchronize(cm->shard_stream));
        fzb_match* r = (fzb_match*)fzb_pinned_get(std::max<size_t>(cnt[0], 1) * sizeof(fzb_match));
        if (!r) return fzb_fail(FZB_ERR_HIP, "hipHostMalloc failed for a shard's run");
        if (cnt[0]) {
            hipError_t e = hipMemcpyAsync(r, cm->out_dev, (size_t)cnt[0] * sizeof(fzb_match), hipMemcpyDeviceToHost, cm->shard_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(cm->shard_stream);
            if (e != hipSuccess) {
                fzb_pinned_put(r);
                return fzb_fail(FZB_ERR_HIP, std::string("device to host: ") + hipGetErrorString(e));
            }
        }
        runs[g] = r;
        lens[g] = cnt[0];
        return FZB_OK;
    });
    (void)hipSetDevice(cur);
    size_t total = 0;
    for (size_t g = 0; g < ns; g++) total += lens[g];
    fzb_match* merged = nullptr;
    if (!rc) {
        merged = (fzb_match*)malloc(std::max<size_t>(total, 1) * sizeof(fzb_match));
        if (!merged) rc = fzb_fail(FZB_ERR_INVALID, "out of memory");
    }
    // k_merge_matches_by_* over the per-shard runs (parallel.rs:78-87).  Every run is ordered per `sort` and the shards are contiguous,
    // ascending index ranges, so the merge is a concatenation for the index orders; the score orders take the tournament.
    if (!rc) rc = fzb_k_merge_runs(sort, runs.data(), lens.data(), ns, merged);
    for (size_t g = 0; g < ns; g++)
        if (runs[g]) fzb_pinned_put(runs[g]);
    if (rc) {
        free(merged);
        return rc;
    }
    *out = merged;
    *out_len = total;
    return FZB_OK;
}

}  // extern "C"
