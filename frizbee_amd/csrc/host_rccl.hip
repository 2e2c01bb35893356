// One process per GPU below the C ABI: `match_list_parallel` (src/matcher/parallel.rs:18-89) over a list whose shards live in DIFFERENT
// processes, the per-shard runs exchanged by RCCL over xGMI (BASELINE.json north_star: "the haystack list shards trivially across the 8
// GPUs of one node with an RCCL all-gather ... of the per-shard (score, index)").  host_shard.hip is the single-process form (peer
// copies); frizbee_amd.distributed.ShardExchange the torch.distributed form the bench uses.  This file is what a Rust host that runs one
// process per GPU binds: nothing here knows about Python.
//
// RCCL is opened at run time (dlopen of librccl.so.1, first use): the library itself has no load-time dependency on it, a process that
// already holds an RCCL (torch's) gets that same copy by its soname, and a box without RCCL fails LOUDLY in fzb_rccl_unique_id /
// fzb_shard_comm_create, never silently.
//
// The exchange of one query (all on the communicator's stream, one host synchronisation - the counts - before the records move):
//   1. the pipeline writes this rank's index-ordered run and its two count words into the communicator's buffers (capacity = the shard's
//      item count: a run can never be truncated);
//   2. ncclAllGather of the count words (8 bytes per rank) -> every rank knows every run's length;
//   3. ONE RCCL group: every receiver posts ncclRecv of exactly count[r] records from every other rank r, every other rank ncclSend of
//      exactly its run to every receiver (receivers = the root, or every rank with FZB_GATHER_ALL = the all-gather of north_star); the
//      receiver's own run stays where the pipeline wrote it;
//   4. receivers: concatenation in rank order (= ascending index order: rank g's shard is a contiguous index range) + `match_list`'s
//      ordering ONCE on the device + one copy to the host (merge_runs_on_device's steps, through fzb_merge_shard_runs).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "host_internal.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // FZB_RCCL_LIB: another path or soname (debugging aid; read here, once, like the knobs)
        const char* names[] = {getenv("FZB_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            const char* e = dlerror();
            api.error += std::string(api.error.empty() ? "" : "; ") + nm + ": " + (e ? e : "dlopen failed");
        }
        if (!api.handle) return;
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(api.handle, name);
            if (!p) { ok = false; api.error += std::string(api.error.empty() ? "" : "; ") + "missing symbol " + name; }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return api;
}

int need_rccl(RcclApi** out) {
    RcclApi& a = rccl_api();
    if (!a.handle) return fzb_fail(FZB_ERR_HIP, "RCCL is not available in this process (" + a.error + "): the multi-process exchange has no other transport");
    *out = &a;
    return FZB_OK;
}

}  // namespace

#define NCCLCHK(api, expr)                                                                                               \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) return fzb_fail(FZB_ERR_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r_));      \
    } while (0)

struct fzb_shard_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = -1;
    hipStream_t stream = nullptr;
    fzb_match_rec* run = nullptr;     // this rank's run (index-ordered records of its shard)
    size_t run_cap = 0;
    fzb_match_rec* gather = nullptr;  // the other ranks' runs, back to back in rank order (receivers only)
    size_t gather_cap = 0;
    u32* words = nullptr;             // device: [0..1] this rank's count words, [2 .. 2 + 2 world) every rank's (the all-gather's output)
    u32* words_host = nullptr;        // page-locked copy of the gathered words
    u64 bytes_sent = 0, bytes_received = 0;  // records of the last query, as bytes
};

extern "C" {

int fzb_rccl_unique_id(uint8_t out_id[FZB_RCCL_ID_BYTES]) {
    static_assert(FZB_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as the bytes of ncclUniqueId");
    if (!out_id) return fzb_fail(FZB_ERR_INVALID, "null argument");
    RcclApi* api;
    int rc = need_rccl(&api);
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(api, api->GetUniqueId(&id));
    memcpy(out_id, id.internal, FZB_RCCL_ID_BYTES);
    return FZB_OK;
}

int fzb_shard_comm_create(const uint8_t id[FZB_RCCL_ID_BYTES], int rank, int world, fzb_shard_comm** out) {
    if (!id || !out) return fzb_fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fzb_fail(FZB_ERR_INVALID, "rank " + std::to_string(rank) + " outside a world of " + std::to_string(world));
    RcclApi* api;
    int rc = need_rccl(&api);
    if (rc) return rc;
    fzb_shard_comm* c = new fzb_shard_comm;
    c->rank = rank;
    c->world = world;
    auto fail_free = [&](int code) { fzb_shard_comm_free(c); return code; };
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) return fail_free(fzb_fail(FZB_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e)));
    ncclUniqueId uid;
    memcpy(uid.internal, id, FZB_RCCL_ID_BYTES);
    ncclResult_t r = api->CommInitRank(&c->comm, world, uid, rank);  // collective: every rank of the world is inside this call
    if (r != ncclSuccess) { c->comm = nullptr; return fail_free(fzb_fail(FZB_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r))); }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { c->stream = nullptr; return fail_free(fzb_fail(FZB_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e))); }
    const size_t nwords = 2 + 2 * (size_t)world;
    if ((e = fzb_dev_alloc((void**)&c->words, nwords * sizeof(u32))) != hipSuccess) { c->words = nullptr; return fail_free(fzb_fail(FZB_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e))); }
    c->words_host = (u32*)fzb_pinned_get(nwords * sizeof(u32));
    if (!c->words_host) return fail_free(fzb_fail(FZB_ERR_HIP, "hipHostMalloc failed for the count words"));
    *out = c;
    return FZB_OK;
}

void fzb_shard_comm_free(fzb_shard_comm* c) {
    if (!c) return;
    int prev = -1;
    const bool switched = c->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != c->device && hipSetDevice(c->device) == hipSuccess;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl_api().handle) (void)rccl_api().CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->run) (void)hipFree(c->run);
    if (c->gather) (void)hipFree(c->gather);
    if (c->words) (void)hipFree(c->words);
    if (c->words_host) fzb_pinned_put(c->words_host);
    if (switched) (void)hipSetDevice(prev);
    delete c;
}

int fzb_shard_comm_rank(const fzb_shard_comm* c) { return c ? c->rank : -1; }
int fzb_shard_comm_world(const fzb_shard_comm* c) { return c ? c->world : 0; }

int fzb_shard_comm_last_exchange(const fzb_shard_comm* c, uint64_t out_bytes[2]) {
    if (!c || !out_bytes) return fzb_fail(FZB_ERR_INVALID, "null argument");
    out_bytes[0] = c->bytes_sent;
    out_bytes[1] = c->bytes_received;
    return FZB_OK;
}

static int grow(fzb_match_rec** p, size_t* cap, size_t want) {
    if (*p && *cap >= want) return FZB_OK;  // (a buffer exists even for an empty run: the pipeline and RCCL are handed real addresses)
    if (*p) HIPCHK(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    const size_t n = want + want / 8 + 1024;
    HIPCHK(fzb_dev_alloc((void**)p, n * sizeof(fzb_match_rec)));
    *cap = n;
    return FZB_OK;
}

// CompiledPatterns::Empty (src/matcher/mod.rs:194-196, 215-220, 381-384): every index of every share, score 0, reversed if the strategy says
// so, never sorted.  Nothing is scored; the ranks only tell each other (share length, first global index) and a receiver writes the list.
static int empty_pattern_list(RcclApi* api, fzb_matcher* m, size_t n, uint32_t index_offset, fzb_shard_comm* c, bool receiver, fzb_match** out, size_t* out_len) {
    if ((u64)n + (u64)index_offset > 0xFFFFFFFFull)
        return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string((u64)n + index_offset) + " > 4294967295 (index offset: " + std::to_string(index_offset) + ")");
    c->words_host[0] = (u32)n;
    c->words_host[1] = index_offset;
    HIPCHK(hipMemcpyAsync(c->words, c->words_host, 2 * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(api, api->AllGather(c->words, c->words + 2, 2, ncclUint32, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(c->words_host, c->words + 2, 2 * (size_t)c->world * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->bytes_sent = c->bytes_received = 0;
    if (!receiver) return FZB_OK;
    size_t total = 0;
    for (int r = 0; r < c->world; r++) total += c->words_host[2 * r];
    if (total > 0xFFFFFFFFull) return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(total) + " > 4294967295 (index offset: 0)");
    fzb_match* list = (fzb_match*)malloc(std::max<size_t>(total, 1) * sizeof(fzb_match));
    if (!list) return fzb_fail(FZB_ERR_INVALID, "out of memory");
    const int sort = m->config.sort;
    const bool reversed = sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;
    size_t k = 0;
    for (int r = 0; r < c->world; r++)
        for (u32 i = 0; i < c->words_host[2 * r]; i++, k++) list[reversed ? total - 1 - k : k] = fzb_match{c->words_host[2 * r + 1] + i, 0, 0, 0};
    *out = list;
    *out_len = total;
    return FZB_OK;
}

int fzb_match_list_parallel_rccl(fzb_matcher* m, const fzb_corpus* shard, uint32_t index_offset, fzb_shard_comm* c, int flags, fzb_match** out, size_t* out_len) {
    if (!m || !shard || !c || !out || !out_len) return fzb_fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    *out_len = 0;
    if (flags & ~FZB_GATHER_ALL) return fzb_fail(FZB_ERR_INVALID, "unknown flag");
    RcclApi* api;
    int rc = need_rccl(&api);
    if (rc) return rc;
    int dev = -1;
    HIPCHK(hipGetDevice(&dev));
    if (dev != c->device) return fzb_fail(FZB_ERR_INVALID, "the communicator was created on device " + std::to_string(c->device) + " but device " + std::to_string(dev) + " is current");
    const bool all = (flags & FZB_GATHER_ALL) != 0;
    const bool receiver = all || c->rank == 0;
    const size_t n = shard->dev.n;
    if (m->empty) return empty_pattern_list(api, m, n, index_offset, c, receiver, out, out_len);
    // 1. this rank's run
    if ((rc = grow(&c->run, &c->run_cap, n))) return rc;
    if ((rc = fzb_match_list_device(m, shard, 0, n, index_offset, (fzb_match*)c->run, n ? n : 1, c->words, c->stream))) return rc;
    // 2. every run's length
    u32* all_words = c->words + 2;
    NCCLCHK(api, api->AllGather(c->words, all_words, 2, ncclUint32, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(c->words_host, all_words, 2 * (size_t)c->world * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(fzb_stream_wait(c->stream));
    size_t others = 0, total = 0;
    for (int r = 0; r < c->world; r++) {
        const u32 written = c->words_host[2 * r], found = c->words_host[2 * r + 1];
        if (found != written) return fzb_fail(FZB_ERR_CAPACITY, "rank " + std::to_string(r) + "'s run was truncated (" + std::to_string(found) + " matches, " + std::to_string(written) + " records)");
        total += written;
        if (r != c->rank) others += written;
    }
    if (total > 0xFFFFFFFFull) return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(total) + " > 4294967295 (index offset: 0)");
    // 3. the records, each run exactly as long as it is
    if (receiver && (rc = grow(&c->gather, &c->gather_cap, others))) return rc;
    const size_t mine = c->words_host[2 * c->rank];
    c->bytes_sent = c->bytes_received = 0;
    if (c->world > 1) {
        NCCLCHK(api, api->GroupStart());
        size_t off = 0;
        for (int r = 0; r < c->world; r++) {
            if (r == c->rank) continue;
            const size_t cnt = c->words_host[2 * r];
            if (receiver && cnt) {
                NCCLCHK(api, api->Recv(c->gather + off, cnt * sizeof(fzb_match_rec), ncclUint8, r, c->comm, c->stream));
                c->bytes_received += cnt * sizeof(fzb_match_rec);
            }
            off += cnt;
            if ((all || r == 0) && mine) {
                NCCLCHK(api, api->Send(c->run, mine * sizeof(fzb_match_rec), ncclUint8, r, c->comm, c->stream));
                c->bytes_sent += mine * sizeof(fzb_match_rec);
            }
        }
        NCCLCHK(api, api->GroupEnd());
    }
    if (!receiver) {
        HIPCHK(hipStreamSynchronize(c->stream));  // the run may be overwritten by the next query only after it has left
        return FZB_OK;
    }
    // 4. rank order = ascending index order: concatenate, order once, one copy to the host
    std::vector<const void*> runs((size_t)c->world);
    std::vector<const uint32_t*> counts((size_t)c->world);
    std::vector<size_t> caps((size_t)c->world);
    size_t off = 0;
    for (int r = 0; r < c->world; r++) {
        const size_t cnt = c->words_host[2 * r];
        runs[(size_t)r] = r == c->rank ? (const void*)c->run : (const void*)(c->gather + off);
        if (r != c->rank) off += cnt;
        counts[(size_t)r] = all_words + 2 * r;
        caps[(size_t)r] = cnt;
    }
    return fzb_merge_shard_runs(m, runs.data(), counts.data(), caps.data(), (size_t)c->world, c->stream, out, out_len);
}

}  // extern "C"
