// TEST DOUBLE, never shipped: the nine RCCL entry points csrc/host_rccl.hip opens with dlopen, for "ranks" that are THREADS of one
// process sharing one GPU (RCCL itself refuses two ranks on one device, and the boxes this is built on have one).  Loaded through
// FZB_RCCL_LIB by tests/test_gpu_sharded.py, so that the exchange logic of fzb_match_list_parallel_rccl - the count all-gather, which
// rank posts which send / receive, at which offset, of how many bytes, the merge of the runs in rank order - runs with a world of 2-8
// on the GPU box.  Semantics: every call is synchronous (the caller's stream is drained first); the all-gather meets at a barrier, sends
// and receives meet pairwise in a (source, destination) mailbox and are copied device to device by the receiver; a receive whose send
// has another size, or a counterpart that does not show up within 30 s, is an ERROR - which is what corruption or a hang would be with
// the real library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct Op {
    bool send;
    const void* sbuf;
    void* rbuf;
    size_t bytes;
    int peer;
    bool matched = false;
};

struct World {
    int n = 0, joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;
    std::map<std::pair<int, int>, std::vector<Op>> mail;  // (source, destination) -> sends in flight
    std::vector<const void*> ag_send;       // per rank: the all-gather's source
    bool failed = false;
    // false = a rank did not arrive within 30 s (it failed before the collective): an error for everyone who waited, not a hang
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = generation;
        if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); return true; }
        if (cv.wait_for(lk, std::chrono::seconds(30), [&] { return generation != g; })) return true;
        waiting--;
        return false;
    }
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

thread_local bool t_in_group = false;
thread_local std::vector<Op> t_ops;
thread_local struct ncclComm* t_group_comm = nullptr;
thread_local hipStream_t t_group_stream = nullptr;

}  // namespace

struct ncclComm {
    World* w;
    int rank;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::random_device rd;
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; i++) id->internal[i] = (char)(rd() & 0xFF);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        World*& slot = g_worlds[std::string(id.internal, NCCL_UNIQUE_ID_BYTES)];
        if (!slot) { slot = new World; slot->n = nranks; slot->ag_send.resize((size_t)nranks); }
        w = slot;
        if (w->n != nranks) return ncclInvalidArgument;
    }
    *comm = new ncclComm{w, rank};
    return w->barrier() ? ncclSuccess : ncclInvalidUsage;  // collective like the real one
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    delete comm;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "fake RCCL: a send / receive / collective without its counterpart within 30 s (or of another size)" : "fake RCCL: error"; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    World* w = comm->w;
    const size_t bytes = sendcount * type_size(datatype);
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    w->ag_send[(size_t)comm->rank] = sendbuff;
    if (!w->barrier()) return ncclInvalidUsage;
    // (on the CALLER's stream, then drained: a device-to-device hipMemcpy on the null stream may return before it has run and is not ordered with the
    // communicator's non-blocking stream - the next thing host_rccl.hip does is read recvbuff on that stream)
    for (int r = 0; r < w->n; r++)
        if (hipMemcpyAsync((char*)recvbuff + (size_t)r * bytes, w->ag_send[(size_t)r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    return w->barrier() ? ncclSuccess : ncclInvalidUsage;
}

ncclResult_t ncclGroupStart() {
    t_in_group = true;
    t_ops.clear();
    t_group_comm = nullptr;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!t_in_group) return ncclInvalidUsage;  // (host_rccl.hip only sends inside a group)
    t_ops.push_back(Op{true, sendbuff, nullptr, count * type_size(datatype), peer});
    t_group_comm = comm;
    t_group_stream = stream;
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!t_in_group) return ncclInvalidUsage;
    t_ops.push_back(Op{false, nullptr, recvbuff, count * type_size(datatype), peer});
    t_group_comm = comm;
    t_group_stream = stream;
    return ncclSuccess;
}

// Pairwise rendezvous, no barrier (a rank whose group is empty - an empty run on a rank that does not receive - never comes here):
// the sends are published in the (source, destination) mailbox, every receive waits for its mailbox entry, copies and marks it taken,
// every send waits until it was taken.  A counterpart that does not show up within 30 s is an error, not a hang.
ncclResult_t ncclGroupEnd() {
    t_in_group = false;
    if (!t_group_comm) return ncclSuccess;
    ncclComm* c = t_group_comm;
    World* w = c->w;
    if (hipStreamSynchronize(t_group_stream) != hipSuccess) return ncclUnhandledCudaError;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(30);
    bool ok = true;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        for (Op& op : t_ops)
            if (op.send) w->mail[{c->rank, op.peer}].push_back(Op{true, op.sbuf, nullptr, op.bytes, op.peer});
    }
    w->cv.notify_all();
    for (Op& op : t_ops) {
        if (op.send) continue;
        std::unique_lock<std::mutex> lk(w->mu);
        auto& box = w->mail[{op.peer, c->rank}];
        auto ready = [&] { for (Op& s : box) if (!s.matched) return true; return false; };
        if (!w->cv.wait_until(lk, deadline, ready)) { ok = false; continue; }
        Op* m = nullptr;
        for (Op& s : box) if (!s.matched) { m = &s; break; }
        if (m->bytes != op.bytes) ok = false;
        else if (hipMemcpyAsync(op.rbuf, m->sbuf, op.bytes, hipMemcpyDeviceToDevice, t_group_stream) != hipSuccess || hipStreamSynchronize(t_group_stream) != hipSuccess) ok = false;
        m->matched = true;
        lk.unlock();
        w->cv.notify_all();
    }
    for (Op& op : t_ops) {
        if (!op.send) continue;
        std::unique_lock<std::mutex> lk(w->mu);
        auto& box = w->mail[{c->rank, op.peer}];
        auto taken = [&] { return !box.empty() && box.front().matched; };
        if (!w->cv.wait_until(lk, deadline, taken)) { ok = false; if (!box.empty()) box.erase(box.begin()); continue; }
        box.erase(box.begin());
    }
    t_ops.clear();
    return ok ? ncclSuccess : ncclInvalidUsage;
}

}  // extern "C"
