// One process per GPU through the C ABI alone - what a Rust (or any) host with one process per device does, as a runnable C++ program:
//   rccl_ranks N [items]      forks N ranks; rank r binds GPU r, takes its contiguous share of a synthetic list (fzb_shard_ranges, by count),
//                             uploads it, and all ranks call fzb_match_list_parallel_rccl (first gather-to-root, then gather-to-all);
//                             every receiver compares the list with fzb_match_list over the WHOLE list on its own GPU.
// The 128-byte communicator id travels from rank 0 to the others through pipes of the parent - the "whatever channel started them" of
// include/frizbee_hip.h.  Needs N visible GPUs (RCCL refuses two ranks on one device): on the one-GPU build boxes only N = 1 runs
// (tests/test_cpp_facade.py); worlds of 2 / 3 / 8 are covered there by thread-ranks over a test double (tests/test_gpu_sharded.py).
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "frizbee_hip.h"

extern "C" int hipSetDevice(int);  // (the one HIP call of the host side: which GPU this process uses)

#define CHECK(expr)                                                                                  \
    do {                                                                                             \
        int rc_ = (expr);                                                                            \
        if (rc_) { fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #expr, rc_, fzb_last_error()); return 1; } \
    } while (0)

static void make_list(size_t n, std::string& bytes, std::vector<uint64_t>& ends) {
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const char* alpha = "abcdefxyz_/.-";
    for (size_t i = 0; i < n; i++) {
        const size_t len = 8 + rnd() % 90;
        std::string h(len, ' ');
        for (auto& c : h) c = alpha[rnd() % 13];
        if (rnd() % 10 < 3) {  // plant "deadbe" as a subsequence
            size_t q = 0;
            for (const char* p = "deadbe"; *p && q < len; p++) { q += rnd() % 3; if (q < len) h[q++] = *p; }
        }
        bytes += h;
        ends.push_back(bytes.size());
    }
}

static int run_rank(int rank, int world, size_t n, const uint8_t* id) {
    // (every rank decides this alike BEFORE the communicator: a rank that left alone would leave the others waiting inside ncclCommInitRank)
    int gpus = 0;
    CHECK(fzb_device_count(&gpus));
    if (gpus < world) { fprintf(stderr, "rank %d: %d GPU(s) visible, %d needed (one per rank)\n", rank, gpus, world); return 1; }
    if (hipSetDevice(rank)) { fprintf(stderr, "rank %d: hipSetDevice failed\n", rank); return 1; }
    std::string bytes;
    std::vector<uint64_t> ends;
    make_list(n, bytes, ends);  // (every rank regenerates the list: the program is about the exchange, not about loading data)
    std::vector<uint64_t> bounds((size_t)world + 1);
    CHECK(fzb_shard_ranges(ends.data(), n, world, 0, bounds.data()));
    const uint64_t lo = bounds[(size_t)rank], hi = bounds[(size_t)rank + 1];
    const uint64_t b0 = lo ? ends[lo - 1] : 0;
    std::vector<uint64_t> my_ends(ends.begin() + (long)lo, ends.begin() + (long)hi);
    for (auto& e : my_ends) e -= b0;
    fzb_corpus *shard = nullptr, *whole = nullptr;
    CHECK(fzb_corpus_upload((const uint8_t*)bytes.data() + b0, my_ends.data(), my_ends.size(), &shard));
    fzb_config cfg;
    fzb_config_default(&cfg);
    fzb_matcher* m = nullptr;
    CHECK(fzb_matcher_create(&cfg, (const uint8_t*)"deadbe", 6, &m));
    fzb_shard_comm* comm = nullptr;
    CHECK(fzb_shard_comm_create(id, rank, world, &comm));
    int bad = 0;
    for (int flags : {FZB_GATHER_ROOT, FZB_GATHER_ALL}) {
        fzb_match* got = nullptr;
        size_t n_got = 0;
        CHECK(fzb_match_list_parallel_rccl(m, shard, (uint32_t)lo, comm, flags, &got, &n_got));
        if (rank == 0 || flags == FZB_GATHER_ALL) {
            if (!whole) CHECK(fzb_corpus_upload((const uint8_t*)bytes.data(), ends.data(), n, &whole));
            fzb_match* want = nullptr;
            size_t n_want = 0;
            CHECK(fzb_match_list(m, whole, &want, &n_want));
            bool same = n_got == n_want && n_want > 0;
            for (size_t i = 0; same && i < n_want; i++) same = got[i].index == want[i].index && got[i].score == want[i].score && got[i].exact == want[i].exact;
            printf("rank %d, %s: %zu records, %s the single-GPU list\n", rank, flags ? "gather to all" : "gather to root", n_got, same ? "equal to" : "DIFFERENT from");
            bad += !same;
            fzb_matches_free(want);
        } else if (n_got) bad++;
        fzb_matches_free(got);
    }
    fzb_shard_comm_free(comm);
    fzb_matcher_free(m);
    fzb_corpus_free(shard);
    if (whole) fzb_corpus_free(whole);
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : 200000;
    if (world < 1 || world > 64) { fprintf(stderr, "usage: rccl_ranks N [items]\n"); return 2; }
    // no HIP / RCCL call before the forks: rank 0 draws the id in its own process and hands it to the parent, which passes it on
    std::vector<int> to_child((size_t)world), pids((size_t)world);
    int from0[2];
    if (pipe(from0)) return 2;
    for (int r = 0; r < world; r++) {
        int p[2];
        if (pipe(p)) return 2;
        const pid_t pid = fork();
        if (pid == 0) {
            close(p[1]);
            close(from0[0]);
            if (r != 0) close(from0[1]);
            for (int q = 0; q < r; q++) close(to_child[(size_t)q]);  // (only the parent writes to the other ranks)
            uint8_t id[FZB_RCCL_ID_BYTES];
            if (r == 0) {
                const int rank = 0;
                CHECK(fzb_rccl_unique_id(id));
                if (write(from0[1], id, sizeof(id)) != (ssize_t)sizeof(id)) return 2;
            } else if (read(p[0], id, sizeof(id)) != (ssize_t)sizeof(id)) return 2;
            return run_rank(r, world, n, id);
        }
        close(p[0]);
        to_child[(size_t)r] = p[1];
        pids[(size_t)r] = pid;
    }
    close(from0[1]);  // (the parent keeps no write end: a rank 0 that dies without an id is an end of file here, not a wait for ever)
    uint8_t id[FZB_RCCL_ID_BYTES];
    const bool have_id = read(from0[0], id, sizeof(id)) == (ssize_t)sizeof(id);
    if (!have_id) fprintf(stderr, "rank 0 sent no communicator id\n");
    for (int r = 1; r < world; r++) {
        if (have_id && write(to_child[(size_t)r], id, sizeof(id)) != (ssize_t)sizeof(id)) fprintf(stderr, "rank %d does not take the id\n", r);
        close(to_child[(size_t)r]);  // (without an id: end of file for the rank, which leaves)
    }
    int bad = have_id ? 0 : 1;
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pids[(size_t)r], &st, 0);
        bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
    }
    printf("rccl_ranks %d: %s\n", world, bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
