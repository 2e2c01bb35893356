"""The multi-device form of the boundary (`fzb_corpus_upload_sharded` + `fzb_match_list_parallel_sharded`: match_list_parallel with
one DEVICE per worker, src/matcher/parallel.rs:18-89) and the RCCL exchange of frizbee_amd.distributed, both on whatever the box has:
with one GPU the shards share it (FZB_SHARD_OVERSUBSCRIBE) and the process group has one rank - the code paths are the ones N GPUs run."""
import os
import subprocess
import sys

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu

SORTS = ("ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc")


@pytest.fixture(params=["pull", "pull_workers", "copy"])
def gather_mode(request):
    """How the runs reach the root: one concatenation kernel that pulls them (every shard on the root device - the only arrangement
    one GPU offers; the shards enqueued by the calling thread, or - pull_workers - by the per-shard worker threads that shards on other
    devices use) or the form shards on OTHER devices take - count to the host, hipMemcpyPeerAsync to the run's place - forced here with
    FZB_SHARD_GATHER=copy so that one GPU exercises it too."""
    if request.param == "copy":
        os.environ["FZB_SHARD_GATHER"] = "copy"
    if request.param == "pull_workers":
        os.environ["FZB_SHARD_INLINE"] = "0"
    F.lib().fzb_debug_reload_knobs()
    yield request.param
    os.environ.pop("FZB_SHARD_GATHER", None)
    os.environ.pop("FZB_SHARD_INLINE", None)
    F.lib().fzb_debug_reload_knobs()


def test_sharded_match_list_parallel_equals_match_list_for_every_sort_and_shard_count(gather_mode):
    rows, ends = synth.fixed_corpus(b"deadbe", 120_001, 32)
    data = rows.numpy().reshape(-1)
    odata = np.concatenate([data, np.zeros(64, np.uint8)])
    have = F.device_count()
    for ndev in (1, 2, 3, 8):
        sc = F.ShardedCorpus(packed=(data, ends), ndev=ndev, oversubscribe=ndev > have)
        assert len(sc) == 120_001 and [s[:2] for s in sc.shards()] == F.shard_ranges(ends, ndev)
        assert all(dev == g % have for g, (_, _, dev) in enumerate(sc.shards()))
        for sort in SORTS:
            for typos in (0, 1):
                m = F.Matcher("deadbe", F.Config(max_typos=typos, sort=F.SortStrategy[sort], pf_lanes=64, sw_lanes=64))
                want = O.Matcher("deadbe", lanes=(64, 64, 32), max_typos=typos, sort=sort).match_packed(odata, ends)
                got = m.match_list_parallel_sharded(sc)
                rep = m.shard_report()  # how the runs reached the root: on one GPU every shard shares the root device
                assert rep.startswith("root device") and rep.count("shard ") == ndev and (have > 1 or rep.count("same device") == ndev), rep
                assert got.tolist() == want.tolist(), (ndev, sort, typos, len(got), len(want))
                assert m.match_list_parallel_sharded(sc).tolist() == want.tolist()  # again: the clones' workspaces are reused
        del sc


def test_sharded_ragged_list_byte_balanced_and_requery_after_set_pattern(gather_mode):
    data, ends = synth.ragged_corpus(b"deadbeef", 60_013)
    odata = np.concatenate([data, np.zeros(64, np.uint8)])
    have = F.device_count()
    sc = F.ShardedCorpus(packed=(data, ends), ndev=4, by_bytes=True, oversubscribe=4 > have)
    shards = sc.shards()
    assert [s[:2] for s in shards] == F.shard_ranges(ends, 4, by_bytes=True)
    sizes = [int(ends[hi - 1]) - (int(ends[lo - 1]) if lo else 0) for lo, hi, _ in shards]
    assert max(sizes) - min(sizes) <= 2 * 128 and len({hi - lo for lo, hi, _ in shards}) > 1  # bytes balanced, counts differ
    m = F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64))
    for needle in ("deadbeef", "dead", "Beef", "", "déad", "deadbeef"):  # Matcher::set_pattern: the per-shard clones follow
        m.set_pattern(needle)
        want = O.Matcher(needle, lanes=(64, 64, 32)).match_packed(odata, ends)
        assert m.match_list_parallel_sharded(sc).tolist() == want.tolist(), needle
    m.set_config(F.Config(max_typos=1, sort=F.SortStrategy.IndexDesc, pf_lanes=64, sw_lanes=64))
    want = O.Matcher("deadbeef", lanes=(64, 64, 32), max_typos=1, sort="IndexDesc").match_packed(odata, ends)
    assert m.match_list_parallel_sharded(sc).tolist() == want.tolist()


def test_sharded_edge_cases(gather_mode):
    have = F.device_count()
    for hs in ([], ["deadbe"], ["x", "deadbe", "", "dead_be"]):
        for ndev in (1, 3):
            sc = F.ShardedCorpus(hs, ndev=ndev, oversubscribe=ndev > have)
            for needle in ("deadbe", ""):
                for sort in SORTS:
                    got = F.Matcher(needle, F.Config(sort=F.SortStrategy[sort], pf_lanes=64)).match_list_parallel_sharded(sc)
                    assert got.tolist() == O.Matcher(needle, sort=sort).match_list(hs).tolist(), (hs, ndev, needle, sort)
    with pytest.raises(F.FrizbeeError) as e:  # more devices than the box has, not allowed to share: refused, never run on fewer
        F.ShardedCorpus(["a"], ndev=have + 1)
    assert e.value.code == 4 and "visible" in str(e.value)
    with pytest.raises(F.FrizbeeError):
        F.ShardedCorpus(["a"], ndev=0)


def test_upload_builds_the_device_layout_for_every_length_mix():
    # fzb_corpus_upload (raw bytes + offsets travel as they are, the padded-16 layout is built by device kernels): lengths around the
    # 16-byte vector, empty haystacks (also runs of them, first and last), multi-chunk and > 1024-byte haystacks, NUL bytes
    rng = np.random.default_rng(11)
    alpha = b"abcdeDEF_-/ 01\0"
    pool = [0, 0, 1, 2, 5, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 100, 127, 128, 129, 200, 1023, 1024, 1025, 1100, 3000]
    for n, lens in ((5000, None), (3, [0, 0, 0]), (4, [0, 7, 0, 0]), (2049, None), (1024, [16] * 1024), (1025, [48] * 1025), (2000, [33] * 2000)):
        ls = rng.choice(pool, n) if lens is None else np.array(lens)
        hs = [bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), int(k))) for k in ls]
        cp = F.Corpus(hs)
        for needle, cfg in (("deadbe", dict(max_typos=0)), ("dea", dict(max_typos=1)), ("ab_c", dict(max_typos=None)), ("é", dict(max_typos=0, unicode="Always"))):
            want = O.Matcher(needle, **cfg).match_list(hs)
            fc = F.Config(max_typos=cfg["max_typos"], unicode=F.UnicodeMatching[cfg.get("unicode", "Smart")], pf_lanes=64)
            assert F.Matcher(needle, fc).match_list(cp).tolist() == want.tolist(), (n, needle)
    with pytest.raises(F.FrizbeeError) as e:
        F.Corpus(packed=(np.zeros(64, np.uint8), np.array([8, 4, 12], np.uint64)))
    assert "non-decreasing" in str(e.value)


def test_uniform_length_promise_is_refused_on_an_uploaded_corpus():
    cp = F.Corpus(["abcdefgh"] * 10)
    assert F.lib().fzb_corpus_set_uniform_len(cp.h, 8) == 0        # the detected value: accepted (no-op)
    assert F.lib().fzb_corpus_set_uniform_len(cp.h, 16) == 1       # anything else would make span-computing and offset-reading kernels disagree
    cp2 = F.Corpus(["abc", "abcdefgh"])
    assert F.lib().fzb_corpus_set_uniform_len(cp2.h, 0) == 0 and F.lib().fzb_corpus_set_uniform_len(cp2.h, 8) == 1


def test_rccl_exchange_with_one_rank_matches_the_oracle():
    """The N > 1 path of bench.py end to end on one GPU: `nccl` process group (world size 1), ShardExchange's asynchronous double-buffered
    gather, per-shard device sort, host k-merge - the merged list must be the oracle's."""
    code = r'''
import os, sys
sys.path[:0] = [%(root)r, %(tests)r, %(tools)r]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import frizbee_amd as F, oracle_lib as O, synth
from frizbee_amd.distributed import ShardExchange, merge_shard_runs, all_gather_matches
n = 300_000
rows, ends = synth.fixed_corpus(b"deadbe", n, 32)
data = rows.numpy().reshape(-1)
cp = F.Corpus(packed=(data, ends))
odata = np.concatenate([data, np.zeros(64, np.uint8)])
for typos in (0, 2):
    m = F.Matcher("deadbe", F.Config(max_typos=typos, pf_lanes=64, sw_lanes=64))
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr(), index_offset=7); torch.cuda.synchronize()
    k = int(cnt[0].item()); assert int(cnt[1].item()) == k
    ex = ShardExchange(ShardExchange.plan(k, device=dev), dev)
    for step in range(5):  # the bench loop: pipeline writes count + records into the slot, asynchronous gather, next step overlaps
        slot = step & 1
        ex.wait(slot)
        m.match_list_device(cp, ex.records_ptr(slot), ex.cap, ex.count_ptr(slot), index_offset=7)
        ex.post(slot)
    runs = ex.collect(0); ex.collect(1)
    for sort in ("ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"):
        want = O.Matcher("deadbe", lanes=(64, 64, 32), max_typos=typos, sort=sort).match_packed(odata, ends); want["index"] += 7
        assert merge_shard_runs(runs, F.SortStrategy[sort]).tolist() == want.tolist(), (typos, sort)
        # the device form of the combine (what bench.py's ordered mode runs on the root): gathered runs -> concatenation + radix sort in HBM
        ms = F.Matcher("deadbe", F.Config(max_typos=typos, sort=F.SortStrategy[sort], pf_lanes=64, sw_lanes=64))
        ms.match_list_device(cp, ex.records_ptr(0), ex.cap, ex.count_ptr(0), index_offset=7); ex.post(0)
        assert ex.collect_merged(0, ms).tolist() == want.tolist(), (typos, sort, "device merge")
        # ... and the same run three times as three "shards" (indices repeat: only the concatenation order and the stable sort are under test)
        torch.cuda.synchronize()
        b = ex.recv[0][0]
        tri = ms.merge_shard_runs([b.data_ptr() + 8] * 3, [b.data_ptr()] * 3, [ex.cap] * 3)
        exp = np.concatenate([runs[0]] * 3)                      # match_list's post-step over the concatenation (src/matcher/mod.rs:215-221)
        if sort.endswith("Desc"): exp = exp[::-1]
        if sort.startswith("Score"): exp = exp[np.argsort(-exp["score"].astype(np.int64), kind="stable")]
        assert tri.tolist() == exp.tolist(), (typos, sort, "three runs")
    runs2 = all_gather_matches(out, cnt[0])
    assert len(runs2) == 1 and runs2[0].tolist() == runs[0].tolist()
    small = ShardExchange(10, dev)   # a capacity below the match count is reported, never truncated silently
    m.match_list_device(cp, small.records_ptr(0), small.cap, small.count_ptr(0)); small.post(0)
    try:
        small.collect(0); raise SystemExit("truncation not reported")
    except RuntimeError as e:
        assert str(k) in str(e), e
    m.match_list_device(cp, small.records_ptr(0), small.cap, small.count_ptr(0)); small.post(0)
    try:
        small.collect_merged(0, m); raise SystemExit("truncation not reported by the device merge")
    except F.FrizbeeError as e:
        assert e.code == 5 and "truncated" in str(e), e
dist.barrier(); dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
''' % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "tools": os.path.join(ROOT, "tools")}
    for transport in ("p2p", "gather"):  # (with one rank the default transport has nothing to move - the root's run is in place; dist.gather still copies it)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, FZB_EXCHANGE_TRANSPORT=transport))
        assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (transport, r.stdout[-2000:], r.stderr[-4000:])


def test_bench_refuses_more_ranks_than_gpus():
    have = F.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout), (r.stdout[-500:], r.stderr[-500:])


def test_two_rank_rehearsal_of_the_bench_on_one_gpu():
    """`FZB_BENCH_BACKEND=gloo python bench.py --gpus 2`: the N > 1 path of the driver's bench with TWO real ranks on a box with one GPU -
    bench.py starts the ranks under torch.distributed.run itself, both score their shard on the visible GPU with a global index offset,
    the exchange moves CPU tensors (gloo), rank 0 combines.  Everything an RCCL run executes except the transport: the rank bookkeeping,
    the double-buffered exchange, ordered_query's grow-and-retry, the per-shard oracle check, and BASELINE configs[3] cut into
    byte-balanced shards.  The line must be valid JSON with ranks_seen == 2 and the merged list equal to the oracle's."""
    import json
    env = dict(os.environ, FZB_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--per-gpu", "1000000", "--steps", "5", "--warmup", "2", "--c4-total", "2000000"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and "REHEARSAL" in j["config"]["backend"]
    assert j["value"] > 0 and j["scaling"] == "weak" and j["config"]["exchange"]["merged_len"] == j["config"]["exchange"]["matches_all_shards"]
    x = j["config"]["exchange"]  # the attribution fields of the N > 1 line (round 6)
    assert x["exchange_bytes_per_rank_per_step"] == 8 + 8 * x["exchange_capacity_records"] and x["exchange_bytes_into_root_per_step"] == x["exchange_bytes_per_rank_per_step"]
    assert x["exchange_capacity_records"] <= 1.05 * (x["matches_all_shards"] / 2 * 1.1) + 4097 and x["value_without_exchange"] > 0 and x["exchange_alone_ms_per_step"] > 0 and x["transport"].startswith("p2p")
    e = j["e2e_sorted_merge"]
    assert e["equals_host_merge"] and e["merged_equals_oracle_list"] and e["every_shard_head_equals_oracle"]["equal_on_every_rank"]
    assert e["grow_and_retry"]["times_grown"] >= 1 and e["grow_and_retry"]["result_equals"]
    (name, c4), = j["configs"].items()
    assert "configs[3]" in name and c4["n_gpus"] == 2 and c4["haystacks"] == 2_000_000 and all(v is True for k, v in c4["checks"].items() if k != "oracle_items_per_shard"), c4["checks"]
    assert abs(c4["shards"][0]["bytes"] - c4["shards"][1]["bytes"]) <= 256 and c4["shards"][0]["range"][1] == c4["shards"][1]["range"][0]


# ---- one process per GPU BELOW the C ABI: fzb_shard_comm / fzb_match_list_parallel_rccl (csrc/host_rccl.hip) ------------------------------

_RCCL_CODE = r'''
import os, sys, threading, time, faulthandler
faulthandler.dump_traceback_later(240, exit=True)   # a hang shows where, and ends
sys.path[:0] = [%(root)r, %(tests)r, %(tools)r]
import numpy as np
import frizbee_amd as F, oracle_lib as O, synth
from frizbee_amd.distributed import RcclShardComm, shard_range, shard_ranges_by_bytes
WORLD = int(sys.argv[1])
T0 = time.time()
def shards_of(data, ends, ranges):
    out = []
    for lo, hi in ranges:
        b0 = int(ends[lo - 1]) if lo else 0
        b1 = int(ends[hi - 1]) if hi else 0
        out.append(F.Corpus(packed=(data[b0:b1].copy(), (ends[lo:hi] - np.uint64(b0)).astype(np.uint64))))
    return out
def run_world(ranges, shards, needle, kw, all_ranks, uid):
    res, errs = [None] * WORLD, []
    def rank_main(r):
        try:
            comm = RcclShardComm(rank=r, world=WORLD, unique_id=uid)
            m = F.Matcher(needle, F.Config(pf_lanes=64, sw_lanes=64, **kw))
            for rep in range(2):  # the second query reuses the communicator's buffers
                res[r] = (comm.match_list_parallel(m, shards[r], ranges[r][0], all_ranks=all_ranks), comm.last_exchange_bytes())
            comm.close()
        except BaseException as e:
            errs.append((r, repr(e)))
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(WORLD)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    return res
cases = 0
# (a) the C2 shape cut by count, (b) a ragged list cut by bytes with one EMPTY shard in the middle when the world allows it
n = 60_000
rows, ends = synth.fixed_corpus(b"deadbe", n, 32)
data = rows.numpy().reshape(-1)
ranges_a = [shard_range(n, r, WORLD) for r in range(WORLD)]
rag, rends = synth.ragged_corpus(b"deadbeef", 40_000, 8, 128, seed=5)
ranges_b = shard_ranges_by_bytes(rends, WORLD)
if WORLD >= 3:  # an empty shard: rank 1 gives its share to rank 2
    ranges_b[2] = (ranges_b[1][0], ranges_b[2][1]); ranges_b[1] = (ranges_b[1][0], ranges_b[1][0])
for (dat, en, ranges, needle) in ((data, ends, ranges_a, "deadbe"), (rag, rends, ranges_b, "deadbeef")):
    shards = shards_of(dat, en, ranges)
    odata = np.concatenate([dat, np.zeros(64, np.uint8)])
    for kw, okw in (({}, {}), ({"max_typos": 1, "sort": F.SortStrategy.ScoreThenIndexDesc}, {"max_typos": 1, "sort": "ScoreThenIndexDesc"}), ({"sort": F.SortStrategy.IndexAsc}, {"sort": "IndexAsc"})):
        want = O.Matcher(needle, lanes=(64, 64, 32), **okw).match_packed(odata, en)
        for all_ranks in (False, True):
            uid = RcclShardComm.unique_id()
            res = run_world(ranges, shards, needle, kw, all_ranks, uid)
            for r in range(WORLD):
                got, (sent, recvd) = res[r]
                if r == 0 or all_ranks:
                    assert got.tolist() == want.tolist(), (needle, kw, all_ranks, r, len(got), len(want))
                else:
                    assert len(got) == 0
            # exactly the records travel: every non-receiving rank sends its run once (to every other rank with all_ranks)
            per_rank = [int(((want["index"] >= lo) & (want["index"] < hi)).sum()) * 8 for lo, hi in ranges]
            for r in range(WORLD):
                sent, recvd = res[r][1]
                assert sent == per_rank[r] * ((WORLD - 1) if all_ranks else (1 if r else 0)), (r, sent, per_rank)
                assert recvd == ((sum(per_rank) - per_rank[r]) if (all_ranks or r == 0) else 0), (r, recvd, per_rank)
            cases += 1
            print("case", cases, needle, okw, all_ranks, "%%.1f s" %% (time.time() - T0), flush=True)
# the empty pattern (CompiledPatterns::Empty): every index of every share, score 0, reversed for the *Desc strategies
shards = shards_of(rag, rends, ranges_b)
for sort in (F.SortStrategy.ScoreThenIndexAsc, F.SortStrategy.IndexDesc):
    res = run_world(ranges_b, shards, "", {"sort": sort}, True, RcclShardComm.unique_id())
    idx = np.arange(len(rends), dtype=np.uint32)[::-1 if sort == F.SortStrategy.IndexDesc else 1]
    for r in range(WORLD):
        got = res[r][0]
        assert got["index"].tolist() == idx.tolist() and not got["score"].any() and not got["exact"].any(), (sort, r)
print("RCCL-CABI-OK", WORLD, cases)
'''


def _fake_rccl():
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.cpp")
    so = os.path.join(ROOT, "tests", "cpp", "libfake_rccl.so")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so, "-L/opt/rocm/lib", "-lamdhip64",
                               "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return so


@pytest.mark.parametrize("world", [2, 3, 8])
def test_c_abi_rccl_exchange_with_several_ranks_as_threads(world):
    """fzb_match_list_parallel_rccl with a world of 2 / 3 / 8: the ranks are threads of ONE process on the one GPU and the nine RCCL entry
    points are a test double (tests/cpp/fake_rccl.cpp, through FZB_RCCL_LIB) that fails on any send / receive without its exact
    counterpart - everything of the exchange except RCCL itself: the count all-gather, who sends what to whom at which offset, empty shards,
    gather-to-root and gather-to-all, the rank-order merge; every receiver's list must be the oracle's for the whole list."""
    code = _RCCL_CODE % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "tools": os.path.join(ROOT, "tools")}
    r = subprocess.run([sys.executable, "-c", code, str(world)], capture_output=True, text=True, timeout=400, env=dict(os.environ, FZB_RCCL_LIB=_fake_rccl()))
    assert r.returncode == 0 and "RCCL-CABI-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_c_abi_rccl_exchange_with_the_real_library_and_one_rank():
    """The same entry points on the REAL RCCL (dlopen of librccl.so.1) with the one rank a one-GPU box allows: ncclCommInitRank, the count
    all-gather on the communicator's stream, the merge; result = fzb_match_list's."""
    code = r'''
import sys
sys.path[:0] = [%(root)r, %(tests)r, %(tools)r]
import numpy as np
import frizbee_amd as F, synth
from frizbee_amd.distributed import RcclShardComm
rows, ends = synth.fixed_corpus(b"deadbe", 200_000, 32)
cp = F.Corpus(packed=(rows.numpy().reshape(-1), ends))
comm = RcclShardComm(rank=0, world=1)
for kw in ({}, {"max_typos": 2}, {"sort": F.SortStrategy.IndexDesc}):
    m = F.Matcher("deadbe", F.Config(pf_lanes=64, sw_lanes=64, **kw))
    want = m.match_list(cp); want["index"] += 11   # (a constant offset moves no record in any of the orders)
    for all_ranks in (False, True):
        got = comm.match_list_parallel(m, cp, 11, all_ranks=all_ranks)
        assert len(got) > 1000 and got.tolist() == want.tolist(), (kw, all_ranks, len(got), len(want))
        assert comm.last_exchange_bytes() == (0, 0)
empty = F.Corpus(packed=(np.zeros(0, np.uint8), np.zeros(0, np.uint64)))
assert len(comm.match_list_parallel(F.Matcher("deadbe"), empty, 0)) == 0
assert len(RcclShardComm(rank=0, world=1).match_list_parallel(F.Matcher("deadbe"), empty, 0)) == 0   # a fresh communicator: nothing allocated yet
got = comm.match_list_parallel(F.Matcher("", F.Config(sort=F.SortStrategy.IndexDesc)), cp, 5)
assert got["index"].tolist() == list(range(5 + len(cp) - 1, 4, -1)) and not got["score"].any()
comm.close()
print("RCCL-CABI-REAL-OK")
''' % {"root": ROOT, "tests": os.path.join(ROOT, "tests"), "tools": os.path.join(ROOT, "tools")}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FZB_RCCL_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "RCCL-CABI-REAL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
