"""world_size-2 gloo test of the multi-GPU composition (shard ranges, all-gather-v of per-shard match lists,
host merge).  No GPU: each rank's shard is scored by the CPU oracle (the checker standing in for the GPU stage),
and the gathered + merged result must equal the single-list `match_list` for every sort strategy."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import synth
    from frizbee_amd import SortStrategy
    from frizbee_amd.distributed import ShardExchange, all_gather_matches, merge_shard_runs, shard_range

    n = 30_001
    rows, ends = synth.fixed_corpus(b"deadbe", n, 32)
    data = np.concatenate([rows.numpy().reshape(-1), np.zeros(64, np.uint8)])
    ok = True
    for sort in ("ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"):
        for k in (0, 2):
            lo, hi = shard_range(n, rank, world)
            # shard scored with a global index offset: IndexAsc + offset == `match_list_into(chunk, start as u32)`
            m = O.Matcher("deadbe", max_typos=k, sort="IndexAsc")
            local = m.match_packed(data[lo * 32 :], ends[lo:hi] - np.uint64(lo * 32))
            local["index"] += lo
            rec = torch.from_numpy(local.view(np.uint8).copy()) if len(local) else torch.zeros(8, dtype=torch.uint8)
            runs = all_gather_matches(rec, len(local))
            merged = merge_shard_runs(runs, SortStrategy[sort])
            want = O.Matcher("deadbe", max_typos=k, sort=sort).match_packed(data, ends)
            ok = ok and merged.tolist() == want.tolist()
            # the sync-free path bench.py uses at N > 1: fixed-capacity gather to the root, double-buffered, three rounds
            cap = ShardExchange.plan(len(local))
            for transport in ("p2p", "gather"):  # batched sends / receives with the root's own run in place (default), dist.gather of the whole list
                ex = ShardExchange(cap, torch.device("cpu"), transport=transport)
                ok = ok and ex.bytes_per_rank() == 8 + 8 * cap
                for step in range(3):
                    slot = step % 2
                    ex.wait(slot)
                    buf = ex.send[slot].numpy()  # stands in for the device pipeline writing count + records in place
                    buf[:8] = np.array([len(local), len(local)], np.uint32).view(np.uint8)  # records written, matches found
                    buf[ex.HEADER : ex.HEADER + len(local) * 8] = local.view(np.uint8)
                    ex.post(slot)
                runs2 = ex.collect(0)  # steps 0 and 2 used slot 0
                if rank == 0:
                    ok = ok and merge_shard_runs(runs2, SortStrategy[sort]).tolist() == want.tolist()
                else:
                    ok = ok and runs2 is None
                # the one-call form bench.py's ordered mode uses (on the GPU: concatenation + radix sort in the root's HBM, fzb_merge_shard_runs;
                # CPU tensors: the host combine) - slot 1 was used by step 1
                class _Sorted:  # what collect_merged reads from a Matcher
                    class config:
                        pass
                _Sorted.config.sort = SortStrategy[sort]
                merged3 = ex.collect_merged(1, _Sorted)
                ok = ok and ((merged3.tolist() == want.tolist()) if rank == 0 else merged3 is None)
    # ---- BASELINE config 4 in miniature: a RAGGED list (8..128 bytes), byte-balanced shards of unequal counts, needle 'deadbeef' ----
    from frizbee_amd.distributed import shard_ranges_by_bytes

    n4 = 20_011
    data4, ends4 = synth.ragged_corpus(b"deadbeef", n4)
    data4 = np.concatenate([data4, np.zeros(64, np.uint8)])
    ranges = shard_ranges_by_bytes(ends4, world)
    lo, hi = ranges[rank]
    b0 = int(ends4[lo - 1]) if lo else 0
    ok = ok and ranges[0][0] == 0 and ranges[-1][1] == n4 and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    ok = ok and ranges[0][1] - ranges[0][0] != ranges[1][1] - ranges[1][0]          # counts differ ...
    ok = ok and abs(int(ends4[ranges[0][1] - 1]) - int(ends4[-1]) // 2) <= 128       # ... the bytes are balanced to within one haystack
    for sort in ("ScoreThenIndexAsc", "IndexDesc"):
        m = O.Matcher("deadbeef", max_typos=0, sort="IndexAsc")
        local = m.match_packed(data4[b0:], ends4[lo:hi] - np.uint64(b0))
        local["index"] += lo
        rec = torch.from_numpy(local.view(np.uint8).copy()) if len(local) else torch.zeros(8, dtype=torch.uint8)
        runs = all_gather_matches(rec, len(local))
        merged = merge_shard_runs(runs, SortStrategy[sort])
        want = O.Matcher("deadbeef", max_typos=0, sort=sort).match_packed(data4, ends4)
        ok = ok and merged.tolist() == want.tolist() and len(want) > 500
    # ---- ordered_query: the exchange was planned from a query with FEW matches; a second query with many more must grow it and succeed ----
    lo, hi = shard_range(n, rank, world)

    def local_run(k):
        loc = O.Matcher("deadbe", max_typos=k, sort="IndexAsc").match_packed(data[lo * 32 :], ends[lo:hi] - np.uint64(lo * 32))
        loc["index"] += lo
        return loc

    few, many = local_run(0), local_run(None)  # max_typos None: every haystack matches
    ex2 = ShardExchange(ShardExchange.plan(len(few), margin=1.0), torch.device("cpu"))  # agreed by every rank; holds the first query only
    ok = ok and ex2.cap < len(many)

    def writer(loc):
        def run(records_ptr, capacity, count_ptr):  # stands in for fzb_match_list_device: writes min(found, capacity) records + both counters
            buf = ex2.send[0].numpy()
            w = min(len(loc), capacity)
            buf[:8] = np.array([w, len(loc)], np.uint32).view(np.uint8)
            buf[ex2.HEADER : ex2.HEADER + w * 8] = loc[:w].view(np.uint8)
        return run

    class _S:
        class config:
            sort = SortStrategy.ScoreThenIndexAsc
    r1 = ex2.ordered_query(writer(few), _S)
    ok = ok and ex2.grown == 0
    r2 = ex2.ordered_query(writer(many), _S)
    ok = ok and ex2.grown == 1 and ex2.cap >= len(many)
    if rank == 0:
        ok = ok and r1.tolist() == O.Matcher("deadbe", max_typos=0, sort="ScoreThenIndexAsc").match_packed(data, ends).tolist()
        ok = ok and r2.tolist() == O.Matcher("deadbe", max_typos=None, sort="ScoreThenIndexAsc").match_packed(data, ends).tolist() and len(r2) == n
    else:
        ok = ok and r1 is None and r2 is None
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_shard_gather_merge_equals_single_list():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)], res


def test_byte_balanced_shards_partition_a_ragged_list():
    from frizbee_amd.distributed import shard_ranges_by_bytes
    rng = np.random.default_rng(3)
    for n in (1, 2, 9, 1000, 100_003):
        ends = np.cumsum(rng.integers(8, 129, n).astype(np.uint64), dtype=np.uint64)
        for w in (1, 2, 3, 8):
            r = shard_ranges_by_bytes(ends, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:])) and all(a <= b for a, b in r)
            if n >= 1000:
                sizes = [int(ends[b - 1]) - (int(ends[a - 1]) if a else 0) for a, b in r]
                assert max(sizes) - min(sizes) <= 2 * 128, (n, w, sizes)
    assert shard_ranges_by_bytes(np.zeros(0, np.uint64), 4) == [(0, 0)] * 4


def test_shard_ranges_partition_the_list():
    from frizbee_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 9, 100_000_000):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
