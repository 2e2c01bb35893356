"""Parity at BASELINE.json's full sizes: the HIP path against the CPU oracle, record for record, on the complete
C2 / C3 / C5 lists (10 M haystacks) and on one GPU's shard of C4 (12.5 M ragged haystacks).  The oracle runs its
match_list_parallel on every host thread (equal to match_list by the reference's own tests, parallel.rs:104-130)."""
import os
import sys

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _check(needle, data, ends, **cfg):
    threads = os.cpu_count() or 1
    om = O.Matcher(needle, lanes=(64, 64, 32), **cfg)
    oi = om.info()
    fm = F.Matcher(needle, F.Config(max_typos=cfg.get("max_typos", 0), pf_lanes=oi["pf_lanes"], sw_lanes=oi["sw_lanes"]))
    corpus = F.Corpus(packed=(data, ends))
    got = fm.match_list(corpus)                                  # ordered on the device (reverse / radix sort kernels)
    want = om.match_packed(np.concatenate([data, np.zeros(64, np.uint8)]), ends, threads=threads)
    assert len(got) == len(want) and len(got) > 0
    assert np.array_equal(got["index"], want["index"]) and np.array_equal(got["score"], want["score"]) and np.array_equal(got["exact"], want["exact"])
    # size-independent properties of the ordered result: scores descending, ties by ascending index, and the split
    # invariance match_list_parallel relies on (two halves with index offsets, merged, give the same list)
    s, i = got["score"].astype(np.int64), got["index"].astype(np.int64)
    assert np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (i[:-1] < i[1:])))
    n = len(ends)
    a = fm.match_list_into(corpus, first=0, count=n // 2, index_offset=0)
    b = fm.match_list_into(corpus, first=n // 2, count=n - n // 2, index_offset=n // 2)
    halves = np.concatenate([a, b])
    order = np.lexsort((halves["index"], -halves["score"].astype(np.int64)))
    assert np.array_equal(halves[order]["index"], got["index"]) and np.array_equal(halves[order]["score"], got["score"])
    return fm


@pytest.mark.parametrize("max_typos", [0, 2])
def test_c2_c3_ten_million_len32(max_typos):
    rows, ends = synth.fixed_corpus(b"deadbe", 10_000_000, 32, device="cuda")
    _check("deadbe", rows.cpu().numpy().reshape(-1), ends, max_typos=max_typos)


@pytest.mark.parametrize("max_typos", [0, 1])
def test_c5_ten_million_utf8(max_typos):
    # the UTF-8 generator is a host-side Python loop: 1 M distinct haystacks, tiled to the full 10 M
    # (max_typos=1: the unicode typo path at full size - superset filter, then the reference's chunked multi-path prefilter at its exact lane
    # width for every one of the ~1 M survivors: the window kernel's quarter-tile form with a grid-stride loop over ~4 000 units)
    data, ends = synth.utf8_corpus(1_000_000, 32)
    data = np.tile(data, 10)
    ends = np.arange(1, 10_000_001, dtype=np.uint64) * np.uint64(32)
    _check("إنما", data, ends, max_typos=max_typos)


def test_c4_one_shard_ragged():
    data, ends = synth.ragged_corpus(b"deadbeef", 12_500_000, device="cuda")
    fm = _check("deadbeef", data, ends, max_typos=0)
    assert fm.last_counters()["multi_chunk_scored"] > 0


def test_c4_whole_hundred_million_ragged_on_one_gpu():
    """BASELINE config 4 UNSHARDED: 100 M haystacks of 8..128 bytes (6.8 GB of bytes, 7.6 GB padded: 64-bit end offsets and a filter view
    above the 4 GiB line) resident on ONE MI355X.  Two distinct 12.5 M-item shards (seeds 12345 / 777), alternated four times.  Checked
    against the oracle on windows of the list - the first items, a window that straddles the 4 GiB line of the padded layout, the last
    items - and through a size-independent property: copy k of a shard must produce copy 0's records with the indices shifted."""
    threads = os.cpu_count() or 1
    n1 = 12_500_000
    parts = [synth.ragged_corpus(b"deadbeef", n1, seed=sd, device="cuda") for sd in (12345, 777)]
    order = [0, 1] * 4
    data = np.concatenate([parts[k][0] for k in order])
    ends = np.empty(n1 * len(order), np.uint64)
    base = 0
    for j, k in enumerate(order):
        ends[j * n1 : (j + 1) * n1] = parts[k][1] + np.uint64(base)
        base += int(parts[k][1][-1])
    n = len(ends)
    corpus = F.Corpus(packed=(data, ends))
    fm = F.Matcher("deadbeef", F.Config(max_typos=0, sort=F.SortStrategy.IndexAsc, pf_lanes=64, sw_lanes=64))
    whole = fm.match_list(corpus)
    assert fm.last_counters()["multi_chunk_scored"] > 0 and len(whole) > 1_000_000
    om = O.Matcher("deadbeef", lanes=(64, 64, 32), max_typos=0, sort="IndexAsc")
    # (the padded layout grows ~ 76 bytes per haystack: the 4 GiB line is near item 56.5 M; the window is found from the lengths)
    padded = np.cumsum((np.diff(ends, prepend=np.uint64(0)) + np.uint64(15)) & ~np.uint64(15), dtype=np.uint64)
    line = int(np.searchsorted(padded, np.uint64(1 << 32)))
    assert 0 < line < n - 300_000
    for first in (0, line - 150_000, n - 300_000):
        cnt = 300_000
        lo = int(ends[first - 1]) if first else 0
        sub = np.concatenate([data[lo : int(ends[first + cnt - 1])], np.zeros(64, np.uint8)])
        want = om.match_packed(sub, ends[first : first + cnt] - np.uint64(lo), threads=threads)
        got = fm.match_list_into(corpus, first=first, count=cnt, index_offset=0)
        assert got.tolist() == want.tolist(), first
        sl = whole[(whole["index"] >= first) & (whole["index"] < first + cnt)].copy()
        sl["index"] -= np.uint32(first)
        assert sl.tolist() == want.tolist(), first
    for j in range(2, len(order)):  # periodicity: shard copy j == shard copy j - 2, shifted by 2 * n1
        a = whole[(whole["index"] >= (j - 2) * n1) & (whole["index"] < (j - 1) * n1)]
        b = whole[(whole["index"] >= j * n1) & (whole["index"] < (j + 1) * n1)]
        assert len(a) == len(b) and np.array_equal(a["index"] + np.uint32(2 * n1), b["index"]) and np.array_equal(a["score"], b["score"]) and np.array_equal(a["exact"], b["exact"]), j
    # and the ordered form of the whole list: scores descending, ties by ascending index, same multiset of records
    fm.set_config(F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    srt = fm.match_list(corpus)
    s, i = srt["score"].astype(np.int64), srt["index"].astype(np.int64)
    assert len(srt) == len(whole) and np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (i[:-1] < i[1:])))
    assert int(srt["index"].astype(np.uint64).sum()) == int(whole["index"].astype(np.uint64).sum()) and int(s.sum()) == int(whole["score"].astype(np.int64).sum())
