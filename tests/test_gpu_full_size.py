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


def test_c5_ten_million_utf8():
    # the UTF-8 generator is a host-side Python loop: 1 M distinct haystacks, tiled to the full 10 M
    data, ends = synth.utf8_corpus(1_000_000, 32)
    data = np.tile(data, 10)
    ends = np.arange(1, 10_000_001, dtype=np.uint64) * np.uint64(32)
    _check("إنما", data, ends, max_typos=0)


def test_c4_one_shard_ragged():
    data, ends = synth.ragged_corpus(b"deadbeef", 12_500_000, device="cuda")
    fm = _check("deadbeef", data, ends, max_typos=0)
    assert fm.last_counters()["multi_chunk_scored"] > 0
