"""k12_fused (frizbee_amd/csrc/kernels_fused.hip, opt-in with FZB_FUSED=1): filter, ordering and scorer of a short-haystack corpus in
one persistent kernel.  Its records must be those of the default three-kernel pipeline (k1_dfa -> k_compact1 -> k2b_dp_short) and of the oracle's
match_list (src/matcher/mod.rs:170-222), in haystack order, for every shape of list that stresses the per-wave queues, the staging and the gather kernel:
no survivors, every haystack a survivor, tile-boundary counts, short ragged lists, truncated result buffers, repeated launches."""
import os
import sys

import numpy as np
import pytest
import torch

import frizbee_amd as F
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu

CFG = dict(sort=F.SortStrategy.IndexAsc, pf_lanes=64, sw_lanes=64)


def both(m, cp):
    """records of the fused kernel and of the three-kernel pipeline for the same query"""
    os.environ["FZB_FUSED"] = "1"
    try:
        fused = m.match_list(cp).copy()
        os.environ["FZB_FUSED_NO_UNIFORM"] = "1"  # the instantiation that takes the lengths from the end offsets
        assert m.match_list(cp).tolist() == fused.tolist()
    finally:
        os.environ.pop("FZB_FUSED_NO_UNIFORM", None)
        os.environ.pop("FZB_FUSED", None)
    split = m.match_list(cp).copy()
    return fused, split


def hay_list(rows):
    return [bytes(r) for r in rows.numpy()]


@pytest.mark.parametrize("n", [1, 63, 1023, 1024, 1025, 4096 + 17, 300_001])
@pytest.mark.parametrize("lanes", [64, 32])
def test_fused_equals_split_pipeline_and_oracle(n, lanes):
    length = 32 if lanes == 64 else 16
    rows, ends = synth.fixed_corpus(b"deadbe", n, length)
    cp = F.Corpus(packed=(rows.numpy().reshape(-1), ends))
    m = F.Matcher("deadbe", F.Config(sort=F.SortStrategy.IndexAsc, pf_lanes=lanes, sw_lanes=lanes))
    fused, split = both(m, cp)
    assert fused.tolist() == split.tolist()
    if n <= 5000:
        want = O.Matcher("deadbe", lanes=(lanes, lanes, lanes // 2), sort="IndexAsc").match_list(hay_list(rows))
        assert fused.tolist() == want.tolist()
    assert (np.diff(fused["index"].astype(np.int64)) > 0).all()  # haystack order


@pytest.mark.parametrize("density", ["none", "all", "first_tile_only", "last_haystack_only", "alternating_tiles"])
def test_survivor_densities(density):
    n = 40 * 1024 + 5
    rng = np.random.default_rng(3)
    rows = rng.integers(ord("f") + 1, ord("z"), size=(n, 32), dtype=np.uint8)  # no needle byte anywhere
    hit = np.frombuffer(b"xx_deadbe_Dead.be-deadBE_yy__dea", dtype=np.uint8)
    if density == "all":
        rows[:] = hit
    elif density == "first_tile_only":
        rows[:1024] = hit
    elif density == "last_haystack_only":
        rows[-1] = hit
    elif density == "alternating_tiles":
        for t in range(0, 40, 2):
            rows[t * 1024:(t + 1) * 1024] = hit
    ends = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(32)
    cp = F.Corpus(packed=(rows.reshape(-1), ends))
    m = F.Matcher("deadbe", F.Config(**CFG))
    fused, split = both(m, cp)
    assert fused.tolist() == split.tolist()
    expect = {"none": 0, "all": n, "first_tile_only": 1024, "last_haystack_only": 1, "alternating_tiles": 20 * 1024}[density]
    assert len(fused) == expect
    if expect:
        one = O.Matcher("deadbe").match_list([bytes(hit)])
        assert set(fused["score"].tolist()) == {int(one["score"][0])}


def test_short_ragged_list_with_end_offsets_and_uppercase_needle():
    # lengths 0..32 (no uniform length: the spans come from the end offsets), needle with capitals (the UPPER instantiation)
    rng = np.random.default_rng(11)
    alpha = b"abdeDEBA_-/ 01"
    hs = [bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(0, 33)))) for _ in range(20_000)]
    for needle in ("deadbe", "DeAd", "e", "a_b"):
        m = F.Matcher(needle, F.Config(**CFG))
        cp = F.Corpus(hs)
        fused, split = both(m, cp)
        want = O.Matcher(needle, sort="IndexAsc").match_list(hs)
        assert fused.tolist() == want.tolist(), needle
        assert split.tolist() == want.tolist(), needle


def test_truncated_result_buffer_and_repeated_launches():
    rows, ends = synth.fixed_corpus(b"deadbe", 150_000, 32)
    cp = F.Corpus(packed=(rows.numpy().reshape(-1), ends))
    m = F.Matcher("deadbe", F.Config(**CFG))
    whole = m.match_list(cp).copy()
    dev = torch.device("cuda", 0)
    for cap in (0, 1, 255, 256, 257, len(whole) // 2, len(whole), len(whole) + 100):
        out = torch.full(((cap + 64) * 8,), 0xAB, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        for _ in range(3):  # the scheduler words and the generation number carry from launch to launch
            m.match_list_device(cp, out.data_ptr(), cap, cnt.data_ptr())
        torch.cuda.synchronize()
        k = min(cap, len(whole))
        assert int(cnt[0].item()) == k
        host = out.cpu().numpy()
        assert host[: k * 8].view(F.MATCH_DTYPE).tolist() == whole[:k].tolist()
        assert (host[max(cap, k) * 8:] == 0xAB).all()


def test_sub_ranges_and_index_offsets():
    rows, ends = synth.fixed_corpus(b"deadbe", 50_000, 32)
    cp = F.Corpus(packed=(rows.numpy().reshape(-1), ends))
    m = F.Matcher("deadbe", F.Config(**CFG))
    whole = m.match_list(cp).copy()
    dev = torch.device("cuda", 0)
    for first, count, off in ((0, 50_000, 7), (1, 1023, 0), (1024, 1024, 100), (33_333, 16_667, 1 << 20)):
        out = torch.zeros((count + 8) * 8, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        m.match_list_device(cp, out.data_ptr(), count, cnt.data_ptr(), first=first, count=count, index_offset=off)
        torch.cuda.synchronize()
        k = int(cnt[0].item())
        got = out.cpu().numpy()[: k * 8].view(F.MATCH_DTYPE)
        sel = whole[(whole["index"] >= first) & (whole["index"] < first + count)]
        assert (got["index"].astype(np.int64) - off + first).tolist() == sel["index"].tolist()
        assert got["score"].tolist() == sel["score"].tolist()
