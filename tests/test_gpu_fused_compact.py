"""Compaction inside the streaming filter (k1_dfa<ET, true>: ticket-ordered tiles, decoupled look-back over epoch-tagged status words) on
lists of haystacks up to 32 bytes: against the oracle, and against the same query with FZB_NO_FUSED_COMPACT=1 (k_compact1 behind the
filter) - list sizes around the tile size and around the grid size (one tile, fewer tiles than workgroups, several tiles per workgroup),
every density (no survivor, every haystack, a dense block in a sparse list), repeated calls on one matcher with changing sizes (stale
status words of earlier launches must not be taken for this launch's), tile-aligned and unaligned sub-ranges, a typo configuration
(LCS automaton) and a unicode needle (both counters written)."""
import os
import random

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _rows(rng, n, needle, density, block=None):
    alphabet = "xyz_-/01QRS"
    out = []
    for i in range(n):
        hit = rng.random() < density or (block and block[0] <= i < block[1])
        L = rng.choice([32, 32, 32, 20, 7, 0]) if n < 5000 else 32
        s = [rng.choice(alphabet) for _ in range(L)]
        if hit and L >= len(needle):
            for q, c in zip(sorted(rng.sample(range(L), len(needle))), needle):
                s[q] = c
        out.append("".join(s))
    return out


def _unfused(fn):
    os.environ["FZB_NO_FUSED_COMPACT"] = "1"
    F.lib().fzb_debug_reload_knobs()
    try:
        return fn()
    finally:
        os.environ.pop("FZB_NO_FUSED_COMPACT", None)
        F.lib().fzb_debug_reload_knobs()


@pytest.mark.parametrize("n", [1, 63, 1023, 1024, 1025, 5 * 1024, 40_000])
@pytest.mark.parametrize("density", [0.0, 0.05, 1.0])
def test_small_lists_against_the_oracle(n, density):
    rng = random.Random(n * 7 + int(density * 100))
    for needle, cfg in (("deadbe", dict()), ("deadbe", dict(max_typos=1)), ("éa", dict())):
        hs = _rows(rng, n, needle, density)
        fm = F.Matcher(needle, F.Config(max_typos=cfg.get("max_typos", 0), pf_lanes=64, sw_lanes=64))
        cp = F.Corpus(hs)
        want = O.Matcher(needle, lanes=(64, 64, 32), **cfg).match_list(hs)
        for _ in range(3):  # the same matcher again: a new epoch over the same status words
            got = fm.match_list(cp)
            assert got.tolist() == want.tolist(), (needle, cfg, n, density)
        assert fm.last_counters()["filter_survivors"] >= len(want)


def test_one_matcher_over_lists_of_changing_size():
    rng = random.Random(99)
    needle = "deadbe"
    fm = F.Matcher(needle, F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    om = O.Matcher(needle, lanes=(64, 64, 32))
    sizes = [300_000, 1500, 2_100_000 // 8, 1, 70_000, 300_000, 1024 * 9]
    corpora = {}
    for n in sizes:
        if n not in corpora:
            hs = _rows(rng, n, needle, 0.03, block=(n // 3, n // 3 + min(n // 10, 5000)))
            corpora[n] = (F.Corpus(hs), om.match_list(hs))
    fm.reserve(corpora[300_000][0])  # one workspace for all of them: the look-back words keep what earlier launches left
    for n in sizes + sizes[::-1]:
        cp, want = corpora[n]
        got = fm.match_list(cp)
        assert got.tolist() == want.tolist(), n


@pytest.mark.parametrize("max_typos", [0, 1])
def test_three_million_against_the_unfused_path_and_sub_ranges(max_typos):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth
    n = 3_000_000  # 2930 tiles: more than the grid's workgroups - every workgroup draws several tickets
    rows, ends = synth.fixed_corpus(b"deadbe", n, 32, device="cuda", seed=77)
    cp = F.Corpus(packed=(rows.cpu().numpy().reshape(-1), ends))
    fm = F.Matcher("deadbe", F.Config(max_typos=max_typos, sort=F.SortStrategy.IndexAsc, pf_lanes=64, sw_lanes=64))
    got = fm.match_list(cp)
    ref = _unfused(lambda: F.Matcher("deadbe", F.Config(max_typos=max_typos, sort=F.SortStrategy.IndexAsc, pf_lanes=64, sw_lanes=64)).match_list(cp))
    assert len(got) > 100_000 and got.tolist() == ref.tolist()
    for first, cnt in ((0, 1024 * 100), (1024 * 7, 1024 * 2048 + 5), (12345, 2_000_001), (n - 999, 999)):
        part = fm.match_list_into(cp, first=first, count=cnt, index_offset=first)
        sl = got[(got["index"] >= first) & (got["index"] < first + cnt)]
        assert sorted(part.tolist()) == sorted(sl.tolist()), (first, cnt)
    # the oracle on a slice (the whole list is covered by tests/test_gpu_full_size.py at 10 M)
    host = rows[:200_000].cpu().numpy().reshape(-1)
    want = O.Matcher("deadbe", lanes=(64, 64, 32), max_typos=max_typos, sort="IndexAsc").match_packed(np.concatenate([host, np.zeros(64, np.uint8)]), ends[:200_000])
    part = fm.match_list_into(cp, first=0, count=200_000, index_offset=0)
    assert sorted(part.tolist()) == sorted(want.tolist())
