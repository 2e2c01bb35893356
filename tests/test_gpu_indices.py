"""Matched indices on the GPU (`fzb_match_list_indices` = Matcher::match_list_indices, src/matcher/mod.rs:234-275): the traced
scorer keeps the score / match matrices in HBM and walks the alignment back on the device (src/smith_waterman/
alignment_iter.rs:35-181).  Bit-exact against the reference's known answers and against the oracle's restatement: same
(index, score, exact) as match_list, same byte positions in the same (reverse) order, same list order."""
import json
import os
import sys

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
IX = json.load(open(os.path.join(G, "indices.json")))
LANES = {64: (64, 64, 32), 32: (32, 32, 16), 16: (16, 16, 8)}


def pair(needle, pf=64, **cfg):
    kw = dict(cfg)
    scoring = kw.pop("scoring", None) or O.DEFAULT_SCORING
    om = O.Matcher(needle, lanes=LANES[pf], scoring=scoring, **kw)
    oi = om.info()
    fc = F.Config(max_typos=kw.get("max_typos", 0), casing=F.CaseMatching[kw.get("casing", "Smart")], unicode=F.UnicodeMatching[kw.get("unicode", "Smart")],
                  sort=F.SortStrategy[kw.get("sort", "ScoreThenIndexAsc")], scoring=F.Scoring(*scoring), matching=F.Matching[kw.get("matching", "Fuzzy")],
                  pf_lanes=oi["pf_lanes"], sw_lanes=oi["sw_lanes"])
    return F.Matcher(needle, fc), om


def tuples(ms):
    return [(m.index, m.score, m.exact, m.indices) for m in ms]


def check(needle, hs, pf=64, ctx="", **cfg):
    fm, om = pair(needle, pf, **cfg)
    got = tuples(fm.match_list_indices(hs))
    want = om.match_list_indices_ordered(hs)
    if got != want:
        bad = [(g, w, hs[w[0]] if w[0] < len(hs) else None) for g, w in zip(got, want) if g != w][:4]
        raise AssertionError(f"{ctx} needle={needle!r} pf={pf} cfg={cfg}: len {len(got)} vs {len(want)}; first diffs (got, want, haystack) {bad}")
    return got


@pytest.mark.parametrize("pf", [64, 16])
def test_reference_known_answers(pf):
    # src/smith_waterman/mod.rs:323-325, 444-476 (the scorer's own indices tests; whole-haystack window = max_typos None)
    for needle, haystack, want, ref in IX["ascii"]:
        got = check(needle, [haystack], pf, ctx=ref, max_typos=None)
        if pf == 16:  # the reference's test backend is 8 x u16 / 16 x u8 lanes
            assert [g[3] for g in got] == [want], ref
    for needle, haystack, start, want, ref in IX["unicode"]:
        if start == 0:
            got = check(needle, [haystack], pf, ctx=ref, max_typos=None)
            if pf == 16:
                assert [g[3] for g in got] == [want], ref
    # src/matcher/mod.rs:605-616: multibyte needle with unicode matching ignored -> every needle byte is a position
    got = check("é", ["xxé"], pf, max_typos=0, unicode="Ignore")
    assert sorted(got[0][3]) == [2, 3]
    # src/matcher/mod.rs:724-735: the empty needle yields every haystack, no positions
    assert tuples(F.Matcher("", F.Config(pf_lanes=64)).match_list_indices(["foo", "bar"])) == [(0, 0, False, []), (1, 0, False, [])]
    assert tuples(F.Matcher("", F.Config(pf_lanes=64, sort=F.SortStrategy.IndexDesc)).match_list_indices(["foo", "bar"])) == [(1, 0, False, []), (0, 0, False, [])]


def _random_list(rng, alpha, needle, n, lengths):
    hs = []
    for _ in range(n):
        L = int(rng.choice(lengths))
        h = [alpha[int(x)] for x in rng.integers(0, len(alpha), L)]
        if L >= len(needle) and rng.random() < 0.7:
            for q, c in zip(np.sort(rng.choice(L, len(needle), replace=False)), needle):
                h[q] = c
        hs.append("".join(h))
    return hs


@pytest.mark.parametrize("pf", [64, 32, 16])
def test_ascii_differential_against_the_oracle(pf):
    rng = np.random.default_rng(100 + pf)
    alpha = "abcABC_-/ 01xyz"
    sorts = ["ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"]
    for it in range(60):
        needle = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(1, 10))))
        hs = _random_list(rng, alpha, needle, 40, [0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 64, 65, 70, 140, 300])
        typos = [None, 0, 1, 2, 3][int(rng.integers(0, 5))]
        got = check(needle, hs, pf, ctx=f"it={it}", max_typos=typos, sort=sorts[it % 4], casing=["Smart", "Ignore", "Respect"][it % 3])
        for index, _, _, ix in got:  # tests/api_properties.rs:116-174 (assert_indices_contract)
            h = hs[index].encode()
            assert all(a > b for a, b in zip(ix[:-1], ix[1:])) and all(0 <= i < len(h) for i in ix) and len(ix) <= len(needle.encode())


@pytest.mark.parametrize("pf", [64, 16])
def test_u16_score_class_and_long_needles(pf):
    rng = np.random.default_rng(7 + pf)
    alpha = "abcdefgh_/ AB"
    for it in range(12):
        needle = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(14, 40))))  # too long for the u8 class
        hs = _random_list(rng, alpha, needle, 30, [40, 64, 65, 100, 129, 500, 1000, 1024])
        check(needle, hs, pf, ctx=f"it={it}", max_typos=[None, 0, 2][it % 3], sort="IndexAsc")


@pytest.mark.parametrize("pf", [64, 16])
def test_unicode_differential_against_the_oracle(pf):
    rng = np.random.default_rng(300 + pf)
    alpha = ["a", "b", "c", "A", "_", " ", "é", "É", "ß", "다", "라", "😀", "x", "1"]
    for it in range(40):
        needle = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(1, 7))))
        hs = _random_list(rng, alpha, list(needle), 30, [0, 1, 4, 8, 15, 16, 17, 30, 40, 70, 150])
        typos = [None, 0, 1, 2][int(rng.integers(0, 4))]
        check(needle, hs, pf, ctx=f"it={it}", max_typos=typos, unicode=["Smart", "Always"][it % 2], sort=["IndexAsc", "ScoreThenIndexDesc"][it % 2])


def test_greedy_fallback_beyond_1024_bytes():
    # src/smith_waterman/algo/mod.rs:55-72: match_greedy's positions, shifted by the trim offset, reversed
    rng = np.random.default_rng(9)
    alpha = "abcdef_/ "
    for needle, typos in (("fade", 0), ("fade", None), ("a_b", 1), ("é_a", 0)):
        hs = _random_list(rng, alpha, list(needle), 12, [1025, 1100, 2000, 5000]) + ["x" * 1500, "fade", "xx" + "y" * 1400 + "fade"]
        check(needle, hs, 64, max_typos=typos, sort="IndexAsc")


@pytest.mark.parametrize("matching", ["Exact", "Prefix", "Suffix", "Substring"])
def test_literal_modes(matching):
    # src/literal/algo.rs:129-155: the whole needle run, reversed
    hs = ["xxabcxx", "abc", "ABC", "abcabc", "xabc", "abcx", "", "ab", "xxé다yy", "é다", "é다é다", "_abc", "aBc_abc"]
    for needle in ("abc", "é다", "a"):
        for sort in ("ScoreThenIndexAsc", "IndexDesc"):
            check(needle, hs, 64, matching=matching, sort=sort)
    assert tuples(F.Matcher("abc", F.Config(matching=F.Matching.Substring, sort=F.SortStrategy.IndexAsc, pf_lanes=64)).match_list_indices(["xxabcxx"]))[0][3] == [4, 3, 2]  # src/literal/mod.rs:177-182


def test_selection_of_a_resident_corpus():
    # the documented use (src/matcher/mod.rs:227-229): match_list over everything, positions for the top of the list only
    rows, ends = synth.fixed_corpus(b"deadbe", 300_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    starts = np.concatenate([[0], ends[:-1]]).astype(np.int64)
    raw = data.tobytes()
    for typos in (0, 1):
        fm, om = pair("deadbe", 64, max_typos=typos)
        top = fm.match_list(cp)[:200]
        sel = top["index"].astype(np.uint32)
        sel = np.concatenate([sel, sel[:5], np.array([0, 1, 2, len(ends) - 1], np.uint32)])  # repeats, non-matching entries, any order
        sub = [raw[int(starts[i]) : int(ends[i])] for i in sel]
        got = tuples(fm.match_list_indices(cp, sel))
        assert got == om.match_list_indices_ordered(sub)
        assert len(got) >= 205
        by_sel = {g[0]: g for g in got}
        for j in range(200):  # same score / exact as match_list reported for that haystack
            assert by_sel[j][1] == int(top["score"][j]) and by_sel[j][2] == bool(top["exact"][j])
    assert fm.match_list_indices(cp, np.zeros(0, np.uint32)) == []
    with pytest.raises(F.FrizbeeError, match="outside the corpus"):
        fm.match_list_indices(cp, np.array([len(ends)], np.uint32))


def test_whole_list_agrees_with_match_list_at_bench_shape():
    # size-independent property at a larger size: (index, score, exact) of match_list_indices == match_list, positions spell the needle
    rows, ends = synth.fixed_corpus(b"deadbe", 200_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    fm = F.Matcher("deadbe", F.Config(pf_lanes=64))
    whole = fm.match_list(cp)
    got = fm.match_list_indices(cp)
    assert [(g.index, g.score, int(g.exact)) for g in got] == [(int(r["index"]), int(r["score"]), int(r["exact"])) for r in whole]
    full = 0
    for g in got:
        h = data[int(g.index) * 32 : int(g.index) * 32 + 32]
        assert all(a > b for a, b in zip(g.indices[:-1], g.indices[1:]))
        if len(g.indices) == 6:
            full += 1
            assert bytes(h[g.indices[::-1]]).lower() == b"deadbe"
    assert full > 0


def _check_expect(got, expect, ref):
    assert len(got) == len(expect), (ref, got)
    for g, e in zip(got, expect):
        assert g[0] == e[0], ref
        if e[1] is not None:
            assert g[2] == e[1], ref
        if len(e) > 2 and e[2] is not None:
            assert g[3] == e[2], (ref, g)
        if len(e) > 3 and e[3] is not None:
            assert sorted(g[3]) == e[3], (ref, g)


@pytest.mark.parametrize("pf", [64, 32, 16])
def test_matcher_level_known_answers(pf):
    # tests/api_properties.rs:429-556, src/matcher/mod.rs:604-643, 745-748: byte offsets of the ORIGINAL haystack after prefilter + trim,
    # incl. the > 1024-byte greedy fallback
    for case in IX["matcher"]:
        cfg = dict(case["config"])
        if "scoring" in cfg:
            cfg["scoring"] = tuple(cfg["scoring"])
        got = check(case["needle"], case["haystacks"], pf, ctx=case["ref"], **cfg) if case["needle"] else tuples(F.Matcher("", F.Config(pf_lanes=pf)).match_list_indices(case["haystacks"]))
        _check_expect(got, case["expect"], case["ref"])


def multi_pair(query_or_patterns, pf=64, **cfg):
    pats_o = O.parse_query(query_or_patterns)
    pats_f = F.parse_query(query_or_patterns)
    om = O.MultiMatcher(pats_o, lanes=LANES[pf], **cfg)
    fc = F.Config(max_typos=cfg.get("max_typos", 0), sort=F.SortStrategy[cfg.get("sort", "ScoreThenIndexAsc")], pf_lanes=pf)
    return F.MultiMatcher(pats_f, fc), om


@pytest.mark.parametrize("pf", [64, 16])
def test_multi_pattern_indices(pf):
    for case in IX["multi"]:  # src/matcher/multi.rs:277-282
        fm, om = multi_pair(case["query"], pf, **case["config"])
        got = tuples(fm.match_list_indices(case["haystacks"]))
        assert got == om.match_list_indices_ordered(case["haystacks"])
        _check_expect(got, case["expect"], case["ref"])
    same = IX["multi_same"]  # src/matcher/multi.rs:253-274
    for query in same["queries"]:
        fm, om = multi_pair(query, pf, **same["config"])
        got = tuples(fm.match_list_indices(same["haystacks"]))
        assert got == om.match_list_indices_ordered(same["haystacks"]), query
        assert [(g[0], g[1], int(g[2])) for g in got] == [(int(r["index"]), int(r["score"]), int(r["exact"])) for r in fm.match_list(same["haystacks"])], query
    rng = np.random.default_rng(500 + pf)
    alpha = "abcAB_ /xé"
    for it in range(40):
        words = ["".join(alpha[int(x)] for x in rng.integers(0, 7, int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 4)))]
        query = " ".join(("!" if rng.random() < 0.25 else "") + ["", "^", "'"][int(rng.integers(0, 3))] + w for w in words)
        hs = ["".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(0, 60)))) for _ in range(40)]
        sort = ["IndexAsc", "ScoreThenIndexDesc", "IndexDesc", "ScoreThenIndexAsc"][it % 4]
        fm, om = multi_pair(query, pf, sort=sort, max_typos=[0, 1, None][it % 3])
        got = tuples(fm.match_list_indices(hs))
        assert got == om.match_list_indices_ordered(hs), (query, sort)
        if it % 5 == 0:  # a selection with repeats
            sel = rng.integers(0, len(hs), 25).astype(np.uint32)
            cp = F.Corpus(hs)
            assert tuples(fm.match_list_indices(cp, sel)) == om.match_list_indices_ordered([hs[int(i)] for i in sel]), (query, sort)
