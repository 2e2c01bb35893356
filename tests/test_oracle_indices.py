"""Traceback / matched indices: the oracle's restatement of src/smith_waterman/alignment_iter.rs and
score_haystack[_unicode]_indices (src/smith_waterman/algo/mod.rs:49-152) against the reference's known answers.  The walk
reads the stored score matrix and match masks cell by cell (with diag >= left >= up tie-breaking), so these vectors pin the
matrices the scorer builds, not only their maxima.  (The HIP path's `fzb_match_list_indices` is checked against this
restatement in tests/test_gpu_indices.py.)"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")
IX = json.load(open(os.path.join(G, "indices.json")))
SW = json.load(open(os.path.join(G, "smith_waterman.json")))


@pytest.mark.parametrize("needle,haystack,want,ref", IX["ascii"])
def test_ascii_indices(needle, haystack, want, ref):
    assert O.sw_indices(needle, haystack)[1] == want, ref


@pytest.mark.parametrize("needle,haystack,start,want,ref", IX["unicode"])
def test_unicode_indices(needle, haystack, start, want, ref):
    assert O.sw_indices(needle, haystack, start_pos=start, unicode=True)[1] == want, ref


@pytest.mark.parametrize("needle,haystack,typos,want", IX["score_typos"])
def test_alignment_path_within_typo_budget(needle, haystack, typos, want):
    assert O.sw_score_typos(needle, haystack, typos) == want


def test_indices_contract_on_random_inputs():
    # tests/api_properties.rs:116-174 (assert_indices_contract): same (index, score, exact) as match_list; indices strictly
    # descending, inside the haystack, at most one per needle byte
    rng = np.random.default_rng(31)
    alpha = "abcABC_-/ 01xyz"
    for it in range(400):
        needle = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(1, 9))))
        hs = []
        for _ in range(20):
            L = int(rng.choice([0, 1, 7, 8, 15, 16, 31, 32, 48, 70, 140]))
            h = [alpha[int(x)] for x in rng.integers(0, len(alpha), L)]
            if L >= len(needle) and rng.random() < 0.6:
                for q, c in zip(np.sort(rng.choice(L, len(needle), replace=False)), needle):
                    h[q] = c
            hs.append("".join(h))
        typos = [None, 0, 1, 2][int(rng.integers(0, 4))]
        for lanes in ((16, 16, 8), (64, 64, 32)):
            m = O.Matcher(needle, lanes=lanes, max_typos=typos, sort="IndexAsc")
            recs, idx = m.match_list_indices(hs)
            assert recs.tolist() == m.match_list(hs).tolist()
            for r, ix in zip(recs, idx):
                h = hs[int(r["index"])].encode()
                assert all(a > b for a, b in zip(ix[:-1], ix[1:])) and all(0 <= i < len(h) for i in ix) and len(ix) <= len(needle.encode())


def test_literal_indices_are_the_contiguous_run_reversed():
    recs, idx = O.Matcher("abc", matching="Substring", sort="IndexAsc").match_list_indices(["xxabcxx"])  # src/literal/mod.rs:177-182
    assert idx == [[4, 3, 2]]
    recs, idx = O.Matcher("é다", matching="Substring", sort="IndexAsc").match_list_indices(["xxé다yy"])  # src/literal/mod.rs:354-361
    assert idx == [[6, 5, 4, 3, 2]]


LANES3 = [(64, 64, 32), (32, 32, 16), (16, 16, 8)]


def check_expect(got, expect, ref):
    """got: (index, score, exact, positions) per record; expect: [index, exact | None, positions | None, sorted positions | None]"""
    assert len(got) == len(expect), (ref, got)
    for g, e in zip(got, expect):
        assert g[0] == e[0], ref
        if e[1] is not None:
            assert g[2] == e[1], ref
        if len(e) > 2 and e[2] is not None:
            assert g[3] == e[2], (ref, g)
        if len(e) > 3 and e[3] is not None:
            assert sorted(g[3]) == e[3], (ref, g)


@pytest.mark.parametrize("lanes", LANES3)
@pytest.mark.parametrize("case", IX["matcher"], ids=lambda c: c["ref"])
def test_matcher_match_list_indices_known_answers(case, lanes):
    m = O.Matcher(case["needle"], lanes=lanes, **case["config"])
    got = m.match_list_indices_ordered(case["haystacks"])
    check_expect(got, case["expect"], case["ref"])
    assert [(g[0], g[1], g[2]) for g in got] == [(int(r["index"]), int(r["score"]), bool(r["exact"])) for r in m.match_list(case["haystacks"])]


@pytest.mark.parametrize("lanes", LANES3)
def test_multi_pattern_indices_known_answers(lanes):
    for case in IX["multi"]:
        got = O.MultiMatcher(O.parse_query(case["query"]), lanes=lanes, **case["config"]).match_list_indices_ordered(case["haystacks"])
        check_expect(got, case["expect"], case["ref"])
    same = IX["multi_same"]  # src/matcher/multi.rs:253-274
    for query in same["queries"]:
        mm = O.MultiMatcher(O.parse_query(query), lanes=lanes, **same["config"])
        got = mm.match_list_indices_ordered(same["haystacks"])
        assert [(g[0], g[1], g[2]) for g in got] == [(int(r["index"]), int(r["score"]), bool(r["exact"])) for r in mm.match_list(same["haystacks"])], query
        assert all(all(a > b for a, b in zip(g[3][:-1], g[3][1:])) for g in got), query


def test_multi_pattern_indices_compose_the_single_pattern_ones():
    # match_one_indices_multi (multi.rs:56-82) restated a second way: union of the positive patterns' own position lists
    rng = np.random.default_rng(77)
    alpha = "abcAB_ /x"
    for it in range(150):
        words = ["".join(alpha[int(x)] for x in rng.integers(0, 6, int(rng.integers(1, 4)))) for _ in range(int(rng.integers(1, 4)))]
        neg = [bool(rng.random() < 0.25) for _ in words]
        query = " ".join(("!" if n else "") + w for w, n in zip(words, neg))
        hs = ["".join(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(0, 40)))) for _ in range(30)]
        pats = O.parse_query(query)
        sort = ["IndexAsc", "ScoreThenIndexDesc"][it % 2]
        got = O.MultiMatcher(pats, sort=sort).match_list_indices_ordered(hs)
        mm = O.MultiMatcher(pats, sort=sort)
        assert sorted((g[0], g[1], g[2]) for g in got) == sorted((int(r["index"]), int(r["score"]), bool(r["exact"])) for r in mm.match_list(hs)), query
        singles = [(p, dict((i, ix) for i, _, _, ix in O.Matcher(p["needle"], matching=p["matching"] or "Fuzzy", sort="IndexAsc").match_list_indices_ordered(hs))) for p in pats]
        for index, _, _, ix in got:
            union = set()
            for p, by_index in singles:
                if not p["negated"]:
                    union |= set(by_index[index])
            assert ix == sorted(union, reverse=True), (query, hs[index])
