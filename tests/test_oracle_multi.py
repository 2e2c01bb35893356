"""Multi-pattern composition (fuzzy patterns): the oracle's restatement of src/matcher/multi.rs against the
reference's known answers, and against the reference's own test oracle for it - every pattern matched on its own, then
composed per haystack (tests/api_properties.rs:316-361) - on seeded random cases shaped like its generator (:250-311)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

MU = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multi.json")))


def pats(case):
    return [O.P(n, negated=neg, max_typos=(O.INHERIT if t == "inherit" else t)) for n, neg, t in case["patterns"]]


@pytest.mark.parametrize("lanes", [(64, 64, 32), (16, 16, 8)])
@pytest.mark.parametrize("case", MU["cases"], ids=lambda c: c["name"])
def test_reference_known_answers(case, lanes):
    r = O.MultiMatcher(pats(case), lanes=lanes, **case["config"]).match_list(case["haystacks"])
    if "expect_indices" in case:
        assert r["index"].tolist() == case["expect_indices"], case["ref"]
    if "expect_len" in case:
        assert len(r) == case["expect_len"], case["ref"]
    if "expect_first_index" in case:
        assert int(r[0]["index"]) == case["expect_first_index"], case["ref"]
    if case.get("expect_sorted"):
        s, i = r["score"].astype(int), r["index"].astype(int)
        assert all(a > b or (a == b and x < y) for a, b, x, y in zip(s[:-1], s[1:], i[:-1], i[1:])), case["ref"]
    if "expect_double_of_single" in case:
        single = O.Matcher(case["expect_double_of_single"], lanes=lanes, **case["config"]).match_list(case["haystacks"])
        assert r["index"].tolist() == single["index"].tolist() and r["exact"].tolist() == single["exact"].tolist(), case["ref"]
        assert r["score"].tolist() == (single["score"] * 2).tolist(), case["ref"]


ALPHA = "abcABC_-/ 01xyz"


def random_case(rng):
    npat = 1 + int(rng.integers(0, 3))
    patterns = []
    for _ in range(npat):
        n = int(rng.choice([0, 1, 2, 3, 7, 8, int(rng.integers(0, 9))]))
        needle = "".join(ALPHA[int(x)] for x in rng.integers(0, len(ALPHA), n))
        patterns.append(O.P(needle, negated=bool(rng.integers(0, 2)), max_typos=[O.INHERIT, 0, 1, 2][int(rng.integers(0, 4))]))
    hs = []
    for _ in range(int(rng.choice([0, 1, 2, 7, 8, 15, 16, 24]))):
        L = int(rng.choice([0, 1, 2, 7, 8, 15, 16, 31, 32, 48, 70]))
        h = [ALPHA[int(x)] for x in rng.integers(0, len(ALPHA), L)]
        for p in patterns:  # plant pattern needles so that intersections are not almost always empty
            if p["needle"] and L >= len(p["needle"]) and rng.random() < 0.6:
                pos = np.sort(rng.choice(L, len(p["needle"]), replace=False))
                for q, c in zip(pos, p["needle"]):
                    h[q] = c
        hs.append("".join(h))
    cfg = dict(max_typos=[None, 0, 1, 2][int(rng.integers(0, 4))], casing=["Ignore", "Smart", "Respect"][int(rng.integers(0, 3))])
    return patterns, hs, cfg


def test_sequential_narrowing_equals_the_references_composition_oracle():
    rng = np.random.default_rng(20260925)
    nonempty = 0
    for _ in range(600):
        patterns, hs, cfg = random_case(rng)
        mm = O.MultiMatcher(patterns, sort="IndexAsc", **cfg)
        got, want = mm.match_list(hs), mm.reference_composition(hs)
        assert got.tolist() == want.tolist(), (patterns, hs, cfg)
        nonempty += len(got) > 0
        srt = O.MultiMatcher(patterns, sort="ScoreThenIndexAsc", **cfg).match_list(hs)
        assert sorted(srt.tolist()) == sorted(want.tolist())
        s, i = srt["score"].astype(int), srt["index"].astype(int)
        assert all(a > b or (a == b and x < y) for a, b, x, y in zip(s[:-1], s[1:], i[:-1], i[1:]))
    assert nonempty > 100


def test_score_sum_saturates():
    # Match.score is u16 and the composition adds with saturating_add (multi.rs:44, 139)
    needle = "a" * 60
    sc = [900, 6, 5, 1, 12, 4, 4, 8, 4]
    one = O.Matcher(needle, scoring=sc, sort="IndexAsc").match_list([needle])
    assert int(one[0]["score"]) > 40000
    two = O.MultiMatcher([O.P(needle), O.P(needle)], scoring=sc, sort="IndexAsc").match_list([needle])
    assert int(two[0]["score"]) == 65535
