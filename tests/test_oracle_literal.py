"""Literal matching modes and the query syntax: the oracle's restatement of src/literal/algo.rs and
src/pattern.rs against the reference's known answers (tests/golden/literal.json)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

LT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "literal.json")))


def sub_score(needle, haystack, casing="Ignore"):
    r = O.Matcher(needle, matching="Substring", casing=casing, sort="IndexAsc").match_list([haystack])
    return int(r[0]["score"]) if len(r) else None


@pytest.mark.parametrize("needle,haystack,casing,want,ref", LT["scores"])
def test_substring_scores(needle, haystack, casing, want, ref):
    assert sub_score(needle, haystack, casing) == want, ref


@pytest.mark.parametrize("needle,haystack,casing,ref", LT["matches"])
def test_substring_matches(needle, haystack, casing, ref):
    assert sub_score(needle, haystack, casing) is not None, ref


@pytest.mark.parametrize("a,b,ref", LT["greater"])
def test_score_orderings(a, b, ref):
    assert sub_score(*a) > sub_score(*b), ref


@pytest.mark.parametrize("matching,needle,haystacks,extra,want,all_exact,ref", LT["lists"])
def test_match_lists(matching, needle, haystacks, extra, want, all_exact, ref):
    r = O.Matcher(needle, matching=matching, sort="IndexAsc", **extra).match_list(haystacks)
    assert r["index"].tolist() == want, ref
    if all_exact:
        assert all(r["exact"]), ref


def test_anchored_literal_scores_equal_fuzzy():
    for needle, hay in LT["prefix_equals_fuzzy"]:  # src/literal/mod.rs:96-110
        fuzzy = O.Matcher(needle).match_list([hay])[0]["score"]
        assert O.Matcher(needle, matching="Prefix", sort="IndexAsc").match_list([hay])[0]["score"] == fuzzy, (needle, hay)
    assert O.Matcher("foo", matching="Exact").match_list(["foo"])[0]["score"] == O.Matcher("foo").match_list(["foo"])[0]["score"]  # :112-114


def test_literal_ignores_max_typos_and_lane_width():
    # src/literal/mod.rs:7 ("ignore the max_typos parameter"); src/literal/backend.rs:118-200 (every backend agrees)
    for needle, hay in LT["corpus"]:
        for matching in ("Exact", "Prefix", "Suffix", "Substring"):
            base = O.Matcher(needle, matching=matching, sort="IndexAsc").match_list([hay]).tolist()
            for lanes in ((16, 16, 8), (32, 32, 16)):
                for typos in (0, 2, None):
                    assert O.Matcher(needle, lanes=lanes, matching=matching, max_typos=typos, sort="IndexAsc").match_list([hay]).tolist() == base


@pytest.mark.parametrize("atom,needle,matching,negated", LT["parse_atoms"])
def test_parse_atom(atom, needle, matching, negated):
    got = O.parse_query(atom.replace(" ", "\\ ") if " " in atom and "\\ " not in atom else atom)
    # an atom is parsed on its own in the reference; through parse_query an unescaped space would split it
    assert len(got) == 1
    assert (got[0]["needle"], got[0]["matching"], got[0]["negated"]) == (needle, matching, negated)


@pytest.mark.parametrize("query,needles,ref", LT["parse_queries"])
def test_parse_query(query, needles, ref):
    assert [p["needle"] for p in O.parse_query(query)] == needles, ref


@pytest.mark.parametrize("query,haystacks,cfg,want,ref", LT["multi_queries"])
def test_multi_pattern_queries(query, haystacks, cfg, want, ref):
    r = O.MultiMatcher(O.parse_query(query), **cfg).match_list(haystacks)
    assert sorted(r["index"].tolist()) == want, ref


def test_multi_pattern_with_literal_modes_equals_the_references_composition_oracle():
    # tests/api_properties.rs:250-361 with every matching mode, per pattern and in the config
    rng = np.random.default_rng(4242)
    alpha = "abcABC_-/ 01"
    modes = [None, "Fuzzy", "Exact", "Prefix", "Suffix", "Substring"]
    nonempty = 0
    for _ in range(800):
        pats = []
        for _ in range(1 + int(rng.integers(0, 3))):
            n = int(rng.choice([0, 1, 2, 3, 7, 8]))
            pats.append(O.P("".join(alpha[int(x)] for x in rng.integers(0, len(alpha), n)), negated=bool(rng.integers(0, 2)), matching=modes[int(rng.integers(0, 6))]))
        hs = []
        for _ in range(int(rng.choice([0, 1, 2, 7, 8, 15, 16, 24]))):
            L = int(rng.choice([0, 1, 2, 7, 8, 15, 16, 31, 32, 48]))
            h = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), L))
            for p in pats:
                if p["needle"] and rng.random() < 0.5:
                    k = int(rng.integers(0, 3))
                    h = p["needle"] + h if k == 0 else h + p["needle"] if k == 1 else h[: L // 2] + p["needle"] + h[L // 2 :]
            hs.append(h)
        cfg = dict(max_typos=[None, 0, 1, 2][int(rng.integers(0, 4))], casing=["Ignore", "Smart", "Respect"][int(rng.integers(0, 3))], matching=modes[1 + int(rng.integers(0, 5))])
        mm = O.MultiMatcher(pats, sort="IndexAsc", **cfg)
        got, want = mm.match_list(hs), mm.reference_composition(hs)
        assert got.tolist() == want.tolist(), (pats, hs, cfg)
        nonempty += len(got) > 0
    assert nonempty > 150
