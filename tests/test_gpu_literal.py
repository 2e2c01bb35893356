"""Literal matching modes and parsed multi-pattern queries through the C ABI against the oracle:
the reference's known answers, its cross-backend corpus, seeded random cases, a 1 M-haystack list."""
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
LT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "literal.json")))
MODES = ("Exact", "Prefix", "Suffix", "Substring")


def hip(needle, haystacks, matching, casing="Smart", sort="IndexAsc", **kw):
    return F.Matcher(needle, F.Config(matching=F.Matching[matching], casing=F.CaseMatching[casing], sort=F.SortStrategy[sort], **kw)).match_list(haystacks)


@pytest.mark.parametrize("needle,haystack,casing,want,ref", LT["scores"])
def test_substring_scores(needle, haystack, casing, want, ref):
    r = hip(needle, [haystack], "Substring", casing)
    assert (int(r[0]["score"]) if len(r) else None) == want, ref


@pytest.mark.parametrize("matching,needle,haystacks,extra,want,all_exact,ref", LT["lists"])
def test_match_lists(matching, needle, haystacks, extra, want, all_exact, ref):
    r = hip(needle, haystacks, matching, **extra)
    assert r["index"].tolist() == want, ref
    assert r.tolist() == O.Matcher(needle, matching=matching, sort="IndexAsc", **extra).match_list(haystacks).tolist()
    if all_exact:
        assert all(r["exact"]), ref


def test_cross_backend_corpus_and_random_cases():
    cases = [(n, [h]) for n, h in LT["corpus"]]
    rng = np.random.default_rng(99)
    pool = ["a", "b", "A", "B", "_", "-", " ", "0", "é", "É", "다", "ß", "и", "И"]
    for _ in range(300):
        needle = "".join(pool[int(i)] for i in rng.integers(0, len(pool), int(rng.integers(1, 6))))
        hs = []
        for _ in range(int(rng.integers(1, 40))):
            h = "".join(pool[int(i)] for i in rng.integers(0, len(pool), int(rng.choice([0, 1, 3, 8, 15, 16, 17, 31, 33, 70]))))
            k = int(rng.integers(0, 5))
            h = needle + h if k == 0 else h + needle if k == 1 else h[: len(h) // 2] + needle.swapcase() + h[len(h) // 2 :] if k == 2 else h
            hs.append(h)
        cases.append((needle, hs))
    for needle, hs in cases:
        for matching in MODES:
            for casing in ("Smart", "Ignore", "Respect"):
                for sort in ("IndexAsc", "ScoreThenIndexAsc"):
                    want = O.Matcher(needle, matching=matching, casing=casing, sort=sort).match_list(hs)
                    got = hip(needle, hs, matching, casing, sort)
                    assert got.tolist() == want.tolist(), (needle, hs, matching, casing, sort)


@pytest.mark.parametrize("query,haystacks,cfg,want,ref", LT["multi_queries"])
def test_parsed_queries_known_answers(query, haystacks, cfg, want, ref):
    pats = F.parse_query(query)
    fc = F.Config(max_typos=cfg.get("max_typos", 0), sort=F.SortStrategy[cfg.get("sort", "ScoreThenIndexAsc")], pf_lanes=64)
    got = F.MultiMatcher(pats, fc).match_list(haystacks)
    assert sorted(got["index"].tolist()) == want, ref
    assert got.tolist() == O.MultiMatcher(O.parse_query(query), **cfg).match_list(haystacks).tolist()


def test_one_million_haystacks_literal_and_query():
    rows, ends = synth.fixed_corpus(b"deadbe", 1_000_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    odata = np.concatenate([data, np.zeros(64, np.uint8)])
    for needle, matching in (("de", "Substring"), ("d", "Prefix"), ("E", "Suffix"), ("ad", "Substring")):
        want = O.Matcher(needle, matching=matching).match_packed(odata, ends)
        got = F.Matcher(needle, F.Config(matching=F.Matching[matching])).match_list(cp)
        assert len(got) == len(want) and len(got) > 0
        assert np.array_equal(got["index"], want["index"]) and np.array_equal(got["score"], want["score"]) and np.array_equal(got["exact"], want["exact"]), (needle, matching)
    for query in ("dead be !x", "^d 'ea be$ !q", "!a !b !c"):
        want = O.MultiMatcher(O.parse_query(query)).match_packed(odata, ends)
        got = F.MultiMatcher(F.parse_query(query), F.Config(pf_lanes=64)).match_list(cp)
        assert len(got) == len(want) and len(got) > 0, query
        assert np.array_equal(got["index"], want["index"]) and np.array_equal(got["score"], want["score"]) and np.array_equal(got["exact"], want["exact"]), query
