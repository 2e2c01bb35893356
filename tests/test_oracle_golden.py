"""Pins the CPU oracle (oracle/frizbee_oracle.hpp) to the reference's own known-answer tests
(tests/golden/*.json, transcribed from /root/reference's #[test] assertions with file:line)."""
import json
import os

import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")
SW = json.load(open(os.path.join(G, "smith_waterman.json")))
PF = json.load(open(os.path.join(G, "prefilter.json")))
MT = json.load(open(os.path.join(G, "matcher.json")))

# every (LANES, u8?) the reference instantiates: scalar 8/u16, 16/u8; AVX2 16/u16, 32/u8; AVX-512 32/u16, 64/u8
WIDTHS = [(8, False), (16, True), (16, False), (32, True), (32, False), (64, True)]


@pytest.mark.parametrize("v", SW["sw_ascii"], ids=lambda v: f"{v['needle']}|{v['haystack']}")
def test_sw_known_answers_scalar8(v):
    assert O.sw_score(v["needle"], v["haystack"], lanes=8, is_u8=False) == v["score"], v["ref"]


@pytest.mark.parametrize("v", SW["sw_ascii"], ids=lambda v: f"{v['needle']}|{v['haystack']}")
def test_sw_known_answers_all_widths(v):
    # these short cases are single- or few-chunk and hold at every width (parity.rs:95-124 asserts the same for its corpus)
    for lanes, u8 in WIDTHS:
        if u8 and not O.score_fits_in_u8(len(v["needle"].encode())):
            continue
        assert O.sw_score(v["needle"], v["haystack"], lanes=lanes, is_u8=u8) == v["score"], (lanes, u8, v["ref"])


def test_sw_long_haystack_boundary_and_greedy():
    for v in SW["sw_long"]:
        hay = "x" * (v["haystack_len"] - 3) + "abc"
        for lanes, u8 in WIDTHS:
            assert O.sw_score(v["needle"], hay, lanes=lanes, is_u8=u8) == v["score"], (lanes, u8, v["haystack_len"])


def test_sw_case_sensitive():
    for v in SW["sw_case"]:
        assert O.sw_score(v["needle"], v["haystack"], case_sensitive=v["case_sensitive"]) == v["score"], v["ref"]
    # case-sensitive 'A' vs 'a' has no alignment -> score 0 (smith_waterman/mod.rs:357)
    assert O.sw_score("A", "a", case_sensitive=True) == 0


def test_sw_orderings():
    for v in SW["sw_greater"]:
        assert O.sw_score(*v["a"]) > O.sw_score(*v["b"]), v["ref"]


def test_sw_unicode_known_answers():
    for v in SW["sw_unicode"]:
        for lanes, u8 in WIDTHS:
            assert O.sw_score(v["needle"], v["haystack"], unicode=True, lanes=lanes, is_u8=u8) == v["score"], (lanes, u8, v["ref"])
    for v in SW["sw_unicode_equal"]:
        assert O.sw_score(*v["a"], unicode=True) == O.sw_score(*v["b"], unicode=True), v["ref"]


def test_sw_cross_width_corpus():
    for v in SW["sw_cross_width"]:
        want = O.sw_score(v["needle"], v["haystack"], lanes=8, is_u8=False)
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(v["needle"].encode())):
                continue
            assert O.sw_score(v["needle"], v["haystack"], lanes=lanes, is_u8=u8) == want, (lanes, u8, v)


def test_greedy_known_answers():
    for v in SW["greedy"]:
        got = O.greedy(v["needle"], v["haystack"])
        assert max(got, 0) == v["score"], v["ref"]
    hg = SW["greedy_huge_gap"]
    assert O.greedy(hg["needle"], "a" + "x" * hg["x_count"] + "b") == hg["score"], hg["ref"]


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_prefilter_truth_table(lanes):
    for v in PF["pf_bool"]:
        got = O.prefilter(v["needle"], v["haystack"], v["max_typos"], v["case_sensitive"], False, lanes)
        assert got[0] == v["matched"], (v, got)
    for v in PF["pf_unicode_bool"]:
        got = O.prefilter(v["needle"], v["haystack"], v["max_typos"], v["case_sensitive"], True, lanes)
        assert got[0] == v["matched"], (v, got)


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_prefilter_windows(lanes):
    for v in PF["pf_window"]:
        got = O.prefilter(v["needle"], v["haystack"], v["max_typos"], v["case_sensitive"], v["unicode"], lanes)
        assert list(got) == v["window"], (v, got)


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_prefilter_chunk_boundary_sweeps(lanes):
    sw = PF["pf_unicode_prefix_sweep"]
    for p in sw["prefix_lens"]:
        hay = "x" * p + sw["needle"]
        assert O.prefilter(sw["needle"], hay, 0, False, True, lanes) == (True, p, len(hay.encode())), p
    sw = PF["pf_ascii_chunk_sweep"]
    for p in sw["prefix_lens"]:
        hay = "x" * p + "abc"
        for needle, k, want in sw["cases"]:
            assert O.prefilter(needle, hay, k, False, False, lanes)[0] == want, (p, needle, k)
    # back-scan of the final char across chunks (prefilter/mod.rs:338-347, 390-398)
    hay = "xxإن" + "x" * 32 + "نzz"
    end = hay.encode().rfind("ن".encode()) + 2
    assert O.prefilter("إن", hay, 0, False, True, lanes) == (True, 2, end)
    hay = "xxé__😀" + "x" * 32 + "다zz"
    end = hay.encode().rfind("다".encode()) + 3
    assert O.prefilter("é다😀", hay, 1, False, True, lanes) == (True, 2, end)
    # wrong-prefix decoys before a real match (prefilter/mod.rs:288-317)
    fp = "ۥ؆"
    hay = fp + "__إن"
    assert O.prefilter("إن", hay, 0, False, True, lanes) == (True, len(fp.encode()) + 2, len(hay.encode()))


def _expand(hs):
    if isinstance(hs, dict):
        n, patches = hs["haystacks_with"]
        out = ["nomatch-%d" % i for i in range(n)]
        for i, s in patches:
            out[i] = s
        return out
    return hs


@pytest.mark.parametrize("lanes", [(64, 64, 32), (32, 32, 16), (16, 16, 8)])
@pytest.mark.parametrize("case", MT["cases"], ids=lambda c: c["name"])
def test_matcher_end_to_end(case, lanes):
    m = O.Matcher(case["needle"], lanes=lanes, **case["config"])
    hs = _expand(case["haystacks"])
    r = m.match_list(hs)
    if "expect_len" in case:
        assert len(r) == case["expect_len"], case["ref"]
    if "expect_indices" in case:
        assert r["index"].tolist() == case["expect_indices"], case["ref"]
    if "expect_scores" in case:
        assert r["score"].tolist() == case["expect_scores"], case["ref"]
    if "expect_exact_indices" in case:
        assert sorted(r["index"][r["exact"] != 0].tolist()) == case["expect_exact_indices"], case["ref"]
    if "expect_exact_list" in case:
        assert [bool(x) for x in r["exact"]] == case["expect_exact_list"], case["ref"]
    if "expect_exact_map" in case:
        got = {int(i): bool(e) for i, e in zip(r["index"], r["exact"])}
        for k, v in case["expect_exact_map"].items():
            assert got.get(int(k)) == v, (k, case["ref"])
    # match_list_parallel must equal match_list for every thread count (parallel.rs:104-130)
    for t in (1, 2, 3, 8):
        rp = m.match_list_parallel(hs, t)
        assert rp.tolist() == r.tolist(), (t, case["ref"])


@pytest.mark.parametrize("lanes", [(64, 64, 32), (32, 32, 16), (16, 16, 8)])
@pytest.mark.parametrize("case", MT["same_result"], ids=lambda c: c["name"])
def test_configs_the_reference_asserts_equal(case, lanes):
    a = O.Matcher(case["needle"], lanes=lanes, **case["config_a"]).match_list(case["haystacks"])
    b = O.Matcher(case["needle"], lanes=lanes, **case["config_b"]).match_list(case["haystacks"])
    assert len(a) == 1 and a.tolist() == b.tolist(), case["ref"]


def test_readme_smoke_score_is_53():
    # BASELINE.json configs[0]; 53 is hand-derived in SURVEY.md section 8 (not a reference-pinned value)
    r = O.Matcher("fBr").match_list(["fooBar", "foo_bar", "barfoo", "prelude", "println!"])
    assert r.tolist() == [(0, 53, 0, 0)]


def test_guards_and_class_selection():
    for p in MT["panics"]:
        with pytest.raises(RuntimeError, match=p["message_contains"]):
            O.Matcher(p["needle"], scoring=p["scoring"])
    assert O.max_needle_len() == MT["max_needle_len_default"]
    for v in MT["score_fits_in_u8"]:
        assert O.score_fits_in_u8(v["needle_len"], v["scoring"]) == v["fits"], v["ref"]
    # default scoring: u8 class iff needle <= 13 bytes
    assert O.score_fits_in_u8(13) and not O.score_fits_in_u8(14)
    assert O.Matcher("deadbe").info() == dict(pf_lanes=64, sw_lanes=64, use_u8=True)
    assert O.Matcher("a" * 14).info() == dict(pf_lanes=64, sw_lanes=32, use_u8=False)
    with pytest.raises(RuntimeError, match="threads must be positive"):
        O.Matcher("a").match_list_parallel(["a"], 0)  # tests/api_properties.rs:620-624, parallel.rs:24


def test_overflow_guard_uses_char_count_for_unicode_needles():
    # src/matcher/algo.rs:383-393
    needle = "一二三四五六七八"
    r = O.Matcher(needle, scoring=[12, 6, 5, 1, 12, 4000, 4, 8, 4]).match_list([needle])
    assert len(r) == 1


def test_penalty_above_u8_range_is_not_truncated():
    # src/matcher/algo.rs:411-421
    def score(mm):
        return int(O.Matcher("abc", max_typos=1, scoring=[12, mm, 5, 1, 12, 4, 4, 8, 4]).match_list(["aXc"])[0]["score"])
    assert score(260) <= score(255)


def test_sort_strategies_and_radix_sort():
    import numpy as np
    rng = np.random.default_rng(42)
    n = 1 << 16
    arr = np.zeros(n, O.MATCH_DTYPE)
    arr["index"] = np.arange(n)
    arr["score"] = rng.integers(0, 1 << 16, n)
    got = O.radix_sort(arr)
    order = np.lexsort((arr["index"], -arr["score"].astype(np.int64)))
    assert got.tolist() == arr[order].tolist()  # stable, descending (src/sort.rs:47-65)
    hs = _expand({"haystacks_with": [4101, [[0, "abc"], [1, "xabc"], [2047, "abc"], [2048, "a_b_c"], [4096, "abc"], [4100, "zabc"]]]})
    for sort in ("ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"):
        m = O.Matcher("abc", sort=sort)
        seq = m.match_list(hs)
        if sort == "ScoreThenIndexDesc":  # tests/api_properties.rs:683-691
            assert all(a["score"] > b["score"] or (a["score"] == b["score"] and a["index"] > b["index"]) for a, b in zip(seq[:-1], seq[1:]))
        if sort == "IndexDesc":  # tests/api_properties.rs:713-719
            assert all(a["index"] > b["index"] for a, b in zip(seq[:-1], seq[1:]))
        for t in (2, 8):
            assert m.match_list_parallel(hs, t).tolist() == seq.tolist(), (sort, t)


def test_k_merge_known_answer():
    # tests/api_properties.rs:668-681
    import numpy as np
    def mk(pairs):
        a = np.zeros(len(pairs), O.MATCH_DTYPE)
        for i, (s, ix) in enumerate(pairs):
            a[i]["score"], a[i]["index"] = s, ix
        return a
    got = O.k_merge("ScoreThenIndexDesc", [mk([(100, 3), (80, 5), (20, 1)]), mk([(100, 2), (90, 4), (80, 0)])])
    assert [(int(x["score"]), int(x["index"])) for x in got] == [(100, 3), (100, 2), (90, 4), (80, 5), (80, 0), (20, 1)]
