"""The reference's own PROPERTY tests, restated and run against the oracle - the strongest pin there is short of running the
reference: its test-suite holds these for every backend on generated inputs, so a faithful restatement must hold them too.

* prefilter (src/prefilter/mod.rs:407-550, 895-1125): for every typo budget, the accept decision equals the LCS criterion
  `lcs(needle, haystack) + max_typos >= len(needle)` (the reference's `reference_matches_by_deleting_needle_{bytes,chars}`), the
  window is valid, and ALL backends (16 / 32 / 64 lanes) return the identical (matched, start, end) - inputs drawn with the
  reference's own `ByteCursor` generator from random byte strings, like its proptest run;
* Smith-Waterman (src/smith_waterman/backend/tests/parity.rs:95-198): on the fixed corpus every width and score class gives
  the same score AND the same alignment positions as the 8 x u16 scalar backend;
* k-way merge known answers (src/k_merge.rs:187-266), the score-class selection (src/matcher/mod.rs:751-785)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from ref_generators import ByteCursor, SwCursor, api_cases, assert_indices_contract

G = os.path.join(os.path.dirname(__file__), "golden")
SW = json.load(open(os.path.join(G, "smith_waterman.json")))


def lcs_len(needle, haystack, eq):  # src/prefilter/mod.rs:1024-1046
    prev = [0] * (len(haystack) + 1)
    for n in needle:
        cur = [0] * (len(haystack) + 1)
        for i, h in enumerate(haystack):
            cur[i + 1] = prev[i] + 1 if eq(n, h) else max(prev[i + 1], cur[i])
        prev = cur
    return prev[len(haystack)]


def byte_eq(case_sensitive):  # bytes_match, src/prefilter/mod.rs:1048-1050
    def eq(n, h):
        return n == h or (not case_sensitive and n < 128 and h < 128 and chr(n).lower() == chr(h).lower())
    return eq


def check_ascii_case(needle, haystack, max_typos, case_sensitive, want=None):
    oracle = max_typos >= len(needle) or lcs_len(needle, haystack, byte_eq(case_sensitive)) + max_typos >= len(needle)
    if want is not None:
        assert oracle == want, (needle, haystack, max_typos, case_sensitive)
    results = [O.prefilter(needle, haystack, max_typos, case_sensitive, False, lanes) for lanes in (16, 32, 64)]
    for r in results:
        assert r[0] == oracle, (needle, haystack, max_typos, case_sensitive, results)
    if oracle:
        assert results[0] == results[1] == results[2], (needle, haystack, max_typos, case_sensitive, results)  # assert_same_case_result
        assert results[0][1] <= results[0][2] <= len(haystack), results  # assert_valid_window


def check_unicode_case(needle, haystack, max_typos):
    nc, hc = list(needle), list(haystack)
    oracle = max_typos >= len(nc) or lcs_len(nc, hc, lambda a, b: a == b) + max_typos >= len(nc)
    hb = haystack.encode()
    results = [O.prefilter(needle, hb, max_typos, True, True, lanes) for lanes in (16, 32, 64)]
    for r in results:
        assert r[0] == oracle, (needle, haystack, max_typos, results)
    if oracle:
        assert results[0] == results[1] == results[2], (needle, haystack, max_typos, results)
        bounds, p = {0}, 0
        for ch in haystack:
            p += len(ch.encode())
            bounds.add(p)
        assert results[0][1] <= results[0][2] <= len(hb) and results[0][1] in bounds and results[0][2] in bounds, results  # char boundaries


def test_prefilter_manual_cases():
    for needle, haystack, max_typos in [  # backend_parity_suite, src/prefilter/mod.rs:407-430
        ("foo", "foo", 0), ("foo", "oof", 0), ("foo", "f_o_o", 0), ("foo", "f_______________o_______________o", 0), ("\0", "abc", 0), ("a", "", 0), ("bar", "ba", 1),
        ("abc", "c", 2), ("bar", "rb", 1), ("a\0b", "ab", 1), ("abcdef", "abdf", 2), ("abcdef", "fcda", 2), ("abc", "", 3), ("abcdefghij", "abxxcxxdxxe", 5),
        ("abcdefghij", "jihgfedcba", 5), ("abcdefghij", "abc", 8), ("abcdefghijklmnop", "abcxxxdefxxxghixxxjklxxxmnop", 4), ("abcdefghijklmnop", "ponmlkjihgfedcba", 10),
    ]:
        check_ascii_case(needle.encode(), haystack.encode(), max_typos, False)
    for needle, haystack, max_typos, case_sensitive, want in [  # reference_oracle_manual_cases, src/prefilter/mod.rs:433-470
        ("abc", b"", 2, False, False), ("abc", b"", 3, False, True), ("abc", b"bc", 1, False, True), ("abc", b"ac", 1, False, True), ("abc", b"ab", 1, False, True),
        ("abc", b"cba", 2, False, True), ("aaa", b"aa", 1, False, True), ("aba", b"aa", 1, False, True), ("Ab", b"ab", 0, False, True), ("Ab", b"ab", 0, True, False),
        ("A\0b", b"a\0b", 0, False, True), ("A\0b", b"a\0b", 0, True, False), ("éa", "é_a".encode(), 0, False, True), ("ÿA", "ÿa".encode(), 0, False, True),
        ("ÿA", "ÿa".encode(), 0, True, False),
    ]:
        check_ascii_case(needle.encode(), haystack, max_typos, case_sensitive, want)
    for prefix_len in (0, 1, 7, 8, 15, 16, 31, 32, 63, 64):  # reference_oracle_chunk_boundaries, src/prefilter/mod.rs:473-503
        for needle, max_typos, want in (("abc", 0, True), ("ac", 0, True), ("abcd", 0, False), ("abcd", 1, True)):
            check_ascii_case(needle.encode(), b"x" * prefix_len + b"abc", max_typos, False, want)


def test_prefilter_equals_the_lcs_criterion_and_all_widths_agree_on_generated_inputs():
    # randomized_backend_parity_and_oracle, src/prefilter/mod.rs:895-908 + PrefilterCase::from_bytes :697-716 (the reference runs 256 cases)
    rng = np.random.default_rng(20260925)
    for _ in range(1200):
        c = ByteCursor(rng.integers(0, 256, int(rng.integers(0, 2049))).tolist())
        needle_len = max(1, c.len(96, [1, 7, 8, 15, 16, 31, 32, 63, 64]))
        haystack_len = c.len(768, [0, 1, 7, 8, 15, 16, 31, 32, 63, 64, 511, 512, 513])
        max_typos = c.next() % 17
        case_sensitive = c.bool()
        needle = "".join(c.char() for _ in range(needle_len)).encode()
        haystack = bytes(c.byte() for _ in range(haystack_len))
        check_ascii_case(needle, haystack, max_typos, case_sensitive)


def test_unicode_prefilter_equals_the_lcs_criterion_and_all_widths_agree():
    # randomized_unicode_backend_parity_and_oracle, src/prefilter/mod.rs:506-519 + UnicodePrefilterCase::from_bytes :726-748 (128 cases there)
    rng = np.random.default_rng(77)
    for _ in range(800):
        c = ByteCursor(rng.integers(0, 256, int(rng.integers(0, 1025))).tolist())
        needle_len = max(1, c.len(48, [1, 2, 3, 7, 8, 15, 16, 31, 32]))
        haystack_len = c.len(192, [0, 1, 2, 3, 7, 8, 15, 16, 31, 32, 63, 64])
        max_typos = c.next() % 6
        needle = "".join(c.unicode_char() for _ in range(needle_len))
        if needle.isascii():
            needle += "é"
        check_unicode_case(needle, "".join(c.unicode_char() for _ in range(haystack_len)), max_typos)
    for needle in ("aé", "éa", "aébc", "é✓", "✓é", "a✓é", "é😀x", "aXé😀"):  # unicode_mixed_width_matches_oracle, src/prefilter/mod.rs:522-550
        for haystack in ("", "aé", "xaéy", "aébc", "é✓", "a✓é", "zzaé😀xx", "éeé", "aaébcbc", "no match", "aXé😀qw"):
            for max_typos in range(4):
                check_unicode_case(needle, haystack, max_typos)


WIDTHS = [(8, False), (16, False), (32, False), (16, True), (32, True), (64, True)]


def test_alignment_positions_agree_across_widths_on_the_fixed_corpus():
    # cross_backend_parity_indices, src/smith_waterman/backend/tests/parity.rs:193-198 (corpus :95-124)
    for v in SW["sw_cross_width"]:
        want = O.sw_indices(v["needle"], v["haystack"], lanes=8, is_u8=False)
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(v["needle"].encode())):
                continue
            assert O.sw_indices(v["needle"], v["haystack"], lanes=lanes, is_u8=u8) == want, (lanes, u8, v)


def test_generated_inputs_give_valid_positions_and_the_same_score_with_and_without_traceback():
    # randomized_cross_backend_parity, src/smith_waterman/backend/tests/parity.rs:207-334: per backend, score_haystack_indices' score equals
    # score_haystack's and the positions are valid (strictly descending, at most one per needle byte, inside the haystack)
    rng = np.random.default_rng(5)
    for _ in range(400):
        c = SwCursor(rng.integers(0, 256, int(rng.integers(0, 2049))).tolist())
        needle_len = max(1, c.len(96, [1, 7, 8, 15, 16, 31, 32, 63, 64]))
        haystack_len = c.len(768, [0, 1, 7, 8, 15, 16, 31, 32, 63, 64, 1023, 1024, 1025])
        b = c.next() % 5
        max_typos = None if b == 0 else (b - 1) % 17
        case_sensitive = c.bool()
        needle = bytes(c.byte() for _ in range(needle_len)).decode("utf-8", "replace")  # String::from_utf8_lossy
        haystack = bytes(c.byte() for _ in range(haystack_len))
        nb = needle.encode()
        if len(nb) > O.max_needle_len():
            continue
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(nb)):
                continue
            score = O.sw_score(nb, haystack, case_sensitive=case_sensitive, lanes=lanes, is_u8=u8)
            got, ix = O.sw_indices(nb, haystack, max_typos=max_typos, case_sensitive=case_sensitive, lanes=lanes, is_u8=u8)
            assert got == score, (needle, lanes, u8)
            assert all(a > b for a, b in zip(ix[:-1], ix[1:])) and len(ix) <= len(nb) and all(i < len(haystack) for i in ix), (needle, lanes, u8, ix)


def _mk(pairs):
    a = np.zeros(len(pairs), O.MATCH_DTYPE)
    for i, (s, ix) in enumerate(pairs):
        a[i]["score"], a[i]["index"] = s, ix
    return a


def test_k_merge_known_answers():
    pairs = lambda arr: [(int(x["score"]), int(x["index"])) for x in arr]
    # merges_two_sorted_match_runs, src/k_merge.rs:187-208
    got = O.k_merge("ScoreThenIndexAsc", [_mk([(100, 1), (80, 3), (20, 4)]), _mk([(100, 0), (90, 2), (80, 5)])])
    assert pairs(got) == [(100, 0), (100, 1), (90, 2), (80, 3), (80, 5), (20, 4)]
    # heap_merge_skips_empty_runs, :210-233
    got = O.k_merge("ScoreThenIndexAsc", [_mk([(90, 2)]), _mk([]), _mk([(100, 0), (80, 4)]), _mk([]), _mk([(95, 1), (80, 3)])])
    assert pairs(got) == [(100, 0), (95, 1), (90, 2), (80, 3), (80, 4)]
    # heap_merge_handles_many_runs, :235-251
    got = O.k_merge("ScoreThenIndexAsc", [_mk([(100 - r, r)]) for r in reversed(range(17))])
    assert pairs(got) == [(100 - i, i) for i in range(17)]
    # merges_by_index_only, :253-266
    got = O.k_merge("IndexAsc", [_mk([(100, 1), (80, 3), (20, 5)]), _mk([(100, 0), (90, 2), (80, 4)])])
    assert [int(x["index"]) for x in got] == [0, 1, 2, 3, 4, 5]


def test_score_class_selection():
    # u8_path_selected_for_short_needle / u16_path_selected_for_long_needle, src/matcher/mod.rs:751-785
    assert O.Matcher("abc").info()["use_u8"] is True
    assert O.Matcher("abcdefghijklmnopqrst").info()["use_u8"] is False


@pytest.mark.parametrize("lanes", [(64, 64, 32), (16, 16, 8)])
def test_generated_public_api_properties(lanes):
    # generated_public_api_properties, tests/api_properties.rs:72-114 (1024 generated cases there)
    for needle, haystacks, cfg in api_cases(300, 11):
        m = O.Matcher(needle, lanes=lanes, **cfg)
        one_shot = m.match_list(haystacks)
        for threads in (1, 2, 3, 8):  # sorted: identical; unsorted: the same multiset (here even the same order)
            assert m.match_list_parallel(haystacks, threads).tolist() == one_shot.tolist(), (needle, cfg, threads)
        assert_indices_contract(needle, haystacks, cfg, one_shot, m.match_list_indices_ordered(haystacks))


@pytest.mark.parametrize("lanes", [(64, 64, 32), (16, 16, 8)])
def test_generated_multi_pattern_properties(lanes):
    # generated_multi_pattern_properties / assert_multi_pattern_case, tests/api_properties.rs:311-416 (512 generated cases there):
    # Matcher::from_patterns == every pattern matched on its own, intersected / subtracted per haystack
    from ref_generators import multi_cases
    for patterns, haystacks, cfg in multi_cases(300, 23):
        pats = [O.P(p["needle"], negated=p["negated"], matching=p["matching"]) for p in patterns]
        by_index = O.MultiMatcher(pats, lanes=lanes, sort="IndexAsc", **cfg)
        reference = by_index.reference_composition(haystacks)
        assert by_index.match_list(haystacks).tolist() == reference.tolist(), (patterns, cfg)
        sorted_ = O.MultiMatcher(pats, lanes=lanes, sort="ScoreThenIndexAsc", **cfg).match_list(haystacks)
        assert all(a["score"] > b["score"] or (a["score"] == b["score"] and a["index"] < b["index"]) for a, b in zip(sorted_[:-1], sorted_[1:])), (patterns, cfg)
        assert sorted(sorted_.tolist()) == sorted(reference.tolist()), (patterns, cfg)


# ---- a second, independent transcription of match_haystack_1_typo (src/prefilter/algo/ascii_typos.rs:14-118), lane count as a parameter ----
def _one_typo(needle, hay, lanes, case_sensitive):
    def pair(c):
        if case_sensitive:
            return (c, c)
        return (c, c - 32) if 97 <= c <= 122 else (c, c + 32) if 65 <= c <= 90 else (c, c)  # case_needle, src/prefilter/mod.rs:49-65

    def occ(chunk, p):
        m = 0
        for i, b in enumerate(chunk):
            if b == p[0] or b == p[1]:
                m |= 1 << i
        return m

    def clear_through_lowest(mask, matches):
        low = matches & -matches
        return mask & ~((low << 1) - 1)

    nd = [pair(c) for c in needle]
    n, ln = len(nd), len(hay)
    if n <= 1:
        return True
    if ln == 0:
        return False
    first, second = 0, 1
    for start in range(0, ln, lanes):
        chunk = list(hay[start : start + lanes])
        chunk_mask = (1 << len(chunk)) - 1
        chunk += [0] * (lanes - len(chunk))
        first_mask, second_mask = occ(chunk, nd[first]), occ(chunk, nd[second])
        first_cm = second_cm = chunk_mask
        while True:
            advanced = False
            cand = first + 1
            if cand > second:
                if cand == n:
                    return True
                second, second_cm = cand, first_cm
                second_mask = occ(chunk, nd[second])
            elif cand == second and first_cm > second_cm:
                second_cm = first_cm
            hits = first_mask & first_cm
            if hits:
                first += 1
                first_cm = clear_through_lowest(first_cm, hits)
                first_mask = occ(chunk, nd[first])
                advanced = True
            hits = second_mask & second_cm
            if hits:
                second += 1
                if second >= n:
                    return True
                second_cm = clear_through_lowest(second_cm, hits)
                second_mask = occ(chunk, nd[second])
                advanced = True
            if not advanced:
                break
    return False


# inputs on which the reference's greedy two-path scan REJECTS at 32 lanes although LCS + 1 >= len(needle) (and 16 / 64 lanes accept):
# its own `randomized_backend_parity_and_oracle` would fail here.  Found by oracle/selfcheck.cpp (~1e-5 of random cases); the HIP path
# follows the reference, not the LCS criterion (tests/test_gpu_parity.py runs the same inputs).
LCS_DEVIATIONS_1_TYPO = [
    ("aa_CB-bA A", "b/-cbcCCc0_C_ cAAc- Ac/_0a0b_b _CBaB0c a_cAbC0A B-CA0c-//aBcbCa0"),
    ("AA _C/0C-C", "B0_bB1a0CC1_0_-/00aACbAAba/C_ bCc_c -A00-Bb acCb-0_b/1Abc-0bCaaA 0a"),
    ("x__x-Cc- 1", "x1/-C/cBcx10bCA_00CxBbac/11xCzz/Az _xBb1CzCxcCCaA_xzCa- zb _z-C-a01x/c xC-"),
]


def test_one_typo_prefilter_against_an_independent_transcription():
    for needle, hay in LCS_DEVIATIONS_1_TYPO:
        n, h = needle.encode(), hay.encode()
        assert lcs_len(n, h, byte_eq(False)) + 1 >= len(n)
        assert [_one_typo(n, h, lanes, False) for lanes in (16, 32, 64)] == [True, False, True]
        assert [O.prefilter(n, h, 1, False, False, lanes)[0] for lanes in (16, 32, 64)] == [True, False, True]
    rng = np.random.default_rng(99)
    alpha = b"abcABC_-/ 01xyz"
    for _ in range(4000):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 12))))
        hay = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 140]))))
        cs = bool(rng.integers(0, 2))
        for lanes in (16, 32, 64):
            assert O.prefilter(needle, hay, 1, cs, False, lanes)[0] == _one_typo(needle, hay, lanes, cs), (needle, hay, cs, lanes)


def test_smith_waterman_against_an_independent_second_transcription():
    # both are transcriptions of src/smith_waterman/algo/ascii.rs; they were written separately, so a slip in one shows up here
    import sw_second_transcription as T2
    rng = np.random.default_rng(2024)
    alpha = b"abcABC_-/ 01xyzXYZ."
    for it in range(600):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 14))))
        hay = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([0, 1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 140]))))
        if rng.random() < 0.6 and len(hay) >= len(needle):  # plant the needle as a subsequence
            h = bytearray(hay)
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                h[q] = c
            hay = bytes(h)
        scoring = list(O.DEFAULT_SCORING)
        if it % 3 == 1:  # random small constants (still inside the overflow guards of both classes)
            scoring = [int(rng.integers(1, 17)), int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 4)), int(rng.integers(0, 17)),
                       int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 17)), int(rng.integers(0, 9))]
        cs, prefix = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(needle), scoring):
                continue
            want = T2.score_haystack(needle, hay, scoring, cs, prefix, lanes, 8 if u8 else 16)
            got = O.sw_score(needle, hay, scoring=scoring, case_sensitive=cs, include_prefix=prefix, lanes=lanes, is_u8=u8)
            assert got == want, (needle, hay, scoring, cs, prefix, lanes, u8, got, want)


def test_the_second_transcription_is_itself_pinned_to_the_reference_known_answers():
    import sw_second_transcription as T2
    sc = list(O.DEFAULT_SCORING)
    for v in SW["sw_ascii"]:  # src/smith_waterman/mod.rs:209-345, the 8 x u16 scalar backend; these cases hold at every width
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(v["needle"].encode())):
                continue
            assert T2.score_haystack(v["needle"].encode(), v["haystack"].encode(), sc, False, True, lanes, 8 if u8 else 16) == v["score"], (v, lanes, u8)
    for v in SW["sw_case"]:
        assert T2.score_haystack(v["needle"].encode(), v["haystack"].encode(), sc, v["case_sensitive"], True, 8, 16) == v["score"], v
    for v in SW["sw_greater"]:
        a, b = (T2.score_haystack(n.encode(), h.encode(), sc, False, True, 8, 16) for n, h in (v["a"], v["b"]))
        assert a > b, v
    for v in SW["sw_cross_width"]:  # parity.rs:95-124
        want = T2.score_haystack(v["needle"].encode(), v["haystack"].encode(), sc, False, True, 8, 16)
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(v["needle"].encode())):
                continue
            assert T2.score_haystack(v["needle"].encode(), v["haystack"].encode(), sc, False, True, lanes, 8 if u8 else 16) == want, (v, lanes, u8)


def test_traceback_against_the_second_transcription():
    import sw_second_transcription as T2
    rng = np.random.default_rng(31337)
    alpha = b"abcABC_-/ 01xyz"
    sc = list(O.DEFAULT_SCORING)
    for it in range(500):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 11))))
        hay = bytearray(alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64, 65, 100]))))
        if rng.random() < 0.7 and len(hay) >= len(needle):
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                hay[q] = c
        hay = bytes(hay)
        cs = bool(rng.integers(0, 2))
        max_typos = [None, 0, 1, 3][int(rng.integers(0, 4))]
        start_pos = int(rng.integers(0, 3))
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(len(needle), sc):
                continue
            bits = 8 if u8 else 16
            mats = {}
            score = T2.score_haystack(needle, hay, sc, cs, start_pos == 0, lanes, bits, mats)
            want = T2.alignment_indices(len(needle), mats, lanes, bits, score, max_typos, start_pos) if score else []
            got = O.sw_indices(needle, hay, start_pos=start_pos, max_typos=max_typos, case_sensitive=cs, lanes=lanes, is_u8=u8)
            assert got == (score, want), (needle, hay, cs, max_typos, start_pos, lanes, u8, got, (score, want))


def test_unicode_scorer_and_traceback_against_the_second_transcription():
    import sw_second_transcription as T2
    sc = list(O.DEFAULT_SCORING)
    for v in SW["sw_unicode"]:  # the second transcription is pinned to src/smith_waterman/mod.rs:229-235 first
        assert T2.score_haystack_unicode(v["needle"], v["haystack"].encode(), sc, False, True, 8, 16) == v["score"], v
    IXG = json.load(open(os.path.join(G, "indices.json")))
    for needle, haystack, start, want, ref in IXG["unicode"]:  # src/smith_waterman/mod.rs:453-506
        mats = {}
        score = T2.score_haystack_unicode(needle, haystack.encode(), sc, False, start == 0, 8, 16, mats)
        assert T2.unicode_indices(needle, haystack.encode(), mats, 8, 16, score, None, start) == want, ref
    rng = np.random.default_rng(4242)
    alpha = ["a", "b", "c", "A", "B", "_", " ", "/", "é", "É", "ß", "ж", "Ж", "다", "라", "😀", "1"]
    for it in range(500):
        asz = int(rng.integers(3, len(alpha) + 1))
        needle = "".join(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 8))))
        hay = [alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([0, 1, 3, 7, 8, 15, 16, 17, 30, 33, 64, 70])))]
        if rng.random() < 0.7 and len(hay) >= len(needle):
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                hay[q] = c
        hay = "".join(hay).encode()
        scoring = sc
        if it % 3 == 1:
            scoring = [int(rng.integers(1, 17)), int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 4)), int(rng.integers(0, 17)),
                       int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 17)), int(rng.integers(0, 9))]
        cs = bool(rng.integers(0, 2))
        max_typos = [None, 0, 1, 3][int(rng.integers(0, 4))]
        start_pos = int(rng.integers(0, 3))
        nchars = len(needle)
        for lanes, u8 in WIDTHS:
            if u8 and not O.score_fits_in_u8(nchars, scoring):
                continue
            bits = 8 if u8 else 16
            mats = {}
            score = T2.score_haystack_unicode(needle, hay, scoring, cs, start_pos == 0, lanes, bits, mats)
            assert O.sw_score(needle, hay, scoring=scoring, case_sensitive=cs, include_prefix=start_pos == 0, unicode=True, lanes=lanes, is_u8=u8) == score, (needle, hay, scoring, cs, lanes, u8)
            want = T2.unicode_indices(needle, hay, mats, lanes, bits, score, max_typos, start_pos, cs) if score else []
            got = O.sw_indices(needle, hay, start_pos=start_pos, unicode=True, max_typos=max_typos, scoring=scoring, case_sensitive=cs, lanes=lanes, is_u8=u8)
            assert got == (score, want), (needle, hay, scoring, cs, max_typos, start_pos, lanes, u8, got, (score, want))


def test_ascii_prefilter_windows_against_the_second_transcription():
    # decision AND window (start, end) of the whole ASCII prefilter family, every typo budget, all three lane widths
    import pf_second_transcription as P2
    for needle, hay in LCS_DEVIATIONS_1_TYPO:  # the second transcription reproduces the reference's 32-lane deviation too
        assert [P2.prefilter(needle.encode(), hay.encode(), 1, False, lanes)[0] for lanes in (16, 32, 64)] == [True, False, True]
    rng = np.random.default_rng(1234)
    alpha = b"abcABC_-/ 01xyz\0"
    for it in range(3000):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 13))))
        hay = bytearray(alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([0, 1, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 140, 200]))))
        if rng.random() < 0.5 and len(hay) >= len(needle):
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                hay[q] = c
        hay = bytes(hay)
        cs = bool(rng.integers(0, 2))
        max_typos = int(rng.integers(0, 6))
        for lanes in (16, 32, 64):
            want = P2.prefilter(needle, hay, max_typos, cs, lanes)
            got = O.prefilter(needle, hay, max_typos, cs, False, lanes)
            assert got[0] == want[0] and (not want[0] or got == want), (needle, hay, max_typos, cs, lanes, got, want)


def test_unicode_prefilter_windows_against_the_second_transcription():
    import pf_second_transcription as P2
    rng = np.random.default_rng(4321)
    alpha = ["a", "b", "A", "_", " ", "é", "É", "ß", "ж", "Ж", "다", "라", "😀", "✓", "1"]
    n_acc = 0
    for it in range(2500):
        asz = int(rng.integers(3, len(alpha) + 1))
        needle = "".join(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 9))))
        hay = [alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([0, 1, 2, 5, 8, 15, 16, 17, 31, 33, 64, 70, 130])))]
        if rng.random() < 0.5 and len(hay) >= len(needle):
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                hay[q] = c
        hay = "".join(hay).encode()
        cs = bool(rng.integers(0, 2))
        max_typos = int(rng.integers(0, 5))
        for lanes in (16, 32, 64):
            want = P2.prefilter_unicode(needle, hay, max_typos, cs, lanes)
            got = O.prefilter(needle, hay, max_typos, cs, True, lanes)
            assert got[0] == want[0] and (not want[0] or got == want), (needle, hay, max_typos, cs, lanes, got, want)
            n_acc += want[0]
    assert n_acc > 2000


def test_the_prefilter_transcription_is_itself_pinned_to_the_reference_known_answers():
    import pf_second_transcription as P2
    PFG = json.load(open(os.path.join(G, "prefilter.json")))
    for lanes in (16, 32, 64):
        for v in PFG["pf_bool"]:  # src/prefilter/mod.rs:187-270
            assert P2.prefilter(v["needle"].encode(), v["haystack"].encode(), v["max_typos"], v["case_sensitive"], lanes)[0] == v["matched"], v
        for v in PFG["pf_unicode_bool"]:
            assert P2.prefilter_unicode(v["needle"], v["haystack"].encode(), v["max_typos"], v["case_sensitive"], lanes)[0] == v["matched"], v
        for v in PFG["pf_window"]:  # src/prefilter/mod.rs:273-400
            f = P2.prefilter_unicode if v["unicode"] else P2.prefilter
            n = v["needle"] if v["unicode"] else v["needle"].encode()
            assert list(f(n, v["haystack"].encode(), v["max_typos"], v["case_sensitive"], lanes)) == v["window"], (v, lanes)


def test_greedy_fallback_against_the_second_transcription():
    import sw_second_transcription as T2
    sc = list(O.DEFAULT_SCORING)
    for v in SW["greedy"]:  # src/smith_waterman/greedy.rs:112-190
        r = T2.match_greedy(v["needle"].encode(), v["haystack"].encode(), sc, False, True)
        assert (r[0] if r else 0) == v["score"], v
    rng = np.random.default_rng(808)
    alpha = b"abcABC_-/ 01xyz"
    for it in range(300):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 9))))
        hay = bytearray(alpha[int(x)] for x in rng.integers(0, asz, int(rng.choice([1025, 1030, 1500, 3000]))))
        if rng.random() < 0.8:
            for q, c in zip(np.sort(rng.choice(len(hay), len(needle), replace=False)), needle):
                hay[q] = c
        hay = bytes(hay)
        cs, prefix = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        scoring = sc if it % 2 else [int(rng.integers(1, 17)), int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 4)), int(rng.integers(0, 17)),
                                     int(rng.integers(0, 9)), int(rng.integers(0, 9)), int(rng.integers(0, 17)), int(rng.integers(0, 9))]
        r = T2.match_greedy(needle, hay, scoring, cs, prefix)
        # beyond 1024 bytes the scorer IS match_greedy (algo/ascii.rs:11-21, algo/mod.rs:55-72): score, and positions reversed + offset
        assert O.sw_score(needle, hay, scoring=scoring, case_sensitive=cs, include_prefix=prefix, lanes=16, is_u8=False) == (r[0] if r else 0), (needle, cs, prefix)
        start_pos = 0 if prefix else 3
        got = O.sw_indices(needle, hay, start_pos=start_pos, scoring=scoring, case_sensitive=cs, lanes=16, is_u8=False)
        assert got == ((r[0], [p + start_pos for p in reversed(r[1])]) if r else (0, [])), (needle, cs, prefix)


def test_ascii_typo_windows_have_a_lane_free_form():
    # What the width-independence the reference asserts amounts to for k >= 1 typos (ASCII): whenever the prefilter accepts,
    #   start = the earliest first occurrence of needle[0..=k] (either case),
    #   end   = one past the last occurrence of any of needle[n-1-k..] (find_end_pos_with_typos, ascii_typos.rs:374-398), len if none.
    # (0 typos: first occurrence of needle[0], last occurrence of needle[n-1]: SURVEY 8 a7.)  DESIGN section 7 builds on this.
    rng = np.random.default_rng(5)
    alpha = b"abcdefAB_- 01"

    def variants(c, cs):
        if cs:
            return (c,)
        return (c, c - 32) if 97 <= c <= 122 else (c, c + 32) if 65 <= c <= 90 else (c,)

    checked = 0
    for _ in range(6000):
        asz = int(rng.integers(2, len(alpha) + 1))
        n = int(rng.integers(2, 10))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, n))
        ln = int(rng.choice([5, 16, 31, 32, 33, 64, 70, 130]))
        hay = bytearray(alpha[int(x)] for x in rng.integers(0, asz, ln))
        if rng.random() < 0.6 and ln >= n:
            for q, c in zip(np.sort(rng.choice(ln, n, replace=False)), needle):
                hay[q] = c
        hay = bytes(hay)
        cs, k = bool(rng.integers(0, 2)), int(rng.integers(1, 4))
        if k >= n:
            continue
        firsts = [min((p for p in (hay.find(bytes([v])) for v in variants(needle[j], cs)) if p >= 0), default=None) for j in range(k + 1)]
        last_set = {v for j in range(n - 1 - k, n) for v in variants(needle[j], cs)}
        end = max((i for i, b in enumerate(hay) if b in last_set), default=len(hay) - 1) + 1
        for lanes in (16, 32, 64):
            w = O.prefilter(needle, hay, k, cs, False, lanes)
            if w[0]:
                assert (w[1], w[2]) == (min(f for f in firsts if f is not None), end), (needle, hay, k, cs, lanes, w)
                checked += 1
    assert checked > 5000


def test_the_typo_prefilter_only_ever_deviates_on_marginal_inputs():
    # Where the reference's multi-path scan disagrees with its LCS oracle (see LCS_DEVIATIONS_1_TYPO) the input is always MARGINAL:
    # LCS + k == n exactly.  One spare (LCS + k >= n + 1) is accepted at every width; LCS + k < n is rejected at every width.
    # (3 M random cases in a C++ harness: 149 deviations, all with zero slack.)  DESIGN section 7 uses this to confine the
    # lane-exact decision kernel to the marginal survivors.
    rng = np.random.default_rng(606)
    alpha = b"abcABC_-/ 01xyz"
    n_spare = n_reject = 0
    for _ in range(12000):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(2, 14))))
        hay = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(0, 150))))
        cs, k = bool(rng.integers(0, 3) == 0), int(rng.integers(1, 5))
        if k >= len(needle):
            continue
        slack = lcs_len(needle, hay, byte_eq(cs)) + k - len(needle)
        if slack == 0:
            continue
        for lanes in (16, 32, 64):
            assert O.prefilter(needle, hay, k, cs, False, lanes)[0] == (slack > 0), (needle, hay, k, cs, lanes, slack)
        n_spare += slack > 0
        n_reject += slack < 0
    assert n_spare > 3000 and n_reject > 500


def test_single_chunk_typo_prefilter_is_the_lcs_criterion():
    # A haystack that fits one prefilter chunk (len <= lanes) is decided exactly by LCS + k >= n at that width: inside one chunk a
    # path that finds nothing more is stuck for good, so the path above it never gets more than one needle byte ahead of a path that
    # still produces candidates, and the catch-up comparison (ascii_typos.rs:45-62) sees every one of them.  The known deviations
    # (LCS_DEVIATIONS_1_TYPO, tests/test_gpu_api_properties.py SHORT_LCS_DEVIATIONS) all span several chunks.  The product uses this to
    # skip the decide pass when the corpus' longest haystack fits a chunk (host.hip, typo fast path); oracle/single_chunk_check.cpp ran
    # 1.26e8 such cases (1.45e7 marginal) without a deviation, this is the same check at CI size, marginal inputs favoured.
    rng = np.random.default_rng(707)
    alpha = b"abcABC_-/ 01xyz"
    marginal = 0
    for _ in range(30000):
        asz = int(rng.integers(2, len(alpha) + 1))
        needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, int(rng.integers(2, 13))))
        cs, k = bool(rng.integers(0, 3) == 0), int(rng.integers(1, 5))
        if k >= len(needle):
            continue
        for lanes in (16, 32, 64):
            ln = int(rng.integers(1, lanes + 1)) if rng.integers(0, 3) else lanes - int(rng.integers(0, 3))
            hay = bytes(alpha[int(x)] for x in rng.integers(0, asz, ln))
            slack = lcs_len(needle, hay, byte_eq(cs)) + k - len(needle)
            marginal += slack == 0
            assert O.prefilter(needle, hay, k, cs, False, lanes)[0] == (slack >= 0), (needle, hay, k, cs, lanes, slack)
    assert marginal > 5000


def test_unicode_typo_windows_have_the_same_lane_free_form():
    # scalars instead of bytes: start = earliest first occurrence of needle scalars [0..=k] (either case variant), end = the latest end
    # of a last occurrence of the scalars [n-1-k..] (find_end_pos_with_unicode_typos, unicode_typos.rs:483-509)
    import pf_second_transcription as P2
    rng = np.random.default_rng(9)
    alpha = ["a", "b", "A", "_", " ", "é", "É", "ж", "Ж", "다", "😀", "1"]
    checked = 0
    for _ in range(4000):
        asz = int(rng.integers(3, len(alpha) + 1))
        n = int(rng.integers(2, 8))
        needle = "".join(alpha[int(x)] for x in rng.integers(0, asz, n))
        ln = int(rng.choice([4, 10, 16, 20, 33, 50, 70]))
        hay = [alpha[int(x)] for x in rng.integers(0, asz, ln)]
        if rng.random() < 0.6 and ln >= n:
            for q, c in zip(np.sort(rng.choice(ln, n, replace=False)), needle):
                hay[q] = c
        hay = "".join(hay).encode()
        cs, k = bool(rng.integers(0, 2)), int(rng.integers(1, 4))
        if k >= n:
            continue
        chars = P2.case_needle_unicode(needle, cs)
        firsts = [p for j in range(k + 1) for p in (hay.find(v) for v in set(chars[j])) if p >= 0]
        ends = [p + len(v) for j in range(n - 1 - k, n) for v in set(chars[j]) for p in (hay.rfind(v),) if p >= 0]
        for lanes in (16, 32, 64):
            w = O.prefilter(needle, hay, k, cs, True, lanes)
            if w[0]:
                assert (w[1], w[2]) == (min(firsts), max(ends) if ends else len(hay)), (needle, hay, k, cs, lanes, w)
                checked += 1
    assert checked > 5000


# ---- unicode typo queries: the scalar-level LCS criterion (round 6: what the streaming filter decides for them) -------------------------
def scalar_events(chars, hay):
    """Needle rows that OCCUR at every byte position of `hay`: row i occurs at p iff hay[p : p + len_i] equals the scalar's bytes or its case
    flip's (unicode_char_mask, src/prefilter/algo/unicode.rs:74-117: a lane is a byte position, nothing asks whether it is a scalar start).
    Occurrences never overlap (a needle scalar is valid UTF-8: lead byte, then continuation bytes), so the events are a sequence."""
    ev = []
    for p in range(len(hay)):
        rows = frozenset(i for i, (a, b) in enumerate(chars) if hay[p : p + len(a)] in (a, b))
        if rows:
            ev.append(rows)
    return ev


def scalar_lcs(chars, hay):
    ev = scalar_events(chars, hay)
    return lcs_len(list(range(len(chars))), ev, lambda i, rows: i in rows)


UNI_ALPHA = ["a", "b", "A", "_", " ", "é", "É", "ж", "Ж", "다", "😀", "1", "ن", "إ"]


def _uni_text(rng, asz, nbytes, exact_len=False):
    """valid UTF-8 of at most `nbytes` bytes from the first `asz` scalars of the alphabet"""
    out, size = [], 0
    while True:
        c = UNI_ALPHA[int(rng.integers(0, asz))]
        if size + len(c.encode()) > nbytes:
            break
        out.append(c)
        size += len(c.encode())
        if not exact_len and rng.random() < 0.03:
            break
    return "".join(out).encode()


def test_single_chunk_unicode_typo_prefilter_is_the_scalar_lcs_criterion():
    # The unicode typo algorithms (unicode_typos.rs:15-466) are the ASCII ones over occurrence masks of whole scalars, so the single-chunk
    # argument of DESIGN.md "Typo configurations" carries over: a haystack that fits ONE prefilter chunk is accepted iff
    # LCS(needle scalars, haystack's scalar occurrences) + k >= n.  Valid UTF-8 and arbitrary bytes (the C ABI takes any bytes; the
    # reference compares bytes position by position).  The product uses this to decide unicode typo queries over single-chunk lists
    # in the streaming filter (host.hip, build_lcs_dfa / pipe_unicode_typo_fast_path).
    import pf_second_transcription as P2
    rng = np.random.default_rng(808)
    marginal = n_cases = 0
    for it in range(9000):
        asz = int(rng.integers(3, len(UNI_ALPHA) + 1))
        needle = "".join(UNI_ALPHA[int(x)] for x in rng.integers(0, asz, int(rng.integers(2, 9))))
        if needle.isascii():
            needle += "é"
        cs, k = bool(rng.integers(0, 3) == 0), int(rng.integers(1, 4))
        chars = P2.case_needle_unicode(needle, cs)
        if k >= len(chars):
            continue
        for lanes in (16, 32, 64):
            ln = int(rng.integers(1, lanes + 1)) if rng.integers(0, 3) else lanes - int(rng.integers(0, 3))
            if it % 5 == 4:  # arbitrary bytes: needle scalars planted between random bytes, continuation bytes without a lead, truncated scalars
                pool = [a for a, _ in chars] + [bytes([int(rng.integers(0, 256))]) for _ in range(6)] + [b"\x80", b"\xd0", b"\xf0\x9f"]
                hay = b"".join(pool[int(rng.integers(0, len(pool)))] for _ in range(ln))[:ln]
            else:
                hay = _uni_text(rng, asz, ln, exact_len=bool(rng.integers(0, 2)))
            slack = scalar_lcs(chars, hay) + k - len(chars)
            marginal += slack == 0
            n_cases += 1
            assert O.prefilter(needle, hay, k, cs, True, lanes)[0] == (slack >= 0), (needle, hay, k, cs, lanes, slack)
    assert marginal > 2500 and n_cases > 20000


def test_the_unicode_typo_prefilter_only_ever_deviates_on_marginal_inputs():
    # multi-chunk haystacks: with one scalar to spare (LCS + k >= n + 1) every width accepts, below the criterion every width rejects
    import pf_second_transcription as P2
    rng = np.random.default_rng(909)
    n_spare = n_reject = 0
    for _ in range(5000):
        asz = int(rng.integers(3, len(UNI_ALPHA) + 1))
        needle = "".join(UNI_ALPHA[int(x)] for x in rng.integers(0, asz, int(rng.integers(2, 10))))
        if needle.isascii():
            needle += "ж"
        cs, k = bool(rng.integers(0, 3) == 0), int(rng.integers(1, 5))
        chars = P2.case_needle_unicode(needle, cs)
        if k >= len(chars):
            continue
        hay = _uni_text(rng, asz, int(rng.integers(0, 200)), exact_len=True)
        slack = scalar_lcs(chars, hay) + k - len(chars)
        if slack == 0:
            continue
        for lanes in (16, 32, 64):
            assert O.prefilter(needle, hay, k, cs, True, lanes)[0] == (slack > 0), (needle, hay, k, cs, lanes, slack)
        n_spare += slack > 0
        n_reject += slack < 0
    assert n_spare > 1000 and n_reject > 300
