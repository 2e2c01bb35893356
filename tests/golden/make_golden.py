#!/usr/bin/env python3
"""Writes tests/golden/*.json: known-answer vectors transcribed BY HAND from the reference's own
tests (the reference is Rust and cannot be executed in the build image, so these are the values its
test-suite asserts, not values produced by running it).  Each vector carries the reference
file:line of the assertion it comes from.  Re-run to regenerate the JSON after editing."""
import json, os

HERE = os.path.dirname(os.path.abspath(__file__))
CHAR = 16  # MATCH_SCORE 12 + MATCHING_CASE_BONUS 4 (src/smith_waterman/mod.rs:159)
PREFIX, DELIM, CAP, GOP, GEX, MATCH = 12, 4, 4, 5, 1, 12

# ---- score_haystack(needle, haystack, include_prefix=true), case-insensitive, default Scoring,
#      asserted on BackendScalar8 (LANES=8, u16): src/smith_waterman/mod.rs:163-166 ----
sw_ascii = [
    ("b", "abc", CHAR, "src/smith_waterman/mod.rs:209"),
    ("c", "abc", CHAR, "src/smith_waterman/mod.rs:210"),
    ("a", "abc", CHAR + PREFIX, "src/smith_waterman/mod.rs:215"),
    ("a", "aabc", CHAR + PREFIX, "src/smith_waterman/mod.rs:216"),
    ("a", "babc", CHAR, "src/smith_waterman/mod.rs:217"),
    ("a", "a", CHAR + PREFIX, "src/smith_waterman/mod.rs:222"),
    ("abc", "abc", 3 * CHAR + PREFIX, "src/smith_waterman/mod.rs:223"),
    ("-", "a--bc", CHAR, "src/smith_waterman/mod.rs:256"),
    ("b", "a-b", CHAR + DELIM, "src/smith_waterman/mod.rs:257"),
    ("a", "a-b-c", CHAR + PREFIX, "src/smith_waterman/mod.rs:258"),
    ("b", "a--b", CHAR + DELIM, "src/smith_waterman/mod.rs:259"),
    ("c", "a--bc", CHAR, "src/smith_waterman/mod.rs:260"),
    ("a", "-a--bc", CHAR + DELIM, "src/smith_waterman/mod.rs:261"),
    ("-", "a-bc", CHAR, "src/smith_waterman/mod.rs:266"),
    ("-", "a--bc", CHAR, "src/smith_waterman/mod.rs:267"),
    ("test", "Uteost", CHAR * 4 - GOP, "src/smith_waterman/mod.rs:273-276"),
    ("test", "Uteoost", CHAR * 4 - GOP - GEX, "src/smith_waterman/mod.rs:277-280"),
    ("test", "Utooooeoooosoooot", CHAR * 4 - GOP * 3 - GEX * 9, "src/smith_waterman/mod.rs:281-284"),
    ("test", "Utooooooeoooooosoooooot", CHAR * 4 - GOP * 3 - GEX * 15, "src/smith_waterman/mod.rs:285-288"),
    ("a", "A", MATCH + PREFIX, "src/smith_waterman/mod.rs:293"),
    ("A", "Aa", CHAR + PREFIX, "src/smith_waterman/mod.rs:294"),
    ("D", "forDist", CHAR + CAP, "src/smith_waterman/mod.rs:295"),
    ("D", "foRDist", CHAR, "src/smith_waterman/mod.rs:296"),
    ("D", "FOR_DIST", CHAR + DELIM, "src/smith_waterman/mod.rs:297"),
    ("foo", "Ufooo", CHAR * 3, "src/smith_waterman/mod.rs:422"),
    ("foo", "Ufo", CHAR * 2 - GOP, "src/smith_waterman/mod.rs:424-431"),
    ("foo", "Uf", CHAR - GOP - GEX, "src/smith_waterman/mod.rs:433-436"),
    ("foo", "U", 0, "src/smith_waterman/mod.rs:438-439"),
]
# long-haystack boundary incl. the greedy fallback at 1025 (src/smith_waterman/mod.rs:508-520)
sw_long = [("abc", L, 3 * CHAR, "src/smith_waterman/mod.rs:510-513") for L in (1023, 1024, 1025)]
# case-sensitive scorer (src/smith_waterman/mod.rs:351-362)
sw_case = [
    ("A", "A", True, CHAR + PREFIX, "src/smith_waterman/mod.rs:353-356"),
    ("A", "a", False, MATCH + PREFIX, "src/smith_waterman/mod.rs:358-361"),
]
# ordering assertions: score(a) > score(b)  (src/smith_waterman/mod.rs:268, 300-349)
sw_greater = [
    (("a_b", "a_bb"), ("a_b", "a__b"), "src/smith_waterman/mod.rs:268"),
    (("swap", "swap(test)"), ("swap", "iter_swap(test)"), "src/smith_waterman/mod.rs:302"),
    (("_", "_private_member"), ("_", "public_member"), "src/smith_waterman/mod.rs:303"),
    (("H", "HELLO"), ("H", "fooHello"), "src/smith_waterman/mod.rs:308"),
    (("foo", "fooo"), ("foo", "f_o_o_o"), "src/smith_waterman/mod.rs:313"),
    (("fo", "foo"), ("fo", "faOo"), "src/smith_waterman/mod.rs:318"),
    (("abc", "a111bc"), ("abc", "a1b1c"), "src/smith_waterman/mod.rs:341"),
    (("b", "b"), ("b", "a-b"), "src/smith_waterman/mod.rs:346"),
    (("b", "a-b"), ("b", "ab"), "src/smith_waterman/mod.rs:347"),
    (("B", "aB"), ("b", "aB"), "src/smith_waterman/mod.rs:348"),
]
# score_haystack_unicode (src/smith_waterman/mod.rs:227-252)
sw_unicode = [
    ("é", "é", CHAR + PREFIX, "src/smith_waterman/mod.rs:229"),
    ("😀", "😀", CHAR + PREFIX, "src/smith_waterman/mod.rs:230"),
    ("éx", "éx", 2 * CHAR + PREFIX, "src/smith_waterman/mod.rs:231"),
    ("ab", "aéb", 2 * CHAR + PREFIX - GOP, "src/smith_waterman/mod.rs:240-243"),
    ("ab", "aé😀b", 2 * CHAR + PREFIX - GOP - GEX, "src/smith_waterman/mod.rs:248-251"),
]
sw_unicode_equal = [(("éx", "ébx"), ("éx", "é😀x"), "src/smith_waterman/mod.rs:236-239")]
# fixed corpus that every backend width must score identically (src/smith_waterman/backend/tests/parity.rs:95-124)
sw_cross_width = [
    ("a", "abc"), ("abc", "abc"), ("foo", "fooBar"), ("foo", "012345foo"), ("foo", "01234567foo"),
    ("foo", "0123456789foo"), ("foo", "0123456789012345foo"), ("foo", "0123456789012345678901234567foo"),
    ("test", "Utooooeoooosoooot"), ("test", "Utooooooeoooooosoooooot"), ("foo", "Ufooo"), ("foo", "Ufo"),
    ("hw", "hello_world"), ("fBr", "fooBar"), ("D", "FOR_DIST"), ("needle", "____________needle____________"),
    ("abcdefghij", "abcdefghij"), ("abcdefghijklmnopqrst", "abcdefghijklmnopqrst"),
]
# match_greedy(needle, haystack, default, case_sensitive=false, include_prefix=true) (src/smith_waterman/greedy.rs:100-192)
greedy = [
    ("b", "abc", CHAR, "src/smith_waterman/greedy.rs:114"), ("c", "abc", CHAR, "src/smith_waterman/greedy.rs:115"),
    ("fbb", "barbazfoobarbaz", CHAR - GOP - GEX + CHAR - GOP - GEX + CHAR, "src/smith_waterman/greedy.rs:116-122"),
    ("a", "b", 0, "src/smith_waterman/greedy.rs:127"), ("ab", "ba", 0, "src/smith_waterman/greedy.rs:128"), ("abc", "ab", 0, "src/smith_waterman/greedy.rs:129"),
    ("a", "abc", CHAR + PREFIX, "src/smith_waterman/greedy.rs:134"), ("a", "aabc", CHAR + PREFIX, "src/smith_waterman/greedy.rs:135"), ("a", "babc", CHAR, "src/smith_waterman/greedy.rs:136"),
    ("-", "a--bc", CHAR, "src/smith_waterman/greedy.rs:141"), ("b", "a-b", CHAR + DELIM, "src/smith_waterman/greedy.rs:142"), ("a", "a-b-c", CHAR + PREFIX, "src/smith_waterman/greedy.rs:143"),
    ("b", "a--b", CHAR + DELIM, "src/smith_waterman/greedy.rs:144"), ("c", "a--bc", CHAR, "src/smith_waterman/greedy.rs:145"), ("a", "-a--bc", CHAR, "src/smith_waterman/greedy.rs:146"),
    ("-", "a-bc", CHAR, "src/smith_waterman/greedy.rs:151"), ("test", "Uterst", CHAR * 4 - GOP, "src/smith_waterman/greedy.rs:165-168"),
    ("test", "Uterrst", CHAR * 4 - GOP - GEX, "src/smith_waterman/greedy.rs:169-172"),
    ("a", "A", MATCH + PREFIX, "src/smith_waterman/greedy.rs:177"), ("A", "Aa", CHAR + PREFIX, "src/smith_waterman/greedy.rs:178"),
    ("d", "forDist", MATCH + CAP, "src/smith_waterman/greedy.rs:179-182"), ("D", "forDist", CHAR + CAP, "src/smith_waterman/greedy.rs:183"),
    ("D", "foRDist", CHAR, "src/smith_waterman/greedy.rs:184"), ("D", "FOR_DIST", CHAR + DELIM, "src/smith_waterman/greedy.rs:185"),
]
json.dump({
    "greedy": [dict(needle=n, haystack=h, score=s, ref=r) for n, h, s, r in greedy],
    "greedy_huge_gap": dict(needle="ab", x_count=70000, score=4, ref="src/smith_waterman/greedy.rs:157-161"),
    "default_scoring": [12, 6, 5, 1, 12, 4, 4, 8, 4],
    "sw_ascii": [dict(needle=n, haystack=h, score=s, ref=r) for n, h, s, r in sw_ascii],
    "sw_long": [dict(needle=n, haystack_len=L, score=s, ref=r) for n, L, s, r in sw_long],
    "sw_case": [dict(needle=n, haystack=h, case_sensitive=cs, score=s, ref=r) for n, h, cs, s, r in sw_case],
    "sw_greater": [dict(a=list(a), b=list(b), ref=r) for a, b, r in sw_greater],
    "sw_unicode": [dict(needle=n, haystack=h, score=s, ref=r) for n, h, s, r in sw_unicode],
    "sw_unicode_equal": [dict(a=list(a), b=list(b), ref=r) for a, b, r in sw_unicode_equal],
    "sw_cross_width": [dict(needle=n, haystack=h, ref="src/smith_waterman/backend/tests/parity.rs:95-124,184-190") for n, h in sw_cross_width],
}, open(os.path.join(HERE, "smith_waterman.json"), "w"), ensure_ascii=False, indent=1)

# ---- prefilter (src/prefilter/mod.rs:187-404) ----
U = "_"
pf_bool = [  # (needle, haystack, max_typos, case_sensitive, want)
    ("foo", "foo", 0, False, True), ("foo", "f_o_o", 0, False, True), ("foo", "FOO", 0, False, True),
    ("abc", "xaxbxcx", 0, False, True), ("fo", U * 15 + "fo", 0, False, True),
    ("foo", "f" + U * 15 + "o" + U * 15 + "o", 0, False, True), ("foo", "oof", 0, False, False),
    ("abc", "cba", 0, False, False), ("foo", "fo", 0, False, False),
    ("foo", "f" + U * 25 + "o" + U * 6, 0, False, False), ("a", "", 0, False, False),
    ("\0", "abc", 0, False, False), ("aa", "a", 0, False, False),
    # typo_matching_cases :212-248
    ("abc", "", 2, False, False), ("abc", "", 3, False, True), ("abc", "bc", 1, False, True),
    ("abc", "ac", 1, False, True), ("abc", "ab", 1, False, True), ("bar", "ba", 1, False, True),
    ("bar", "ar", 1, False, True), ("hello", "hll", 2, False, True), ("abcdef", "abdf", 2, False, True),
    ("TeSt", "ES", 2, False, True), ("abc", "c", 2, False, True), ("a\0b", "ab", 1, False, True),
    ("foo", "fo", 5, False, True), ("abc", "a" + U * 15 + "b", 1, False, True),
    ("test", "t" + U * 15 + "s" + U * 15 + "t", 1, False, True),
    ("d63NacaDJaaaa", "63aeeaaaeeaaaaaaaNacaDJaaAa", 1, False, True), ("bar", "rb", 1, False, False),
    ("abcdef", "fcda", 2, False, False), ("TeSt", "ES", 1, False, False), ("abc", "cba", 1, False, False),
    ("abc", "cba", 2, False, True), ("aaa", "aa", 0, False, False), ("aaa", "aa", 1, False, True),
    ("aba", "aa", 1, False, True), ("aaba", "aba", 1, False, True),
    # case_sensitive_matching_cases :250-270
    ("foo", "foo", 0, True, True), ("foo", "FOO", 0, True, False), ("FoO", "xxFoOxx", 0, True, True),
    ("abc", "xaxbxcx", 0, True, True), ("abc", "xAxBxCx", 0, True, False), ("TeSt", "eS", 2, True, True),
    ("TeSt", "ES", 2, True, False), ("Ab", "b", 1, True, True), ("Ab", "ab", 0, True, False), ("Ab", "ab", 1, True, True),
]
pf_window = [  # (needle, haystack, max_typos, case_sensitive, unicode, [matched,start,end], ref)
    ("foo", "xxfooxfoo", 0, False, False, [True, 2, 9], "src/prefilter/mod.rs:274"),
    ("abc", "xxaybzczz", 0, False, False, [True, 2, 7], "src/prefilter/mod.rs:275"),
    ("abcd", "xxaydz", 2, False, False, [True, 2, 5], "src/prefilter/mod.rs:276"),
    ("abc", "xyz", 3, False, False, [True, 0, 3], "src/prefilter/mod.rs:277"),
    ("إن", "xxإنyy", 0, False, True, [True, 2, 6], "src/prefilter/mod.rs:283"),
    ("니다", "xx니__다yy", 0, False, True, [True, 2, 10], "src/prefilter/mod.rs:284"),
    ("😀", "xx😀yy", 0, False, True, [True, 2, 6], "src/prefilter/mod.rs:285"),
    ("é", "٩É", 0, False, True, [True, 2, 4], "src/prefilter/mod.rs:322"),
    ("É", "é", 0, False, True, [True, 0, 2], "src/prefilter/mod.rs:401"),
    ("إن", "ن", 1, False, True, [True, 0, 2], "src/prefilter/mod.rs:352-355"),
    ("éन😀", "😀", 2, False, True, [True, 0, 4], "src/prefilter/mod.rs:361-364"),
    ("😀éनZ", "Z", 3, False, True, [True, 0, 1], "src/prefilter/mod.rs:370-373"),
]
pf_unicode_bool = [  # (needle, haystack, max_typos, case_sensitive, want, ref)
    ("إن", "ۥ؆", 0, False, False, "src/prefilter/mod.rs:305-308"),
    ("é", "٩É", 0, True, False, "src/prefilter/mod.rs:323"),
    ("éé", "٩É٩É٩É", 1, False, True, "src/prefilter/mod.rs:324"),
    ("إن", "ن", 0, False, False, "src/prefilter/mod.rs:356"),
    ("éन😀", "😀", 1, False, False, "src/prefilter/mod.rs:365"),
    ("😀éनZ", "Z", 2, False, False, "src/prefilter/mod.rs:374"),
    ("إن", "ۥ", 1, False, False, "src/prefilter/mod.rs:386"),
    ("إن", "؆", 1, False, False, "src/prefilter/mod.rs:387"),
    ("É", "é", 0, True, False, "src/prefilter/mod.rs:402"),
]
json.dump({
    "pf_bool": [dict(needle=n, haystack=h, max_typos=t, case_sensitive=cs, matched=w, ref="src/prefilter/mod.rs:187-270") for n, h, t, cs, w in pf_bool],
    "pf_window": [dict(needle=n, haystack=h, max_typos=t, case_sensitive=cs, unicode=u, window=w, ref=r) for n, h, t, cs, u, w, r in pf_window],
    "pf_unicode_bool": [dict(needle=n, haystack=h, max_typos=t, case_sensitive=cs, matched=w, ref=r) for n, h, t, cs, w, r in pf_unicode_bool],
    # sweeps: expected window computed from the rule the reference asserts
    "pf_unicode_prefix_sweep": dict(needle="إن", prefix_lens=[0, 1, 7, 14, 15, 16, 31, 32, 63, 64], ref="src/prefilter/mod.rs:326-336"),
    "pf_ascii_chunk_sweep": dict(prefix_lens=[0, 1, 7, 8, 15, 16, 31, 32, 63, 64],
                                 cases=[["abc", 0, True], ["ac", 0, True], ["abcd", 0, False], ["abcd", 1, True]], ref="src/prefilter/mod.rs:472-503"),
}, open(os.path.join(HERE, "prefilter.json"), "w"), ensure_ascii=False, indent=1)

# ---- end-to-end Matcher (src/matcher/mod.rs:532-654, src/matcher/algo.rs:344-456, tests/api_properties.rs) ----
D4 = ["deadbeef", "deadbf", "deadbeefg", "deadbe"]
def hs_with(n, items):
    # expanded by the test as: ["nomatch-%d" % i ...] with the listed overrides (tests/api_properties.rs haystacks_with)
    return {"haystacks_with": [n, [list(x) for x in items]]}
matcher = [
    dict(name="test_basic", needle="deadbe", haystacks=D4, config=dict(max_typos=None), expect_indices=[3, 0, 2, 1], ref="src/matcher/mod.rs:533-547"),
    dict(name="test_no_typos", needle="deadbe", haystacks=D4, config=dict(max_typos=0), expect_len=3, ref="src/matcher/mod.rs:550-557"),
    dict(name="test_exact_match", needle="deadbe", haystacks=D4, config=dict(), expect_exact_indices=[3], ref="src/matcher/mod.rs:560-572"),
    dict(name="test_exact_matches", needle="deadbe", haystacks=["deadbe", "deadbeef", "deadbe", "deadbf", "deadbe", "deadbeefg", "deadbe"], config=dict(),
         expect_exact_indices=[0, 2, 4, 6], ref="src/matcher/mod.rs:575-594"),
    dict(name="test_small_needle", needle="1", haystacks=["1"], config=dict(max_typos=2), expect_indices=[0], expect_exact_indices=[0], ref="src/matcher/mod.rs:597-603"),
    dict(name="case_smart_lower", needle="foo", haystacks=["foo", "FOO", "fOo", "xxfooxx"], config=dict(sort="IndexAsc"), expect_indices=[0, 1, 2, 3], ref="src/matcher/mod.rs:620-628"),
    dict(name="case_respect", needle="foo", haystacks=["foo", "FOO", "fOo", "xxfooxx"], config=dict(sort="IndexAsc", casing="Respect"), expect_indices=[0, 3], ref="src/matcher/mod.rs:630-638"),
    dict(name="case_smart_upper", needle="FoO", haystacks=["foo", "FOO", "FoO", "xxFoOxx"], config=dict(sort="IndexAsc", casing="Smart"), expect_indices=[2, 3], ref="src/matcher/mod.rs:646-654"),
    dict(name="zero_gap_capitalization", needle="BBBB", haystacks=["aBaBaBaB"],
         config=dict(scoring=[40, 0, 0, 0, 0, 40, 0, 0, 0]), expect_scores=[320], ref="src/matcher/algo.rs:424-440"),
    dict(name="unsorted_preserves_order", needle="foo", haystacks=["foo", "nomatch", "xfoo", "f_o_o", "bar"], config=dict(sort="IndexAsc"), expect_indices=[0, 2, 3], ref="src/matcher/algo.rs:443-456"),
    dict(name="empty_needle", needle="", haystacks=["foo", "bar"], config=dict(), expect_indices=[0, 1], expect_scores=[0, 0], ref="tests/api_properties.rs:420-428"),
    dict(name="exact_flag_tracks", needle="deadbe", haystacks=["deadbe", "deadbeef", "deadbe", "deadbf", "xxdeadbexx"], config=dict(),
         expect_exact_map={"0": True, "1": False, "2": True, "4": False}, ref="tests/api_properties.rs:437-449"),
    dict(name="unicode_zero_typo", needle="إن", haystacks=["xxإنyy", "إن", "ۥ؆", "nomatch", "x" * 65], config=dict(max_typos=0, sort="IndexAsc"),
         expect_indices=[0, 1], expect_exact_list=[False, True], ref="tests/api_properties.rs:452-476"),
    dict(name="parallel_tie_chunks", needle="abc", haystacks=hs_with(4097, [(2047, "abc"), (2048, "abc"), (4096, "abc")]), config=dict(),
         expect_indices=[2047, 2048, 4096], ref="tests/api_properties.rs:656-665"),
    dict(name="unicode_typo_scalar_count_1", needle="إن", haystacks=["ن", "😀", "x"], config=dict(max_typos=1, sort="IndexAsc"), expect_indices=[0], ref="tests/api_properties.rs:559-567"),
    dict(name="unicode_typo_scalar_count_2", needle="éन😀", haystacks=["ن", "😀", "x"], config=dict(max_typos=2, sort="IndexAsc"), expect_indices=[1], ref="tests/api_properties.rs:569-573"),
    dict(name="parallel_chunk_boundaries", needle="abc", haystacks=hs_with(4101, [(0, "abc"), (2047, "xabc"), (2048, "abxc"), (2049, "alpha/beta/abc"), (4095, "ABC"), (4096, "a_b_c"), (4100, "zabc")]),
         config=dict(), expect_len=7, ref="tests/api_properties.rs:627-654"),
    dict(name="custom_scoring_within_guard", needle="abc", haystacks=["abc", "a_b_c"], config=dict(scoring=[8, 6, 5, 1, 12, 4, 1, 8, 4]), expect_len=2, ref="tests/api_properties.rs:770-778"),
    dict(name="greedy_fallback_membership", needle="abc", haystacks=["a" + "z" * 1100 + "b"], config=dict(max_typos=1), expect_len=1, ref="src/matcher/algo.rs:396-408"),
    dict(name="readme_smoke_fBr", needle="fBr", haystacks=["fooBar", "foo_bar", "barfoo", "prelude", "println!"], config=dict(), expect_indices=[0],
         ref="README.md usage example; BASELINE.json configs[0] (score 53 is hand-derived in SURVEY.md, not reference-pinned)"),
]
# pairs of configurations the reference asserts give identical match lists
same = [
    dict(name="long_prefiltered_fallback_ascii", needle="ab", haystacks=["xa" + "_" * 1200 + "b"],
         config_a=dict(sort="IndexAsc", scoring=[12, 6, 0, 0, 12, 0, 4, 8, 0]), config_b=dict(sort="IndexAsc", scoring=[12, 6, 0, 0, 12, 0, 4, 8, 0], max_typos=None),
         ref="tests/api_properties.rs:515-531"),
    dict(name="long_prefiltered_fallback_unicode", needle="éb", haystacks=["xé" + "_" * 1200 + "b"],
         config_a=dict(sort="IndexAsc", scoring=[12, 6, 0, 0, 12, 0, 4, 8, 0]), config_b=dict(sort="IndexAsc", scoring=[12, 6, 0, 0, 12, 0, 4, 8, 0], max_typos=None),
         ref="tests/api_properties.rs:543-546"),
]
json.dump({"cases": matcher, "same_result": same,
           "panics": [dict(needle="f", scoring=[12, 6, 5, 1, 12, 60000, 40000, 8, 4], message_contains="needle too long", ref="src/matcher/algo.rs:372-380")],
           "max_needle_len_default": 10922, "max_needle_len_ref": "src/lib.rs:545-547",
           "score_fits_in_u8": [dict(needle_len=4, scoring=[12, 6, 5, 1, 12, 4, 4, 8, 4], fits=True, ref="src/smith_waterman/mod.rs:523"),
                                dict(needle_len=4, scoring=[12, 6, 5, 8, 12, 4, 4, 8, 4], fits=False, ref="src/smith_waterman/mod.rs:526-530")]},
          open(os.path.join(HERE, "matcher.json"), "w"), ensure_ascii=False, indent=1)
# ---- multi-pattern composition, fuzzy patterns (src/matcher/multi.rs tests; patterns as (needle, negated, max_typos or "inherit")) ----
multi = [
    dict(name="scores_sum", patterns=[["foo", False, "inherit"], ["foo", False, "inherit"]], haystacks=["foo", "xfoox", "bar"], config=dict(sort="IndexAsc"),
         expect_double_of_single="foo", ref="src/matcher/multi.rs:192-207"),
    dict(name="score_sorted", patterns=[["foo", False, "inherit"], ["bar", False, "inherit"]], haystacks=["xfoobarx", "foobar", "zzz"], config=dict(),
         expect_len=2, expect_first_index=1, expect_sorted=True, ref="src/matcher/multi.rs:230-237"),
    dict(name="max_typos_override_beats_config", patterns=[["helloz", False, 1]], haystacks=["hello", "world"], config=dict(max_typos=0, sort="IndexAsc"),
         expect_indices=[0], ref="src/matcher/multi.rs:338-349"),
    dict(name="no_override_is_strict", patterns=[["helloz", False, "inherit"]], haystacks=["hello", "world"], config=dict(max_typos=0, sort="IndexAsc"),
         expect_indices=[], ref="src/matcher/multi.rs:335-337"),
    dict(name="max_typos_override_per_pattern", patterns=[["foo", False, "inherit"], ["barz", False, 1]], haystacks=["foo bar", "fox bar"],
         config=dict(max_typos=0, sort="IndexAsc"), expect_indices=[0], ref="src/matcher/multi.rs:352-368"),
    dict(name="smart_case_per_pattern", patterns=[["Foo", False, "inherit"], ["bar", False, "inherit"]], haystacks=["Foo BAR", "foo bar"],
         config=dict(casing="Smart", sort="IndexAsc"), expect_indices=[0], ref="src/matcher/multi.rs:389-398"),
    dict(name="unicode_per_pattern", patterns=[["다나", False, "inherit"], ["foo", False, "inherit"]], haystacks=["다나 foo", "dana foo", "다나"],
         config=dict(sort="IndexAsc"), expect_indices=[0], ref="src/matcher/multi.rs:401-406"),
    dict(name="empty_patterns_match_everything", patterns=[], haystacks=["foo", "bar"], config=dict(), expect_len=2, ref="src/matcher/multi.rs:409-413"),
    dict(name="only_empty_needles_match_everything", patterns=[["", True, "inherit"], ["", False, "inherit"]], haystacks=["foo", "bar"], config=dict(), expect_len=2,
         ref="src/matcher/multi.rs:415-416; src/matcher/mod.rs:193-195"),
]
json.dump({"cases": multi}, open(os.path.join(HERE, "multi.json"), "w"), ensure_ascii=False, indent=1)
# ---- literal matching (src/literal/mod.rs tests): Substring scores with CaseMatching::Ignore unless stated ----
EXACT = 8
lit_scores = [  # (needle, haystack, casing, expected score or None, ref)
    ("bar", "foobar", "Ignore", 3 * CHAR, "src/literal/mod.rs:121"),
    ("bar", "foo_bar", "Ignore", 3 * CHAR + DELIM, "src/literal/mod.rs:123-126"),
    ("ab", "ab_ab", "Ignore", 2 * CHAR + PREFIX, "src/literal/mod.rs:132"),
    ("b", "abc", "Ignore", CHAR, "src/literal/mod.rs:212"), ("c", "abc", "Ignore", CHAR, "src/literal/mod.rs:213"),
    ("a", "abc", "Ignore", CHAR + PREFIX, "src/literal/mod.rs:218"), ("a", "aabc", "Ignore", CHAR + PREFIX, "src/literal/mod.rs:219"), ("a", "babc", "Ignore", CHAR, "src/literal/mod.rs:220"),
    ("a", "a", "Ignore", CHAR + PREFIX + EXACT, "src/literal/mod.rs:227-230"), ("abc", "abc", "Ignore", 3 * CHAR + PREFIX + EXACT, "src/literal/mod.rs:231-234"),
    ("-", "a--bc", "Ignore", CHAR, "src/literal/mod.rs:239"), ("b", "a-b", "Ignore", CHAR + DELIM, "src/literal/mod.rs:240"), ("a", "a-b-c", "Ignore", CHAR + PREFIX, "src/literal/mod.rs:241"),
    ("b", "a--b", "Ignore", CHAR + DELIM, "src/literal/mod.rs:242"), ("c", "a--bc", "Ignore", CHAR, "src/literal/mod.rs:243"), ("a", "-a--bc", "Ignore", CHAR + DELIM, "src/literal/mod.rs:244"),
    ("-", "a-bc", "Ignore", CHAR, "src/literal/mod.rs:249"),
    ("a", "Ab", "Ignore", MATCH + PREFIX, "src/literal/mod.rs:255"), ("A", "Aa", "Ignore", CHAR + PREFIX, "src/literal/mod.rs:256"),
    ("D", "forDist", "Ignore", CHAR + CAP, "src/literal/mod.rs:257"), ("D", "foRDist", "Ignore", CHAR, "src/literal/mod.rs:258"), ("D", "FOR_DIST", "Ignore", CHAR + DELIM, "src/literal/mod.rs:259"),
    ("A", "0A", "Respect", CHAR, "src/literal/mod.rs:284-287"), ("A", "0a", "Respect", None, "src/literal/mod.rs:288"), ("A", "0a", "Ignore", MATCH, "src/literal/mod.rs:289-292"),
    ("é", "é", "Ignore", CHAR + PREFIX + EXACT, "src/literal/mod.rs:301-304"), ("éx", "éx", "Ignore", 2 * CHAR + PREFIX + EXACT, "src/literal/mod.rs:306-309"), ("é", "xé", "Ignore", CHAR, "src/literal/mod.rs:311"),
    ("Ꭰ", "\u1b70", "Ignore", None, "src/literal/mod.rs:335-339"),
    ("ß", "SS", "Ignore", None, "src/literal/mod.rs:352"), ("ß", "ss", "Ignore", None, "src/literal/mod.rs:353"),
    ("é", "É", "Respect", None, "src/literal/mod.rs:323-327"), ("и", "И", "Respect", None, "src/literal/mod.rs:323-327"), ("α", "Α", "Respect", None, "src/literal/mod.rs:323-327"),
]
lit_matches = [  # (needle, haystack, casing): must match (score unspecified)
    ("é", "É", "Ignore", "src/literal/mod.rs:319-322"), ("и", "И", "Ignore", "src/literal/mod.rs:319-322"), ("α", "Α", "Ignore", "src/literal/mod.rs:319-322"),
    ("Ꭰ", "ꭰ", "Ignore", "src/literal/mod.rs:340-343"), ("ß", "ß", "Ignore", "src/literal/mod.rs:351"),
]
lit_greater = [  # substring score(a) > score(b), casing Ignore
    (("swap", "swap(test)"), ("swap", "iter_swap(test)"), "src/literal/mod.rs:264"), (("_", "_private_member"), ("_", "public_member"), "src/literal/mod.rs:265"),
    (("H", "HELLO"), ("H", "fooHello"), "src/literal/mod.rs:270"), (("b", "b"), ("b", "a-b"), "src/literal/mod.rs:275"), (("b", "a-b"), ("b", "ab"), "src/literal/mod.rs:276"),
    (("B", "aB"), ("b", "aB"), "src/literal/mod.rs:277"),
]
lit_lists = [  # (matching, needle, haystacks, config extras, expected indices, expect all exact?, ref)
    ("Exact", "foo", ["foo", "foobar", "xfoo", "FOO"], {}, [0, 3], True, "src/literal/mod.rs:55-61"),
    ("Prefix", "foo", ["foobar", "barfoo", "foo", "xfoobar"], {}, [0, 2], None, "src/literal/mod.rs:66-72"),
    ("Suffix", "foo", ["foobar", "barfoo", "foo", "xfoobar"], {}, [1, 2], None, "src/literal/mod.rs:73-79"),
    ("Substring", "bar", ["xxbarxx", "bar", "nope", "foo_bar"], {}, [0, 1, 3], None, "src/literal/mod.rs:84-91"),
    ("Prefix", "foo", ["foo", "FOO", "fOo"], {"casing": "Respect"}, [0], None, "src/literal/mod.rs:138-151"),
    ("Prefix", "foo", ["foo", "FOO", "fOo"], {}, [0, 1, 2], None, "src/literal/mod.rs:153-159"),
    ("Substring", "é다😀", ["é다😀", "xxé다😀yy", "é다", "plain"], {}, [0, 1], None, "src/literal/mod.rs:165-171"),
    ("Exact", "é다😀", ["é다😀", "xxé다😀yy", "é다", "plain"], {}, [0], True, "src/literal/mod.rs:172-174"),
    ("Substring", "abcd", ["abc"], {}, [], None, "src/literal/mod.rs:187"), ("Prefix", "abcd", ["abc"], {}, [], None, "src/literal/mod.rs:188"),
    ("Suffix", "abcd", ["abc"], {}, [], None, "src/literal/mod.rs:189"), ("Exact", "abcd", ["abc"], {}, [], None, "src/literal/mod.rs:190"),
] + [("Substring", "bar", ["x" * k + "bar"], {}, [0], None, "src/literal/mod.rs:196-201") for k in (0, 1, 7, 8, 15, 16, 31, 32, 63, 64, 65)]
lit_prefix_equals_fuzzy = [("foo", "foo"), ("foo", "foobar"), ("fooBar", "fooBarBaz"), ("a", "abc")]  # src/literal/mod.rs:96-110 (+ exact == fuzzy for foo/foo :112-114)
# cross-backend corpus: every backend must agree (src/literal/backend.rs:118-153) - used as extra oracle-vs-HIP cases
lit_corpus = [("x", "y" * 200), ("needle", "xxneedlexxneedle_needle"), ("é다😀", "xxé다😀yyé다😀"), ("다", "가나다라마"), ("z", "abcdefghijklmnopqrstuvwxyz"), ("FoO", "prefix_FoO_FOO_foo"),
              ("a", "a" * 35), ("ab", "ab" * 12), ("aa", "baaab"), ("aaaa", "xaaaaaaaax"), ("aaa", "aaaaa"), ("bar", "x" * 30 + "bar"), ("ba", "a" * 30 + "ba"),
              ("foobar", "foobatefoobarfoobar"), ("é", "xÉyéZÉ"), ("café", "un CAFÉ, deux cafés"), ("Ꭰ", "\u1b70Ꭰꭰ\u1b70"), ("иха", "МУХА_ИХА_иха"), ("αβ", "ΑΒβα_αβ")]
# Pattern::parse / parse_query (src/pattern.rs:307-382): atom -> (needle, matching or None, negated)
parse_atoms = [
    ("foo", "foo", None, False), ("^foo", "foo", "Prefix", False), ("foo$", "foo", "Suffix", False), ("'foo", "foo", "Substring", False), ("^foo$", "foo", "Exact", False),
    ("!foo", "foo", "Substring", True), ("!^foo", "foo", "Prefix", True), ("!foo$", "foo", "Suffix", True), ("!'foo", "foo", "Substring", True), ("!^foo$", "foo", "Exact", True),
    ("\\^foo", "^foo", None, False), ("foo\\$", "foo$", None, False), ("\\'foo", "'foo", None, False), ("\\!foo", "!foo", None, False), ("foo\\ bar", "foo bar", None, False),
    ("!\\^foo", "^foo", "Substring", True), ("!\\!foo", "!foo", "Substring", True),
    ("foo\\\\$", "foo\\\\", "Suffix", False), ("foo\\bar", "foo\\bar", None, False), ("foo\\", "foo\\", None, False), ("a\\\\\\ b", "a\\\\ b", None, False),
]
parse_queries = [  # query -> needles
    ("foo !^bar", ["foo", "bar"], "src/pattern.rs:347-351"), ("  foo \t bar  ", ["foo", "bar"], "src/pattern.rs:353-356"), ("foo\\ bar baz", ["foo bar", "baz"], "src/pattern.rs:361-364"),
    ("foo\\\\ bar", ["foo\\\\", "bar"], "src/pattern.rs:370-373"), ("", [], "src/pattern.rs:378"), ("   ", [], "src/pattern.rs:379"), ("! ^$ '", [], "src/pattern.rs:380"),
]
# multi-pattern queries with the literal negations (src/matcher/multi.rs:165-228, 409-417)
multi_queries = [
    ("foo !bar", ["foobar", "foo", "barfoo", "bar", "qux"], dict(sort="IndexAsc"), [1], "src/matcher/multi.rs:165-170"),
    ("foo !^bar", ["foo/bar", "bar/foo", "foo", "foobar"], dict(sort="IndexAsc"), [0, 2, 3], "src/matcher/multi.rs:178-182"),
    ("foo !bar$", ["foo/bar", "bar/foo", "foo", "foobar"], dict(sort="IndexAsc"), [1, 2], "src/matcher/multi.rs:185-189"),
    ("!foo", ["foo", "bar", "xfoox", "qux"], dict(sort="IndexAsc"), [1, 3], "src/matcher/multi.rs:213-219"),
    ("!foo !qux", ["foo", "bar", "xfoox", "qux"], dict(sort="IndexAsc"), [1], "src/matcher/multi.rs:222-223"),
    ("foo !foo", ["foo", "foobar"], dict(), [], "src/matcher/multi.rs:226-230"),
    ("! ^$", ["foo", "bar"], dict(sort="IndexAsc"), [0, 1], "src/matcher/multi.rs:415-416"),
    ("^foo", ["fooX", "xfoo"], dict(sort="IndexAsc", max_typos=None), [0], "src/matcher/multi.rs:308-316"),
]
json.dump(dict(scores=lit_scores, matches=lit_matches, greater=lit_greater, lists=lit_lists, prefix_equals_fuzzy=lit_prefix_equals_fuzzy, corpus=lit_corpus,
               parse_atoms=parse_atoms, parse_queries=parse_queries, multi_queries=multi_queries),
          open(os.path.join(HERE, "literal.json"), "w"), ensure_ascii=False, indent=1)
# ---- traceback / matched indices (src/smith_waterman/mod.rs tests; BackendScalar8 = 8 lanes x u16, default scoring, case-insensitive) ----
idx_ascii = [("aa", "aaa", [1, 0], "src/smith_waterman/mod.rs:323"), ("ab", "abab", [1, 0], "src/smith_waterman/mod.rs:324"), ("abc", "xabcabc", [3, 2, 1], "src/smith_waterman/mod.rs:325"),
             ("_", "abc", [], "src/smith_waterman/mod.rs:444"), ("a", "abc", [0], "src/smith_waterman/mod.rs:445"), ("b", "abc", [1], "src/smith_waterman/mod.rs:446"),
             ("c", "abc", [2], "src/smith_waterman/mod.rs:447"), ("ac", "________________abc", [18, 16], "src/smith_waterman/mod.rs:448"), ("foo", "Uf", [1], "src/smith_waterman/mod.rs:449")]
idx_ascii += [("abc", "x" * (L - 3) + "abc", [L - 1, L - 2, L - 3], "src/smith_waterman/mod.rs:510-520") for L in (1023, 1024, 1025)]
idx_unicode = [  # (needle, haystack, haystack_start_pos, expected)
    ("é", "é", 0, [1, 0], "src/smith_waterman/mod.rs:454"), ("😀", "😀", 0, [3, 2, 1, 0], "src/smith_waterman/mod.rs:455"), ("aé", "aé", 0, [2, 1, 0], "src/smith_waterman/mod.rs:456"),
    ("é", "é", 3, [4, 3], "src/smith_waterman/mod.rs:460-468"), ("éx", "é😀x", 3, [9, 4, 3], "src/smith_waterman/mod.rs:472-480"),
    ("ab", "aéb", 0, [3, 0], "src/smith_waterman/mod.rs:485"), ("ab", "aé😀b", 0, [7, 0], "src/smith_waterman/mod.rs:486"), ("éx", "é😀x", 0, [6, 1, 0], "src/smith_waterman/mod.rs:487"),
    ("éé", "ééé", 0, [3, 2, 1, 0], "src/smith_waterman/mod.rs:492"), ("😀x", "_______😀x", 0, [11, 10, 9, 8, 7], "src/smith_waterman/mod.rs:493-496"),
    ("😀.a", "..😀a", 0, [6, 1], "src/smith_waterman/mod.rs:503"), ("😀.é", "..😀é", 0, [7, 6, 1], "src/smith_waterman/mod.rs:504"),
    ("😀 a", "  😀a", 0, [6, 1], "src/smith_waterman/mod.rs:505"), ("😀é", "..😀é", 0, [7, 6, 5, 4, 3, 2], "src/smith_waterman/mod.rs:506"),
]
score_typos = [  # get_score_typos(needle, haystack, max_typos): score if an alignment path within the budget exists (src/smith_waterman/mod.rs:421-440)
    ("foo", "Ufooo", 0, 3 * CHAR), ("foo", "Ufo", 0, None), ("foo", "Ufo", 1, 2 * CHAR - GOP), ("foo", "Ufo", 2, 2 * CHAR - GOP), ("foo", "Uf", 1, None),
    ("foo", "Uf", 2, CHAR - GOP - GEX), ("foo", "U", 2, None), ("foo", "U", 3, 0), ("foo", "U", 4, 0),
]
# Matcher::match_list_indices known answers (API level: prefilter -> trim -> scorer -> traceback, byte offsets of the ORIGINAL haystack).
# expect = one [index, exact or None (not asserted), positions or None, "sorted" positions or None] per returned record.
FLAT = [12, 6, 0, 0, 12, 0, 4, 8, 0]  # Scoring { gap_open_penalty: 0, gap_extend_penalty: 0, capitalization_bonus: 0, delimiter_bonus: 0, ..default }
idx_matcher = [
    dict(needle="إن", haystacks=["xxإنyy", "إن", "\u06e5\u0606", "nomatch", "x" * 65], config=dict(max_typos=0, sort="IndexAsc"),
         expect=[[0, False, [5, 4, 3, 2]], [1, True, [3, 2, 1, 0]]], ref="tests/api_properties.rs:452-485"),
    dict(needle="éx", haystacks=["é😀x"], config=dict(max_typos=None, sort="IndexAsc"), expect=[[0, None, [6, 1, 0]]], ref="tests/api_properties.rs:488-503"),
    dict(needle="😀x", haystacks=["_______😀x"], config=dict(max_typos=None, sort="IndexAsc"), expect=[[0, None, [11, 10, 9, 8, 7]]], ref="tests/api_properties.rs:505-513"),
    dict(needle="ab", haystacks=["xa" + "_" * 1200 + "b"], config=dict(sort="IndexAsc", scoring=FLAT), expect=[[0, None, [1202, 1]]], ref="tests/api_properties.rs:516-540"),
    dict(needle="ab", haystacks=["xa" + "_" * 1200 + "b"], config=dict(sort="IndexAsc", scoring=FLAT, max_typos=None), expect=[[0, None, [1202, 1]]], ref="tests/api_properties.rs:516-540"),
    dict(needle="éb", haystacks=["xé" + "_" * 1200 + "b"], config=dict(sort="IndexAsc", scoring=FLAT), expect=[[0, None, [1203, 2, 1]]], ref="tests/api_properties.rs:542-556"),
    dict(needle="éb", haystacks=["xé" + "_" * 1200 + "b"], config=dict(sort="IndexAsc", scoring=FLAT, max_typos=None), expect=[[0, None, [1203, 2, 1]]], ref="tests/api_properties.rs:542-556"),
    dict(needle="é", haystacks=["xxé"], config=dict(unicode="Ignore", sort="IndexAsc"), expect=[[0, None, None, [2, 3]]], ref="src/matcher/mod.rs:604-616"),
    dict(needle="foo", haystacks=["foo", "FOO", "fOo", "xxfooxx"], config=dict(casing="Respect", sort="IndexAsc"), expect=[[0, None, None], [3, None, None]], ref="src/matcher/mod.rs:630-643"),
    dict(needle="", haystacks=["foo", "bar"], config=dict(), expect=[[0, False, []], [1, False, []]], ref="tests/api_properties.rs:429-433; src/matcher/mod.rs:745-748"),
]
# CompiledPatterns::Multi (match_one_indices_multi, src/matcher/multi.rs:56-82)
idx_multi = [
    dict(query="foo fo", haystacks=["foo"], config=dict(), expect=[[0, None, [2, 1, 0]]], ref="src/matcher/multi.rs:277-282"),
]
# multi_pattern_match_list_indices_matches_match_list (src/matcher/multi.rs:253-274): same (index, score, exact) as match_list, positions strictly descending
idx_multi_same = dict(haystacks=["foobar", "foo", "barfoo", "bar", "qux", "FooBar"], queries=["foo !bar", "foo bar", "!foo", "foo fo"], config=dict(sort="IndexAsc"),
                      ref="src/matcher/multi.rs:253-274")
json.dump(dict(ascii=idx_ascii, unicode=idx_unicode, score_typos=score_typos, matcher=idx_matcher, multi=idx_multi, multi_same=idx_multi_same),
          open(os.path.join(HERE, "indices.json"), "w"), ensure_ascii=False, indent=1)
print("golden written")
