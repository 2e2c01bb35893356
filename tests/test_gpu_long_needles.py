"""Needles beyond 64 bytes / 63 rows on the GPU (the reference takes needles up to its overflow guard, 3 639 rows
with the default scoring, src/lib.rs:506-527; class selection src/matcher/mod.rs:448-498).  Every fuzzy and literal entry point against
the oracle at the AVX-512 and the SSE / scalar lane pairs."""
import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

pytestmark = pytest.mark.gpu

LANES = {64: (64, 64, 32), 16: (16, 16, 8)}


def rand_text(rng, n, alpha=b"abcdef_/ABC-. 01"):
    return bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), n))


def haystacks_for(rng, needle, count):
    """exact copies, the needle with insertions / deletions / case changes / substitutions, random text of all lengths incl. empty,
    too short, multi-chunk and beyond the 1024-byte greedy threshold"""
    n = len(needle)
    hs = [needle, needle.upper(), needle.lower(), b"", b"x", needle[: n // 2], needle + needle, b"__" + needle + b"--", needle[1:], needle[:-1]]
    for _ in range(count):
        kind = rng.integers(0, 6)
        if kind == 0:
            hs.append(rand_text(rng, int(rng.integers(0, 3 * n + 40))))
            continue
        out = bytearray()
        for ch in needle:
            r = rng.random()
            if r < 0.02 * kind:
                continue  # deletion
            if r < 0.04 * kind:
                out += rand_text(rng, 1)  # substitution
            else:
                out.append(ch ^ 0x20 if rng.random() < 0.1 and chr(ch).isalpha() else ch)
            if rng.random() < 0.15:
                out += rand_text(rng, int(rng.integers(1, 4 if kind < 5 else 20)))
        hs.append(bytes(rand_text(rng, int(rng.integers(0, 5))) + out + rand_text(rng, int(rng.integers(0, 5)))))
    return hs


@pytest.mark.parametrize("pf", [64, 16])
def test_long_ascii_needles_match_the_oracle(pf):
    rng = np.random.default_rng(pf)
    for n in (65, 100, 200, 1000):
        needle = rand_text(rng, n, b"abcdefgh_/")
        hs = haystacks_for(rng, needle, 160 if n < 1000 else 60)
        cp = F.Corpus(hs)
        for typos in (0, 1, 2, 3, 7, None):
            for sort in ("ScoreThenIndexAsc", "IndexDesc"):
                want = O.Matcher(needle, lanes=LANES[pf], max_typos=typos, sort=sort).match_list(hs)
                got = F.Matcher(needle, F.Config(max_typos=typos, sort=F.SortStrategy[sort], pf_lanes=pf)).match_list(cp)
                assert got.tolist() == want.tolist(), (n, typos, sort, len(got), len(want))
            assert len(want) >= 5, (n, typos)  # the list does exercise the scorer


def test_long_needle_boundaries_63_64_65_rows():
    rng = np.random.default_rng(7)
    for n in (62, 63, 64, 65, 66):  # 63 rows / 64 bytes is where the by-value needle ends
        needle = rand_text(rng, n, b"abcdefgh")
        hs = haystacks_for(rng, needle, 120)
        for typos in (0, 1, None):
            want = O.Matcher(needle, max_typos=typos).match_list(hs)
            got = F.Matcher(needle, F.Config(max_typos=typos, pf_lanes=64)).match_list(hs)
            assert got.tolist() == want.tolist(), (n, typos)


@pytest.mark.parametrize("pf", [64, 16])
def test_long_unicode_needles_match_the_oracle(pf):
    rng = np.random.default_rng(100 + pf)
    alpha = list("éàüñабвгд中文字abc_ ")
    for nchars in (64, 70, 150):
        needle = "".join(alpha[int(x)] for x in rng.integers(0, len(alpha), nchars))
        hs = [needle, needle.upper(), "", needle[: nchars // 2], "__" + needle + "--"]
        for _ in range(100):
            out = []
            for ch in needle:
                r = rng.random()
                if r < 0.03:
                    continue
                out.append(ch if r > 0.06 else alpha[int(rng.integers(0, len(alpha)))])
                if rng.random() < 0.2:
                    out.append(alpha[int(rng.integers(0, len(alpha)))])
            hs.append("".join(out))
        cp = F.Corpus(hs)
        for typos in (0, 1, 2, 4, None):
            want = O.Matcher(needle, lanes=LANES[pf], max_typos=typos).match_list(hs)
            got = F.Matcher(needle, F.Config(max_typos=typos, pf_lanes=pf)).match_list(cp)
            assert got.tolist() == want.tolist(), (nchars, typos, len(got), len(want))
            assert len(want) >= 3


def test_long_needle_in_the_u8_class_and_other_scorings():
    # a scoring under which a 100-byte needle still fits the u8 class (64 score lanes on the AVX-512 pair), and a heavier one
    rng = np.random.default_rng(3)
    needle = rand_text(rng, 100, b"abcdef")
    hs = haystacks_for(rng, needle, 150)
    for sc in ([1, 1, 1, 0, 0, 0, 0, 0, 0], [2, 1, 2, 1, 3, 1, 1, 2, 1], [30, 9, 7, 2, 20, 9, 5, 11, 6]):
        for pf in (64, 16):
            for typos in (0, 2, None):
                fm = F.Matcher(needle, F.Config(max_typos=typos, scoring=F.Scoring(*sc), pf_lanes=pf))
                om = O.Matcher(needle, lanes=LANES[pf], max_typos=typos, scoring=sc)
                assert fm.info()["use_u8"] == om.info()["use_u8"]
                assert fm.match_list(hs).tolist() == om.match_list(hs).tolist(), (sc, pf, typos)
    assert F.Matcher(needle, F.Config(scoring=F.Scoring(1, 1, 1, 0, 0, 0, 0, 0, 0))).info()["use_u8"] is True


def test_long_needle_literal_modes_indices_multi_and_requery():
    rng = np.random.default_rng(5)
    needle = rand_text(rng, 90, b"abcdefgh_")
    hs = haystacks_for(rng, needle, 150) + [b"zz" + needle, needle + b"zz", b"q" + needle.upper() + b"q"]
    cp = F.Corpus(hs)
    for matching in ("Exact", "Prefix", "Suffix", "Substring"):
        want = O.Matcher(needle, matching=matching).match_list(hs)
        got = F.Matcher(needle, F.Config(matching=F.Matching[matching], pf_lanes=64)).match_list(cp)
        assert got.tolist() == want.tolist() and len(want) >= 1, matching
    # matched byte positions (traced scorer with the needle in device memory)
    for typos in (0, 1, None):
        got = [(m.index, m.score, m.exact, m.indices) for m in F.Matcher(needle, F.Config(max_typos=typos, pf_lanes=64)).match_list_indices(cp)]
        assert got == O.Matcher(needle, max_typos=typos).match_list_indices_ordered(hs), typos
    got = [(m.index, m.score, m.exact, m.indices) for m in F.Matcher(needle, F.Config(matching=F.Matching.Substring)).match_list_indices(cp)]
    assert got == O.Matcher(needle, matching="Substring").match_list_indices_ordered(hs)
    # one long pattern in a multi-pattern matcher, positive and negated
    q = [F.Pattern(needle), F.Pattern("ab"), F.Pattern("zz", negated=True)]
    oq = [O.P(needle), O.P("ab"), O.P("zz", negated=True)]
    assert F.MultiMatcher(q, F.Config(pf_lanes=64)).match_list(cp).tolist() == O.MultiMatcher(oq).match_list(hs).tolist()
    q = [F.Pattern("ab"), F.Pattern(needle, negated=True, max_typos=1)]
    oq = [O.P("ab"), O.P(needle, negated=True, max_typos=1)]
    assert F.MultiMatcher(q, F.Config(pf_lanes=64)).match_list(cp).tolist() == O.MultiMatcher(oq).match_list(hs).tolist()
    # Matcher::set_pattern: short -> long -> longer -> short on one resident list
    m = F.Matcher("abc", F.Config(pf_lanes=64))
    for nd in ("abc", needle, needle + needle[:30], needle[:64], needle[:65], "ab"):
        m.set_pattern(nd)
        assert m.match_list(cp).tolist() == O.Matcher(nd).match_list(hs).tolist(), len(nd)


def test_needle_at_the_guard_bound_and_long_haystacks():
    # 3 639 rows is the longest needle the reference's guard lets through with the default scoring; haystacks beyond 1024 bytes take
    # match_greedy (src/smith_waterman/greedy.rs) with the long needle
    rng = np.random.default_rng(9)
    needle = rand_text(rng, 3639, b"abcdefghij")
    hs = [needle, needle[:2000], b"x" + needle + b"y", needle[:1800] + b"__" + needle[1800:], rand_text(rng, 5000), b""]
    for typos in (0, 2, None):
        for pf in (64, 16):
            want = O.Matcher(needle, lanes=LANES[pf], max_typos=typos).match_list(hs)
            got = F.Matcher(needle, F.Config(max_typos=typos, pf_lanes=pf)).match_list(hs)
            assert got.tolist() == want.tolist(), (typos, pf)
    n2 = rand_text(rng, 300, b"abcdef")  # a 300-row needle whose windows stay below 1024 bytes: the DP with 300 rows x up to 32 chunks
    hs2 = haystacks_for(rng, n2, 80)
    for typos in (0, 5, None):
        assert F.Matcher(n2, F.Config(max_typos=typos, pf_lanes=64)).match_list(hs2).tolist() == O.Matcher(n2, max_typos=typos).match_list(hs2).tolist(), typos


def test_long_needle_windows_beyond_1024_bytes_are_handed_to_the_greedy_kernel():
    """A 100-byte needle whose windows reach beyond 1024 bytes (whole-haystack windows and typo windows over long haystacks): the
    thread-per-window scorer queues them for the wave-per-haystack kernel (match_greedy, src/smith_waterman/greedy.rs:7-91), the rest it
    scores itself - every record against the oracle, both score classes."""
    rng = np.random.default_rng(11)
    needle = rand_text(rng, 100, b"abcdefgh_/")
    hs = haystacks_for(rng, needle, 120)
    for L in (1000, 1024, 1025, 1500, 2100):
        for _ in range(3):
            body = bytearray(rand_text(rng, L))
            at = sorted(rng.choice(L, size=len(needle), replace=False).tolist())
            for q, ch in zip(at, needle):
                body[q] = ch
            hs.append(bytes(body))
        hs.append(rand_text(rng, L))
    cp = F.Corpus(hs)
    for sc in (None, [1, 1, 1, 0, 0, 0, 0, 0, 0]):
        for typos in (None, 0, 3):
            kw = dict(scoring=sc) if sc else {}
            want = O.Matcher(needle, max_typos=typos, **kw).match_list(hs)
            got = F.Matcher(needle, F.Config(max_typos=typos, pf_lanes=64, **({"scoring": F.Scoring(*sc)} if sc else {}))).match_list(cp)
            assert got.tolist() == want.tolist(), (sc, typos, len(got), len(want))
            assert len(want) >= 20
