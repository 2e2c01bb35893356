"""Parity of the HIP path (through the C ABI) against the CPU oracle: bit-exact (index, score, exact) records in the
same order, on the reference's golden cases, on seeded random inputs, and on the benchmark's synthetic shapes."""
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
MT = json.load(open(os.path.join(G, "matcher.json")))

LANES = {64: (64, 64, 32), 32: (32, 32, 16), 16: (16, 16, 8)}


def both(needle, haystacks, pf=64, packed=None, **cfg):
    """Run HIP and oracle with the same config; returns (hip, oracle) record arrays."""
    kw = dict(cfg)
    scoring = kw.pop("scoring", None) or O.DEFAULT_SCORING
    om = O.Matcher(needle, lanes=LANES[pf], scoring=scoring, **kw)
    oi = om.info()
    fc = F.Config(max_typos=kw.get("max_typos", 0), casing=F.CaseMatching[kw.get("casing", "Smart")], unicode=F.UnicodeMatching[kw.get("unicode", "Smart")],
                  sort=F.SortStrategy[kw.get("sort", "ScoreThenIndexAsc")], scoring=F.Scoring(*scoring), pf_lanes=oi["pf_lanes"], sw_lanes=oi["sw_lanes"])
    fm = F.Matcher(needle, fc)
    if packed is not None:
        data, ends = packed
        got = fm.match_list(F.Corpus(packed=(data, ends)))
        want = om.match_packed(np.concatenate([data, np.zeros(64, np.uint8)]), ends)
    else:
        got = fm.match_list(haystacks)
        want = om.match_list(haystacks)
    return got, want, fm


def assert_same(got, want, ctx=""):
    if got.tolist() != want.tolist():
        n = min(len(got), len(want))
        bad = [i for i in range(n) if got[i].tolist() != want[i].tolist()][:5]
        raise AssertionError(f"{ctx}: len {len(got)} vs {len(want)}; first diffs {[(i, got[i].tolist(), want[i].tolist()) for i in bad]}")


def _expand(hs):
    if isinstance(hs, dict):
        n, patches = hs["haystacks_with"]
        out = ["nomatch-%d" % i for i in range(n)]
        for i, s in patches:
            out[i] = s
        return out
    return hs


@pytest.mark.parametrize("pf", [64, 32, 16])
@pytest.mark.parametrize("case", MT["cases"], ids=lambda c: c["name"])
def test_reference_golden_cases_through_hip(case, pf):
    hs = _expand(case["haystacks"])
    got, want, fm = both(case["needle"], hs, pf=pf, **case["config"])
    assert_same(got, want, case["name"])
    if "expect_indices" in case:
        assert got["index"].tolist() == case["expect_indices"], case["ref"]
    if "expect_scores" in case:
        assert got["score"].tolist() == case["expect_scores"], case["ref"]
    if "expect_exact_indices" in case:
        assert sorted(got["index"][got["exact"] != 0].tolist()) == case["expect_exact_indices"], case["ref"]
    for t in (1, 8):  # match_list_parallel == match_list (parallel.rs:104-130)
        assert fm.match_list_parallel(hs, t).tolist() == got.tolist()
    with pytest.raises(F.PanicError, match="threads must be positive"):
        fm.match_list_parallel(hs, 0)


@pytest.mark.parametrize("pf", [64, 32, 16])
@pytest.mark.parametrize("case", MT["same_result"], ids=lambda c: c["name"])
def test_configs_the_reference_asserts_equal_through_hip(case, pf):
    ga, wa, _ = both(case["needle"], case["haystacks"], pf=pf, **case["config_a"])
    gb, wb, _ = both(case["needle"], case["haystacks"], pf=pf, **case["config_b"])
    assert_same(ga, wa, case["name"])
    assert_same(gb, wb, case["name"])
    assert len(ga) == 1 and ga.tolist() == gb.tolist(), case["ref"]


def test_readme_smoke():
    got = F.Matcher("fBr").match_list(["fooBar", "foo_bar", "barfoo", "prelude", "println!"])
    assert got.tolist() == [(0, 53, 0, 0)]


ALPHA = b"abcABC_-/ 01xyzdeDEf.:"


def rand_case(rng, nmax=12, hmax=150, nh=400, alpha=ALPHA):
    asz = int(rng.integers(2, len(alpha)))
    nl = int(rng.integers(1, nmax + 1))
    needle = bytes(alpha[int(x)] for x in rng.integers(0, asz, nl))
    hs = []
    for _ in range(nh):
        hl = int(rng.integers(0, hmax)) if rng.random() > 0.25 else int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129]))
        h = bytearray(alpha[int(x)] for x in rng.integers(0, asz, hl))
        if hl > nl and rng.random() < 0.5:  # plant the needle (possibly with holes) so DP paths are exercised
            pos = np.sort(rng.choice(hl, nl, replace=False))
            for p, c in zip(pos, needle):
                if rng.random() < 0.9:
                    h[p] = c if rng.random() < 0.7 else bytes([c]).swapcase()[0]
        hs.append(bytes(h))
    return needle.decode(), hs


@pytest.mark.parametrize("pf", [64, 32, 16])
@pytest.mark.parametrize("max_typos", [0, 1, 2, 3, 5, None])
def test_random_ascii_parity(pf, max_typos):
    rng = np.random.default_rng(1000 + (max_typos or 77) * 7 + pf)
    for it in range(12):
        needle, hs = rand_case(rng)
        casing = ["Smart", "Ignore", "Respect"][it % 3]
        sort = ["ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"][it % 4]
        got, want, _ = both(needle, hs, pf=pf, max_typos=max_typos, casing=casing, sort=sort)
        assert_same(got, want, f"needle={needle!r} typos={max_typos} pf={pf} casing={casing}")


@pytest.mark.parametrize("pf", [64, 16])
def test_random_long_needles_u16_class(pf):
    rng = np.random.default_rng(4242 + pf)
    for it in range(8):
        needle, hs = rand_case(rng, nmax=40, hmax=300, nh=200)
        needle = (needle * 41)[: int(rng.integers(14, 41))]
        for k in (0, 2, None):
            got, want, fm = both(needle, hs, pf=pf, max_typos=k)
            assert not fm.info()["use_u8"]
            assert_same(got, want, f"needle={needle!r} typos={k}")


def test_random_scorings():
    rng = np.random.default_rng(99)
    for it in range(24):
        needle, hs = rand_case(rng, nmax=8, hmax=100, nh=200)
        sc = [int(rng.integers(0, 30)), int(rng.integers(0, 12)), int(rng.integers(0, 10)), int(rng.integers(0, 4)), int(rng.integers(0, 20)),
              int(rng.integers(0, 10)), int(rng.integers(0, 8)), int(rng.integers(0, 12)), int(rng.integers(0, 10))]
        if it % 6 == 0:
            sc = [int(x) * 40 for x in sc]  # push into the u16 class
        try:
            got, want, _ = both(needle, hs, max_typos=[0, 1, None][it % 3], scoring=sc)
        except RuntimeError as e:  # both sides must refuse identically
            with pytest.raises(F.PanicError):
                F.Matcher(needle, F.Config(scoring=F.Scoring(*sc)))
            continue
        assert_same(got, want, f"needle={needle!r} scoring={sc}")


def test_long_haystacks_multichunk_and_greedy():
    rng = np.random.default_rng(5)
    hs = []
    for L in [63, 64, 65, 127, 128, 129, 500, 1023, 1024, 1025, 1026, 1500, 3000]:
        for rep in range(4):
            h = bytearray(rng.choice(list(b"xyz_-/ABab"), L).tolist())
            pos = np.sort(rng.choice(L, 4, replace=False))
            for p, c in zip(pos, b"abcd"):
                h[p] = c
            if rep == 1:
                h[0:1] = b"a"
                h[-1:] = b"d"  # window spans everything -> greedy when > 1024
            hs.append(bytes(h))
    hs.append(b"a" + b"z" * 1100 + b"b")  # src/matcher/algo.rs:396-408
    for k in (0, 1, None):
        for pf in (64, 16):
            got, want, fm = both("abcd", hs, pf=pf, max_typos=k)
            assert_same(got, want, f"typos={k} pf={pf}")
    got, want, _ = both("abc", [b"a" + b"z" * 1100 + b"b"], max_typos=1)
    assert_same(got, want)
    assert len(got) == 1


UNI = ["é", "ن", "다", "😀", "न", " ", "_", "/", "a", "b", "c", "A", "B", "É", "0", "إ", "م"]


@pytest.mark.parametrize("pf", [64, 32, 16])
@pytest.mark.parametrize("max_typos", [0, 1, 2, 4, None])
def test_random_unicode_parity(pf, max_typos):
    rng = np.random.default_rng(31337 + (max_typos or 9) + pf)
    for it in range(10):
        asz = int(rng.integers(3, len(UNI)))
        needle = "".join(UNI[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 7))))
        if needle.isascii():
            needle += "é"
        hs = []
        for _ in range(250):
            n = int(rng.integers(0, 60)) if rng.random() > 0.2 else int(rng.choice([0, 1, 7, 8, 15, 16, 31, 32, 33]))
            chars = [UNI[int(x)] for x in rng.integers(0, asz, n)]
            if n > len(needle) and rng.random() < 0.5:
                pos = np.sort(rng.choice(n, len(needle), replace=False))
                for p, c in zip(pos, needle):
                    if rng.random() < 0.9:
                        chars[p] = c
            hs.append("".join(chars))
        casing = ["Smart", "Ignore", "Respect"][it % 3]
        got, want, _ = both(needle, hs, pf=pf, max_typos=max_typos, casing=casing)
        assert_same(got, want, f"needle={needle!r} typos={max_typos} pf={pf} casing={casing}")


def test_unicode_always_on_ascii_needle_and_ignore_on_unicode_needle():
    rng = np.random.default_rng(8)
    needle, hs = rand_case(rng, nmax=5, hmax=80, nh=300)
    hs = [h + "é다".encode() if i % 3 == 0 else h for i, h in enumerate(hs)]
    for k in (0, 1, None):
        got, want, _ = both(needle, hs, max_typos=k, unicode="Always")
        assert_same(got, want, f"Always typos={k}")
        got, want, _ = both("é" + needle, hs, max_typos=k, unicode="Ignore")
        assert_same(got, want, f"Ignore typos={k}")


@pytest.mark.parametrize("needle,max_typos", [(b"deadbe", 0), (b"deadbe", 2), (b"deadbe", None)])
def test_bench_shape_len32(needle, max_typos):
    n = 1_000_000 if max_typos is not None else 100_000
    rows, ends = synth.fixed_corpus(needle, n, 32)
    data = rows.numpy().reshape(-1)
    got, want, fm = both(needle.decode(), None, packed=(data, ends), max_typos=max_typos)
    assert_same(got, want, f"len32 typos={max_typos}")
    assert len(got) > 0


def test_bench_shape_ragged_8_128():
    data, ends = synth.ragged_corpus(b"deadbeef", 300_000)
    got, want, fm = both("deadbeef", None, packed=(data, ends), max_typos=0)
    assert_same(got, want, "ragged")
    assert fm.last_counters()["multi_chunk_scored"] > 0  # some windows are wider than one chunk


def test_bench_shape_utf8():
    data, ends = synth.utf8_corpus(100_000, 32)
    got, want, fm = both("إنما", None, packed=(data, ends), max_typos=0)
    assert_same(got, want, "utf8")
    assert len(got) > 1000


def test_subrange_and_index_offset_match_chunked_reference_calls():
    # what match_list_parallel's workers do: match_list_into(chunk, start as u32) (src/matcher/parallel.rs:55-63)
    rows, ends = synth.fixed_corpus(b"deadbe", 50_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    fm = F.Matcher("deadbe", F.Config(pf_lanes=64, sw_lanes=64))
    whole = fm.match_list_into(cp)
    parts = [fm.match_list_into(cp, first=s, count=min(2048, 50_000 - s), index_offset=s) for s in range(0, 50_000, 2048)]
    assert np.concatenate(parts).tolist() == whole.tolist()
    shifted = fm.match_list_into(cp, first=100, count=1000, index_offset=7)
    ref = whole[(whole["index"] >= 100) & (whole["index"] < 1100)].copy()
    ref["index"] = ref["index"] - 100 + 7
    assert shifted.tolist() == ref.tolist()
    with pytest.raises(F.PanicError, match="too many items in haystack"):
        fm.match_list_into(cp, first=0, count=10, index_offset=0xFFFFFFFF)


def test_empty_inputs():
    fm = F.Matcher("abc")
    assert len(fm.match_list([])) == 0
    assert fm.match_list(["", "", "abc"]).tolist() == O.Matcher("abc").match_list(["", "", "abc"]).tolist()
    got = F.Matcher("", F.Config(sort=F.SortStrategy.IndexDesc)).match_list(["x", "y", "z"])
    assert got["index"].tolist() == [2, 1, 0] and got["score"].tolist() == [0, 0, 0]
    got, want, _ = both("abc", ["", "a", ""], max_typos=None)
    assert_same(got, want)
    got, want, _ = both("abc", ["", "a", ""], max_typos=3)
    assert_same(got, want)

