"""The reference's generated public-API property test (tests/api_properties.rs:72-166) through the HIP path: for cases drawn with
the reference's own generator (needle / haystack list / max_typos / casing / matching mode / sort), `match_list`,
`match_list_parallel` and `match_list_indices` equal the oracle's record for record, and the reference's indices contract holds."""
import pytest

import frizbee_amd as F
import oracle_lib as O
from ref_generators import api_cases, assert_indices_contract

pytestmark = pytest.mark.gpu
LANES = {64: (64, 64, 32), 32: (32, 32, 16), 16: (16, 16, 8)}


@pytest.mark.parametrize("pf", [64, 32, 16])
def test_generated_public_api_properties_through_hip(pf):
    for it, (needle, haystacks, cfg) in enumerate(api_cases(400, 1000 + pf)):
        om = O.Matcher(needle, lanes=LANES[pf], **cfg)
        fc = F.Config(max_typos=cfg["max_typos"], casing=F.CaseMatching[cfg["casing"]], matching=F.Matching[cfg["matching"]], sort=F.SortStrategy[cfg["sort"]], pf_lanes=pf)
        fm = F.Matcher(needle, fc)
        corpus = F.Corpus(haystacks)
        want = om.match_list(haystacks)
        got = fm.match_list(corpus)
        assert got.tolist() == want.tolist(), (it, needle, cfg)
        assert fm.match_list(corpus).tolist() == want.tolist(), (it, needle, cfg)  # reusable == one-shot
        for threads in (1, 3):
            assert fm.match_list_parallel(corpus, threads).tolist() == want.tolist(), (it, needle, cfg, threads)
        ix = [(m.index, m.score, m.exact, m.indices) for m in fm.match_list_indices(corpus)]
        assert ix == om.match_list_indices_ordered(haystacks), (it, needle, cfg)
        assert_indices_contract(needle, haystacks, cfg, got, ix)


def test_reuse_handles_state_changes():
    # src/matcher/mod.rs:787-850: one matcher taken through set_pattern / set_config sequences equals a freshly built one every time
    def fresh(needle, hs, **cfg):
        return O.Matcher(needle, **cfg).match_list(hs).tolist()

    def conf(**cfg):
        return F.Config(max_typos=cfg.get("max_typos", 0), casing=F.CaseMatching[cfg.get("casing", "Smart")], sort=F.SortStrategy[cfg.get("sort", "ScoreThenIndexAsc")], pf_lanes=64)

    long_needle = "abcdefghijklmnopqrst"
    first = ["xxabcdefghijklmnopqrstxx", "abcdefghijklmnopqrst", "no-match"]
    c1 = dict(max_typos=None, sort="IndexAsc")
    m = F.Matcher(long_needle, conf(**c1))
    assert m.info()["use_u8"] is False  # u16_path_selected_for_long_needle, src/matcher/mod.rs:770-785
    assert m.match_list(first).tolist() == fresh(long_needle, first, **c1)
    second = ["fooBar", "foo_bar", "fbr", "bar"]
    c2 = dict(casing="Smart", sort="IndexAsc")
    m.set_pattern("fB")
    m.set_config(conf(**c2))
    assert m.info()["use_u8"] is True  # u8_path_selected_for_short_needle, src/matcher/mod.rs:751-768
    assert m.match_list(second).tolist() == fresh("fB", second, **c2)
    uni = ["é다😀", "xxé__다__😀yy", "é다", "plain ascii"]
    c3 = dict(max_typos=0, sort="IndexAsc")
    m.set_pattern("é다😀")
    m.set_config(conf(**c3))
    assert m.match_list(uni).tolist() == fresh("é다😀", uni, **c3)
    c4 = dict(casing="Ignore", max_typos=1)
    m.set_pattern("fB")
    m.set_config(conf(**c4))
    assert m.match_list(first).tolist() == fresh("fB", first, **c4)


@pytest.mark.parametrize("pf", [64, 16])
def test_generated_multi_pattern_properties_through_hip(pf):
    # generated_multi_pattern_properties, tests/api_properties.rs:311-416, with the reference's MultiPatternCase generator
    from ref_generators import multi_cases
    for it, (patterns, haystacks, cfg) in enumerate(multi_cases(250, 2000 + pf)):
        opats = [O.P(p["needle"], negated=p["negated"], matching=p["matching"]) for p in patterns]
        fpats = [F.Pattern(p["needle"], negated=p["negated"], matching=None if p["matching"] is None else F.Matching[p["matching"]]) for p in patterns]
        corpus = F.Corpus(haystacks)
        for sort in ("IndexAsc", "ScoreThenIndexAsc"):
            om = O.MultiMatcher(opats, lanes=LANES[pf], sort=sort, **cfg)
            fm = F.MultiMatcher(fpats, F.Config(max_typos=cfg["max_typos"], casing=F.CaseMatching[cfg["casing"]], matching=F.Matching[cfg["matching"]], sort=F.SortStrategy[sort], pf_lanes=pf))
            got = fm.match_list(corpus)
            assert got.tolist() == om.match_list(haystacks).tolist(), (it, patterns, cfg, sort)
            if sort == "IndexAsc":
                assert got.tolist() == om.reference_composition(haystacks).tolist(), (it, patterns, cfg)
            ix = [(m.index, m.score, m.exact, m.indices) for m in fm.match_list_indices(corpus)]
            assert ix == om.match_list_indices_ordered(haystacks), (it, patterns, cfg, sort)


@pytest.mark.parametrize("pf", [64, 32, 16])
def test_follows_the_reference_where_its_prefilter_deviates_from_the_lcs_criterion(pf):
    # tests/test_oracle_reference_properties.py LCS_DEVIATIONS_1_TYPO: the reference's two-path scan rejects these at 32 lanes (LCS accepts)
    from test_oracle_reference_properties import LCS_DEVIATIONS_1_TYPO
    hs = [h for _, h in LCS_DEVIATIONS_1_TYPO] + ["filler", ""]
    for needle, hay in LCS_DEVIATIONS_1_TYPO:
        want = O.Matcher(needle, lanes=LANES[pf], max_typos=1, sort="IndexAsc", casing="Ignore").match_list(hs)
        got = F.Matcher(needle, F.Config(max_typos=1, sort=F.SortStrategy.IndexAsc, casing=F.CaseMatching.Ignore, pf_lanes=pf)).match_list(hs)
        assert got.tolist() == want.tolist(), (needle, pf)
        assert (hs.index(hay) in want["index"].tolist()) == (pf != 32), (needle, pf)


# Haystacks of at most 32 bytes on which the reference's chunked typo prefilter REJECTS at 16 lanes although the LCS criterion (and the
# 64-lane backend) accepts - found by random search with the oracle (about 1 in 10^4 of such inputs).  A corpus this short takes the
# typo fast path (LCS filter with the "nothing to spare" bit -> decide pass over the marginal survivors -> k2b_dp_short with the
# lane-free window): the rejected survivors must disappear from the list and the records behind them move up.
SHORT_LCS_DEVIATIONS = [
    ("_C _B  AcA", "a 110C_Bc10_c1C 0c1_bB1Ab/c/ A", 2),
    ("-BBaca_Ab", "--_-c/ / 0cCa/_/c-B/bC_baB1-_ -", 2),
    ("_cAa0AA_0", "0Aca1Acc Aa_Cc B1AAa1a_a0A1_1", 1),
    ("AAC_1_ b_", "0 _cAB1c0C-BbBC-0-a-1Ac1_ 0c01Ba", 2),
    ("/Abc B1BbB", "b1 //bB01-b_A- _ C_c 01BB1 b1/", 2),
    ("c1_CAbbAB", "bAa_b BB-Acc0C--1cbCcbb0aBA-_0_", 2),
]


@pytest.mark.parametrize("pf", [64, 16])
def test_typo_fast_path_drops_the_survivors_the_exact_prefilter_rejects(pf):
    import random

    rng = random.Random(5)
    for needle, hay, k in SHORT_LCS_DEVIATIONS:
        assert O.prefilter(needle, hay, k, False, False, 16)[0] is False and O.prefilter(needle, hay, k, False, False, 64)[0] is True
        # the deviating haystack several times between accepted and rejected ones, over more than one 1024-haystack tile
        filler = ["".join(rng.choice("abcABC_-/ 01") for _ in range(rng.randint(0, 32))) for _ in range(3000)]
        hs = filler[:700] + [hay] + filler[700:1500] + [hay, hay, needle, needle[:-1]] + filler[1500:] + [hay]
        want = O.Matcher(needle, lanes=(pf, 64, 32), max_typos=k, sort="IndexAsc", casing="Ignore").match_list(hs)
        m = F.Matcher(needle, F.Config(max_typos=k, sort=F.SortStrategy.IndexAsc, casing=F.CaseMatching.Ignore, pf_lanes=pf, sw_lanes=64))
        got = m.match_list(hs)
        assert got.tolist() == want.tolist(), (needle, pf, len(got), len(want))
        assert (700 in want["index"].tolist()) == (pf != 16), (needle, pf)
        c = m.last_counters()
        assert c["kept_by_exact_prefilter"] == len(want) and c["filter_survivors"] >= len(want)
