"""Multi-pattern composition through the C ABI (fzb_multi_*) against the oracle's restatement of src/matcher/multi.rs:
the reference's known answers, seeded random pattern sets shaped like the reference's generator
(tests/api_properties.rs:250-311, fuzzy patterns), and a 1 M-haystack list."""
import json
import os
import sys

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth  # noqa: E402
from test_oracle_multi import pats as oracle_pats, random_case  # noqa: E402

pytestmark = pytest.mark.gpu
MU = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "multi.json")))
LANES = {64: (64, 64, 32), 16: (16, 16, 8)}


def hip_patterns(opats):
    return [F.Pattern(p["needle"], negated=p["negated"], max_typos=None if p["max_typos"] == O.INHERIT else p["max_typos"],
                      casing=None if p["casing"] is None else F.CaseMatching[p["casing"]]) for p in opats]


def hip_config(pf, **cfg):
    # the sub-matchers' (pf, sw) pairs are per score class; 0 for sw lets the library take the pair's width for each pattern's class
    return F.Config(max_typos=cfg.get("max_typos", 0), casing=F.CaseMatching[cfg.get("casing", "Smart")], sort=F.SortStrategy[cfg.get("sort", "ScoreThenIndexAsc")],
                    pf_lanes=pf, sw_lanes=0)


def both(opats, hs, pf=64, **cfg):
    want = O.MultiMatcher(opats, lanes=LANES[pf], **cfg).match_list(hs)
    got = F.MultiMatcher(hip_patterns(opats), hip_config(pf, **cfg)).match_list(hs)
    return got, want


@pytest.mark.parametrize("pf", [64, 16])
@pytest.mark.parametrize("case", MU["cases"], ids=lambda c: c["name"])
def test_reference_known_answers_through_hip(case, pf):
    got, want = both(oracle_pats(case), case["haystacks"], pf=pf, **case["config"])
    assert got.tolist() == want.tolist(), case["ref"]
    if "expect_indices" in case:
        assert got["index"].tolist() == case["expect_indices"], case["ref"]
    if "expect_len" in case:
        assert len(got) == case["expect_len"], case["ref"]


@pytest.mark.parametrize("pf", [64, 16])
def test_random_pattern_sets(pf):
    rng = np.random.default_rng(777 + pf)
    nonempty = 0
    for it in range(150):
        opats, hs, cfg = random_case(rng)
        for sort in ("IndexAsc", "ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexDesc")[: 1 + it % 4]:
            got, want = both(opats, hs, pf=pf, sort=sort, **cfg)
            assert got.tolist() == want.tolist(), (opats, hs, cfg, sort)
        nonempty += len(got) > 0
    assert nonempty > 20


def test_one_million_haystacks_three_patterns():
    rows, ends = synth.fixed_corpus(b"deadbe", 1_000_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    odata = np.concatenate([data, np.zeros(64, np.uint8)])
    for opats, cfg in (
        ([O.P("dead"), O.P("be"), O.P("x", negated=True)], dict(max_typos=0)),
        ([O.P("q", negated=True), O.P("deadbe", max_typos=2)], dict(max_typos=0)),
        ([O.P("de"), O.P("ad", max_typos=1), O.P("be")], dict(max_typos=None)),
        ([O.P("z", negated=True), O.P("Q", negated=True)], dict(max_typos=0, sort="IndexDesc")),
    ):
        want = O.MultiMatcher(opats, **cfg).match_packed(odata, ends)
        got = F.MultiMatcher(hip_patterns(opats), hip_config(64, **cfg)).match_list(cp)
        assert len(got) == len(want) and len(got) > 0
        assert np.array_equal(got["index"], want["index"]) and np.array_equal(got["score"], want["score"]) and np.array_equal(got["exact"], want["exact"]), opats


def test_device_entry_point_subrange_and_offset():
    import torch
    rows, ends = synth.fixed_corpus(b"deadbe", 100_000, 32)
    data = rows.numpy().reshape(-1)
    cp = F.Corpus(packed=(data, ends))
    mm = F.MultiMatcher([F.Pattern("dead"), F.Pattern("x", negated=True)], F.Config(sort=F.SortStrategy.IndexAsc, pf_lanes=64))
    whole = mm.match_list(cp)
    dev = torch.device("cuda", 0)
    out = torch.zeros(100_000 * 8, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    mm.match_list_device(cp, out.data_ptr(), 100_000, cnt.data_ptr(), first=2048, count=50_000, index_offset=7)
    torch.cuda.synchronize()
    n = int(cnt[0].item())
    got = out[: n * 8].cpu().numpy().view(F.MATCH_DTYPE)
    ref = whole[(whole["index"] >= 2048) & (whole["index"] < 52048)].copy()
    ref["index"] = ref["index"] - 2048 + 7
    assert got.tolist() == ref.tolist()
