"""Every environment switch of frizbee_amd/csrc/knobs.h that forces ANOTHER SHIPPING FORM of a stage (the form that serves the needles,
scorings or corpora outside the fast form's preconditions) produces the oracle's records on the GPU, on ordinary inputs.
One process: fzb_debug_reload_knobs() re-reads the environment; matchers are created after it (some decisions are taken at creation)."""
import os
import random

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _ragged(rng, n, max_len, needle, alphabet):
    out = []
    for _ in range(n):
        L = rng.randint(0, max_len)
        s = [rng.choice(alphabet) for _ in range(L)]
        if rng.random() < 0.35 and L >= len(needle):
            for q, c in zip(sorted(rng.sample(range(L), len(needle))), needle):
                s[q] = c if rng.random() < 0.8 else c.swapcase()
        out.append("".join(s))
    return out


@pytest.fixture(scope="module")
def lists():
    rng = random.Random(2024)
    short = _ragged(rng, 30000, 32, "deadbe", "deabfxyz_-/ 01DEAB")
    short = [s.ljust(32, "q")[:32] for s in short]                       # a uniform 32-byte list: the short-haystack kernels
    ragged = _ragged(rng, 40000, 128, "deadbeef", "deabfxyz_-/ 01DEAB")  # view, classes, multi-chunk tail classes
    uni = _ragged(rng, 20000, 24, "éa", list("abéÉñ_ -/xyzü"))
    wide = _ragged(rng, 30000, 230, "deadbeef", "deabfxyz_-/ 01DEAB")         # haystacks beyond 128 bytes: the view's 16-vector groups
    uniwide = _ragged(rng, 8000, 150, "éa", list("abéÉñ_ -/xyzü")) + ["é" + "x" * 1100 + "a", "ñ" * 600 + "éa", "é" + "xü" * 150 + "a", "_é" + "y" * 700 + "a_", "ü" * 200 + "éa" + "x" * 300]  # unicode windows beyond a chunk / four chunks / 1024 bytes
    return {"short": (short, F.Corpus(short)), "ragged": (ragged, F.Corpus(ragged)), "uni": (uni, F.Corpus(uni)), "wide": (wide, F.Corpus(wide)), "uniwide": (uniwide, F.Corpus(uniwide))}


def _same(lists, which, needle, **cfg):
    hs, cp = lists[which]
    fc = F.Config(max_typos=cfg.get("max_typos", 0), pf_lanes=64, sw_lanes=64)
    got = F.Matcher(needle, fc).match_list(cp)
    want = O.Matcher(needle, lanes=(64, 64, 32), **cfg).match_list(hs)
    assert len(want) > 0 and got.tolist() == want.tolist(), (which, needle, cfg, {k: v for k, v in os.environ.items() if k.startswith("FZB_")})


CASES = [  # (environment, [(list, needle, oracle config)])
    ({}, [("wide", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=None)), ("uniwide", "éa", dict(max_typos=1))]),  # as shipped: the 16-vector view kernel, wide unicode windows
    ({"FZB_NO_LCS_DFA": "1"}, [("short", "deadbe", dict(max_typos=1)), ("ragged", "deadbeef", dict(max_typos=2)), ("uni", "éa", dict(max_typos=1))]),           # the bit-vector LCS filter (needles beyond 226 automaton states)
    ({"FZB_TYPO_EXACT_WINDOW": "1"}, [("short", "deadbe", dict(max_typos=2)), ("ragged", "deadbeef", dict(max_typos=1)), ("uni", "éa", dict(max_typos=1))]),  # every typo survivor through the lane-exact window kernel
    ({"FZB_NO_DP_CFU": "1"}, [("uni", "éa", dict()), ("uniwide", "éa", dict(max_typos=None))]),                              # the unicode scorers' first form
    ({"FZB_NO_DP_CLASSES": "1"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict())]),                            # k2b_dp + the queued multi-chunk scorer (dp_cfm.h form)
    ({"FZB_NO_DP_CFM": "1"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1))]),                    # classes + the queued multi-chunk scorer (dp_body.h) on the second stream
    ({"FZB_NO_DP_CLASSES": "1", "FZB_NO_DP_CFM": "1"}, [("ragged", "deadbeef", dict())]),                                    # both first forms, one stream
    ({"FZB_UNICODE_MULTI": "1"}, [("uniwide", "éa", dict(max_typos=None)), ("uniwide", "éa", dict()), ("uniwide", "éa", dict(max_typos=1))]),   # wide unicode windows: thread per haystack (hands its stragglers on)
    ({"FZB_UNICODE_MULTI": "0"}, [("uniwide", "éa", dict(max_typos=None)), ("uniwide", "éa", dict())]),                      # ... wave per haystack
    ({"FZB_PARK_LDS_KB": "0"}, [("ragged", "deadbeef", dict()), ("ragged", "deadbeef", dict(max_typos=1))]),                 # parked rows in the global slab
    ({"FZB_PARK_LDS_KB": "0", "FZB_NO_DP_CFM": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_NO_CDFA": "1"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1))]),                       # the burst filter over the byte automaton
    ({"FZB_COOP_BELOW": "100000000"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1)), ("ragged", "DeadBeef", dict())]),  # multi-chunk windows: four lanes per window (the default below 49 152 queued)
    ({"FZB_COOP_BELOW": "0"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict())]),                                # ... never
    ({"FZB_DEBUG_SYNC": "1"}, [("short", "deadbe", dict()), ("ragged", "deadbeef", dict()), ("uniwide", "éa", dict(max_typos=None))]),
    ({"FZB_NO_FUSED_CLASSIFY": "1"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict()), ("ragged", "deadbeef", dict(max_typos=None))]),  # k_compact1 + k2w_classify as two launches
    ({"FZB_NO_FUSED_CLASSIFY": "1", "FZB_COOP_BELOW": "0"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_SPIN_WAIT_US": "0"}, [("short", "deadbe", dict()), ("ragged", "deadbeef", dict())]),   # synchronous entry points block at once instead of polling first
]


@pytest.mark.parametrize("env,work", CASES, ids=[",".join(f"{k}={v}" for k, v in e.items()) for e, _ in CASES])
def test_alternative_paths_give_the_oracles_records(lists, env, work):
    saved = {k: os.environ.get(k) for k in env}
    try:
        os.environ.update(env)
        F.lib().fzb_debug_reload_knobs()
        for which, needle, cfg in work:
            _same(lists, which, needle, **cfg)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        F.lib().fzb_debug_reload_knobs()


def test_filter_view_switch_is_honoured_at_upload_and_at_launch():
    rng = random.Random(7)
    hs = _ragged(rng, 20000, 100, "deadbeef", "deabfxyz_-/ 01")
    want = O.Matcher("deadbeef", lanes=(64, 64, 32)).match_list(hs)
    try:
        os.environ["FZB_FILTER_VIEW"] = "0"
        F.lib().fzb_debug_reload_knobs()
        cp = F.Corpus(hs)                      # uploaded without a view: the class-composite filter over the canonical layout
        assert cp.build_view() is False
        assert F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64)).match_list(cp).tolist() == want.tolist()
    finally:
        os.environ.pop("FZB_FILTER_VIEW", None)
        F.lib().fzb_debug_reload_knobs()
    assert cp.build_view() is True             # ... and with it
    assert F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64)).match_list(cp).tolist() == want.tolist()
