"""Every environment switch of frizbee_amd/csrc/knobs.h that selects ANOTHER FORM of a stage (the older / literal form of the same
arithmetic, kept for comparison) produces the oracle's records on the GPU; the tuning switches (grid shapes) leave the records unchanged.
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
    ({"FZB_NO_LCS_DFA": "1"}, [("short", "deadbe", dict(max_typos=1)), ("ragged", "deadbeef", dict(max_typos=2))]),
    ({"FZB_TYPO_EXACT_WINDOW": "1"}, [("short", "deadbe", dict(max_typos=2)), ("ragged", "deadbeef", dict(max_typos=1))]),
    ({"FZB_NO_DP_CFU": "1"}, [("uni", "éa", dict())]),
    ({"FZB_K2U_WAVES": "3"}, [("uni", "éa", dict())]),
    ({"FZB_NO_DP_CLASSES": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_NO_DP_CFM": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_NO_TAIL_CLASSES": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_SMALL_LIST": "0"}, [("ragged", "deadbeef", dict())]),                      # four scorer launches on two streams
    ({"FZB_SMALL_LIST": "0", "FZB_NO_OVERLAP": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_HANDOFF_MIN_TILES": "0"}, [("wide", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1))]),  # the 16-vector view kernel, staging
    ({"FZB_NO_OVERLAP": "1"}, [("uniwide", "éa", dict(max_typos=None))]),      # whole-haystack unicode windows: the scorer queues the wide ones itself, one stream
    ({"FZB_UNICODE_MULTI": "1"}, [("uniwide", "éa", dict(max_typos=None))]),   # ... queued ahead (default), thread per haystack beside the single-chunk scorer
    ({"FZB_UNICODE_MULTI": "0"}, [("uniwide", "éa", dict(max_typos=None)), ("uniwide", "éa", dict())]),
    ({"FZB_UNICODE_MULTI": "1", "FZB_UNICODE_FWD": "0"}, [("uniwide", "éa", dict(max_typos=None)), ("uniwide", "éa", dict())]),  # the thread-per-haystack scorer keeps its stragglers
    ({"FZB_UNICODE_MULTI": "1"}, [("uniwide", "éa", dict()), ("uniwide", "éa", dict(max_typos=1))]),                               # ... hands them on (default)
    ({"FZB_WINDOW_FOUR_PASS": "1"}, [("ragged", "deadbeef", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=1)), ("uni", "éa", dict(max_typos=1))]),  # the lane-exact window kernel, 256-thread form
    ({"FZB_WINDOW_NO_MASK_CACHE": "1"}, [("ragged", "deadbe", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=1)), ("uni", "éa", dict(max_typos=2))]),  # row masks recomputed at every request
    ({"FZB_WINDOW_NO_MASK_CACHE": "1", "FZB_WINDOW_FOUR_PASS": "1"}, [("uniwide", "éa", dict(max_typos=1))]),
    ({"FZB_DFA_WGS": "3"}, [("short", "deadbe", dict()), ("short", "deadbe", dict(max_typos=2))]),
    ({"FZB_DFA_UNI32": "1"}, [("short", "deadbe", dict()), ("short", "deadbe", dict(max_typos=1))]),          # k1_dfa without per-lane lengths on the uniform 32-byte list
    ({"FZB_DFA_STRIDE256": "1"}, [("short", "deadbe", dict()), ("short", "deadbe", dict(max_typos=2))]),      # the table at a 256-byte row pitch (v_perm result = address)
    ({"FZB_DFA_STRIDE256": "1", "FZB_DFA_UNI32": "1"}, [("short", "deadbe", dict())]),
    ({"FZB_WINDOW_WHOLE_TILES": "1"}, [("ragged", "deadbeef", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=3))]),  # the PRE window kernel as one 1024-thread workgroup per tile
    ({"FZB_DFA_WGS": "8"}, [("short", "deadbe", dict()), ("uni", "éa", dict())]),
    ({"FZB_PARK_LDS_KB": "0"}, [("ragged", "deadbeef", dict()), ("ragged", "deadbeef", dict(max_typos=1))]),  # parked rows in the global slab
    ({"FZB_PARK_LDS_KB": "0", "FZB_SMALL_LIST": "0"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_NO_HANDOFF": "1"}, [("ragged", "deadbeef", dict()), ("ragged", "deadbeef", dict(max_typos=1))]),
    ({"FZB_HANDOFF": "1", "FZB_HANDOFF_MIN_TILES": "8"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict())]),  # the handoff switched on (off by default since round 5), from 8 tiles
    ({"FZB_WINDOW_NO_PRE": "1"}, [("ragged", "deadbeef", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=1)), ("uniwide", "éa", dict(max_typos=2)), ("uni", "éa", dict(max_typos=1))]),  # the window kernel's threads compute their own masks (round 4's one-pass form)
    ({"FZB_HANDOFF_MIN_TILES": "0"}, [("ragged", "deadbeef", dict()), ("ragged", "DeadBeef", dict())]),  # the handoff on a small list (default: big lists only)
    ({"FZB_HANDOFF_MIN_TILES": "0", "FZB_VIEW_PLAIN_LOADS": "1"}, [("ragged", "deadbeef", dict())]),
    ({"FZB_VIEW_READ_LEN": "1"}, [("ragged", "deadbeef", dict()), ("wide", "deadbeef", dict(max_typos=1)), ("uniwide", "éa", dict())]),  # the view filter reads the lengths it does not need (round 4's form)
    ({"FZB_NO_CDFA": "1"}, [("ragged", "deadbeef", dict())]),                         # the burst filter over the byte automaton
    ({"FZB_NO_CDFA": "1", "FZB_RAGGED_BURST": "0"}, [("ragged", "deadbeef", dict())]),  # ... and its rolling form
    ({"FZB_DEBUG_SYNC": "1"}, [("short", "deadbe", dict())]),
    ({"FZB_COMPACT_GRID_MUL": "2", "FZB_CLASSIFY_PER": "4", "FZB_DP_WGS_PER_CU": "2", "FZB_VIEW_WGS": "3"}, [("short", "deadbe", dict()), ("ragged", "deadbeef", dict())]),
    ({"FZB_CLASSIFY_PER": "1", "FZB_CDFA_WGS": "3", "FZB_RAGGED_WGS": "4"}, [("ragged", "deadbeef", dict())]),
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
