"""Unicode windows wider than one chunk on the GPU: k2u_dp_unicode_multi (thread per haystack, chunk by chunk, the previous chunk's top
half parked per needle row; dp_unicode.h dp_unicode_multi_chunk) for windows up to 1024 bytes and the wave-per-haystack kernel's greedy
fallback beyond - against the oracle (score_haystack_unicode over its chunks, src/smith_waterman/algo/unicode.rs:10-217; match_greedy,
greedy.rs:7-91) at the three lane pairs, with and without a prefilter (0 typos, typos, All Scores), both score classes."""
import random

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O
from test_gpu_parity import assert_same, both

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["wave_per_haystack", "thread_per_haystack"])
def wide_mode(request):
    """Who scores the windows of 65..1024 bytes: the wave-per-haystack kernel (FZB_UNICODE_MULTI=0; also what the default chooses on the
    device for queues as short as these) or k2u_dp_unicode_multi (FZB_UNICODE_MULTI=1; the default's choice from 32 768 windows on) -
    which hands windows beyond four chunks on to the wave-per-haystack kernel"""
    import os
    os.environ["FZB_UNICODE_MULTI"] = "0" if request.param == "wave_per_haystack" else "1"
    F.lib().fzb_debug_reload_knobs()
    yield request.param
    os.environ.pop("FZB_UNICODE_MULTI", None)
    F.lib().fzb_debug_reload_knobs()


ALPHA = list("abéÉñ人_ -/xyzüßإنما")


def _sentence(rng, nbytes, needle, plant):
    s = ""
    while True:
        c = rng.choice(ALPHA)
        if len((s + c).encode()) > nbytes:
            break
        s += c
    if plant and len(s) >= len(needle):
        lst = list(s)
        for q, c in zip(sorted(rng.sample(range(len(lst)), len(needle))), needle):
            lst[q] = c
        cand = "".join(lst)
        if len(cand.encode()) <= nbytes + 8:
            s = cand
    return s


@pytest.mark.parametrize("pf", [64, 32, 16])
def test_wide_unicode_windows_against_the_oracle(pf, wide_mode):
    rng = random.Random(8800 + pf)
    sizes = [20, 60, 70, 100, 130, 200, 400, 700, 1000, 1024, 1030, 1500]
    hs = [_sentence(rng, rng.choice(sizes), "éa人", rng.random() < 0.5) for _ in range(6000)]
    hs[17] = "é" + "x" * 1010 + "a人"          # a window of exactly the whole 1024-byte limit region
    hs[18] = "é" + "ü" * 600 + "a人"           # > 1024 bytes: greedy fallback
    for needle, cfg in (("éa人", dict()), ("éa", dict(max_typos=1)), ("ña", dict(max_typos=None)), ("إن", dict(max_typos=None)),
                        ("É_", dict(max_typos=0, casing="Respect")), ("ab", dict(max_typos=None, unicode="Always"))):
        got, want, fm = both(needle, hs, pf=pf, **cfg)
        assert len(want) > 0
        assert_same(got, want, f"{needle!r} {cfg} pf={pf}")
        c = fm.last_counters()  # windows of 65..1024 bytes (either scorer's queue) and windows beyond 1024 bytes (greedy fallback)
        assert c["multi_chunk_scored"] > 0 and c["generic_scored"] > 0, (needle, cfg, c)
    # a needle of the u16 score class (more rows than a byte's worth of score): the parked rows are not packed to bytes
    needle = "éa人_üñ" * 3
    hs2 = [_sentence(rng, rng.choice([150, 300, 600]), needle, rng.random() < 0.6) for _ in range(1500)]
    got, want, fm = both(needle, hs2, pf=pf, max_typos=None)
    assert_same(got, want, f"u16 class pf={pf}")
    assert not fm.info()["use_u8"] and fm.last_counters()["multi_chunk_scored"] > 0


def test_default_chooses_by_queue_length():
    """No knob: the device picks the scorer by the queue's length - the wave-per-haystack kernel for this list's few hundred wide windows,
    k2u_dp_unicode_multi beyond 32 768 (a list of 150 000 wide windows)."""
    import os
    os.environ.pop("FZB_UNICODE_MULTI", None)
    F.lib().fzb_debug_reload_knobs()
    rng = random.Random(3)
    few = [_sentence(rng, rng.choice([30, 90, 150]), "éa", True) for _ in range(3000)]
    got, want, _ = both("éa", few, pf=64, max_typos=None)
    assert_same(got, want, "short queue")
    base = [_sentence(rng, rng.choice([80, 100, 130]), "éa", rng.random() < 0.5) for _ in range(3000)]
    many = base * 50
    got, want, fm = both("éa", many, pf=64, max_typos=None)
    assert fm.last_counters()["multi_chunk_scored"] > 131072
    assert_same(got, want, "long queue")


def test_all_scores_over_an_arabic_shaped_list():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth
    data, ends = synth.arabic_corpus(n=40_000)
    for cfg in (dict(max_typos=None), dict(max_typos=1), dict(max_typos=0)):
        got, want, fm = both("إن", None, pf=64, packed=(data, ends), **cfg)
        assert len(want) > 1000
        assert_same(got, want, f"arabic {cfg}")


def test_forwarded_stragglers_never_land_on_unread_front_entries(wide_mode):
    """More wide windows than the thread-per-haystack scorer has threads (256 CUs x 256 = 65 536: entries beyond that are read AFTER
    earlier iterations have forwarded their stragglers), every haystack wider than a chunk and a few thousand wider than four: front +
    forwarded > the list's size, so before the back of the queue got 4096 entries of its own the forwarded entries could overwrite front
    entries that had not been scored yet (round 4 advisor finding, host.hip `qcap`)."""
    rng = random.Random(77)
    base = [_sentence(rng, rng.choice([80, 100, 130, 200, 300, 400]), "éa", rng.random() < 0.5) for _ in range(4000)]
    many = base * 40  # 160 000 windows under All Scores, all beyond 64 bytes, about half of them beyond 256
    assert min(len(s.encode()) for s in base) > 64
    got, want, fm = both("éa", many, pf=64, max_typos=None)
    assert len(want) == len(many)
    assert_same(got, want, "queue front vs forwarded back")
    if wide_mode != "wave_per_haystack":
        assert fm.last_counters()["multi_chunk_scored"] >= len(many) - 4096
