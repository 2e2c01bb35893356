"""Four lanes per window (dp_quad.h: sixteen windows per wavefront, dp_cfm.h's arithmetic over DPP row shifts) against the oracle: multi-chunk ASCII
windows of 65..1024 bytes at the 64-lane (u8 class) and 32-lane (u16 class) backends - windows of up to 3 (4) chunks run row by row in registers,
wider ones chunk by chunk with their rows parked in LDS - needles of 1..20 rows, random scorings, windows that end exactly on chunk boundaries,
typo configurations (windows from the lane-exact prefilter).  It is the DEVICE's choice below 49 152 queued windows, so these small lists take it
by default; FZB_COOP_BELOW=0 is the thread-per-window form (tests/test_gpu_knobs.py runs both).  Long needles (k2d_dp_long_quad: the same two
forms, the parked rows in a global slab): tests/test_gpu_long_needles.py and the last test here."""
import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O
from test_gpu_parity import assert_same, both

pytestmark = pytest.mark.gpu
ALPHA = "abcdeABCDE_-/ .0123xyz"


def _hay(rng, L, needle, plant):
    s = [ALPHA[int(x)] for x in rng.integers(0, len(ALPHA), L)]
    if plant and L >= len(needle):
        for q, c in zip(np.sort(rng.choice(L, len(needle), replace=False)), needle):
            s[q] = c if rng.random() < 0.8 else c.swapcase()
    return "".join(s)


@pytest.mark.parametrize("pf", [64, 32])
@pytest.mark.parametrize("needle", ["d", "de", "deadbeef", "a_b-c/d.e", "Dead", "abcdeabcdeabcdeabcde"])
def test_multi_chunk_windows_of_a_ragged_list(pf, needle):
    rng = np.random.default_rng(len(needle) * 100 + pf)
    lens = [65, 64, 63, 66, 127, 128, 129, 191, 192, 193, 256, 257, 511, 512, 513, 1000, 1023, 1024, 33, 8]
    hs = [_hay(rng, int(lens[i % len(lens)]) if i % 3 == 0 else int(rng.integers(1, 300)), needle, rng.random() < 0.7) for i in range(6000)]
    got, want, fm = both(needle, hs, pf=pf, max_typos=0)
    assert fm.last_counters()["multi_chunk_scored"] > 500 and len(want) > 1500
    assert_same(got, want, f"{needle} pf={pf}")
    for k in (1, None):
        g2, w2, _ = both(needle, hs, pf=pf, max_typos=k)
        assert_same(g2, w2, f"{needle} pf={pf} typos={k}")


def test_u16_score_class_and_random_scorings():
    rng = np.random.default_rng(77)
    needle = "deadbeefdeadbeef"  # 16 rows: 18 * 16 > 255 -> the u16 class, 32 score lanes beside 64 prefilter lanes
    assert not O.score_fits_in_u8(len(needle))
    hs = [_hay(rng, int(rng.integers(16, 400)), needle, rng.random() < 0.8) for _ in range(5000)]
    got, want, fm = both(needle, hs, pf=64, max_typos=0)
    assert fm.last_counters()["multi_chunk_scored"] > 500
    assert_same(got, want, "u16 class")
    for _ in range(6):
        sc = [int(rng.integers(1, 20)), int(rng.integers(2, 12)), int(rng.integers(1, 9)), int(rng.integers(0, 3)), int(rng.integers(0, 16)), int(rng.integers(0, 9)),
              int(rng.integers(0, 9)), int(rng.integers(0, 12)), int(rng.integers(0, 9))]
        sc[3] = min(sc[3], sc[1] // 2)  # 2 * gap_extend <= mismatch_penalty: dp_cf.h's precondition keeps the classified path (the one with the four-lane slice)
        g2, w2, _ = both("deadbeef", hs, pf=64, max_typos=0, scoring=tuple(sc))
        assert_same(g2, w2, f"scoring {sc}")


@pytest.mark.parametrize("pf", [64, 32])
def test_long_needle_windows_on_both_sides_of_the_register_form(pf):
    """k2d_dp_long_quad (pf 64: 32 x u16 score lanes): windows of up to 256 bytes are scored row by row with every chunk in registers, wider ones
    chunk by chunk with the rows parked in the global slab one row ahead; a list that holds both (and windows exactly 256 / 257 bytes wide), 70-
    and 130-row needles.  pf 32 (16 x u16) has no four-lane form: the thread-per-window scorer, same list."""
    rng = np.random.default_rng(900 + pf)
    for n in (70, 130):
        low = "abcde_-/ .0123xyz"  # (no uppercase letter: smart case stays insensitive, so the planted needle matches in either case)
        needle = "".join(low[int(x)] for x in rng.integers(0, len(low), n))
        lens = [n, n + 1, 255, 256, 257, 258, 300, 511, 512, 513, 1000, 1024]
        hs = [_hay(rng, int(lens[i % len(lens)]) if i % 2 == 0 else int(rng.integers(n, 700)), needle, rng.random() < 0.85) for i in range(1500)]
        got, want, fm = both(needle, hs, pf=pf, max_typos=0)
        assert len(want) > 500
        assert_same(got, want, f"long needle {n} rows pf={pf}")
