"""Unicode typo queries (round 6): over a list whose haystacks all fit ONE prefilter chunk the streaming filter runs the scalar-level LCS
automaton and that IS the reference's decision (unicode_typos.rs:15-466 on a single chunk; the CPU evidence is
tests/test_oracle_reference_properties.py::test_single_chunk_unicode_typo_prefilter_is_the_scalar_lcs_criterion and
tests/test_host_abi.py::test_scalar_lcs_automaton_...); the scorer computes the lane-free typo window itself.  Longer lists keep the automaton
as the superset in front of the lane-exact window kernel.  Both against the oracle, record for record, at every emulated lane pair."""
import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O
from test_gpu_parity import assert_same, both

pytestmark = pytest.mark.gpu

ALPHA = ["a", "b", "A", "_", " ", "é", "É", "ж", "Ж", "다", "😀", "1", "ن", "إ", "م", "ا"]


def _list(rng, n, max_bytes, needle, exact_len=False):
    out = []
    for _ in range(n):
        budget = max_bytes if exact_len else int(rng.integers(0, max_bytes + 1))
        chars, size = [], 0
        while True:
            c = ALPHA[int(rng.integers(0, len(ALPHA)))]
            if size + len(c.encode()) > budget:
                break
            chars.append(c)
            size += len(c.encode())
        if rng.random() < 0.4 and len(chars) >= len(needle):  # plant the needle (with one or two scalars knocked out)
            pos = np.sort(rng.choice(len(chars), len(needle), replace=False))
            for q, c in zip(pos, needle):
                if rng.random() < 0.85:
                    chars[q] = c if rng.random() < 0.8 else c.swapcase()
            while len("".join(chars).encode()) > max_bytes:
                chars.pop()
        out.append("".join(chars))
    return out


@pytest.mark.parametrize("pf", [64, 32, 16])
@pytest.mark.parametrize("needle,k", [("إنما", 1), ("إنما", 2), ("éa", 1), ("aÉжb", 2), ("다😀a다", 1), ("aébécé", 3), ("жж", 1)])
def test_single_chunk_lists_are_decided_in_the_stream(pf, needle, k):
    rng = np.random.default_rng(pf * 1000 + k * 100 + sum(ord(c) for c in needle) % 97)
    hs = _list(rng, 20000, pf, needle)  # every haystack fits one prefilter chunk of this lane pair
    hs[7] = ""
    got, want, fm = both(needle, hs, pf=pf, max_typos=k)
    assert len(want) > 500
    assert_same(got, want, f"{needle} k={k} pf={pf}")
    c = fm.last_counters()
    assert c["kept_by_exact_prefilter"] == c["filter_survivors"] == len(want)  # nothing went through the lane-exact window kernel
    for casing in ("Ignore", "Respect"):
        g2, w2, _ = both(needle, hs, pf=pf, max_typos=k, casing=casing)
        assert_same(g2, w2, f"{needle} k={k} pf={pf} {casing}")


def test_arbitrary_bytes_and_the_u16_score_class():
    # not UTF-8 (the C ABI takes any bytes): needle scalars between stray continuation / lead bytes and truncated scalars; and a scoring whose
    # scores need 16 bits, so that a 33..64-byte window spans two 32-lane score chunks (queued for the multi-chunk unicode scorers)
    rng = np.random.default_rng(5)
    needle = "إنما"
    pool = [c.encode() for c in ALPHA] + [b"\x80", b"\xd8", b"\xd9", b"\xf0\x9f", b"\xa5", b"\x86"] + [bytes([int(x)]) for x in rng.integers(0, 256, 8)]
    rows = []
    for _ in range(20000):
        ln = int(rng.integers(0, 65))
        rows.append(b"".join(pool[int(rng.integers(0, len(pool)))] for _ in range(ln))[:ln])
    ends, data, off = [], bytearray(), 0
    for r in rows:
        data += r
        off += len(r)
        pad = (-off) % 16
        data += b"\0" * pad
        ends.append(off)
        off += pad
    data += b"\0" * 96
    packed = (np.frombuffer(bytes(data), np.uint8), np.array(ends, np.uint64))
    for k in (1, 2, 3):
        got, want, _ = both(needle, None, pf=64, packed=packed, max_typos=k)
        assert len(want) > 300
        assert_same(got, want, f"bytes k={k}")
    big = (12, 6, 5, 1, 40, 30, 4, 8, 4)  # prefix / delimiter bonuses that push the score class to u16: 32 score lanes beside 64 prefilter lanes
    assert not O.score_fits_in_u8(len(needle.encode()), big)
    got, want, _ = both(needle, None, pf=64, packed=packed, max_typos=1, scoring=big)
    assert_same(got, want, "u16 class")


def test_longer_lists_keep_the_window_kernel_behind_the_tighter_filter():
    rng = np.random.default_rng(6)
    hs = _list(rng, 15000, 200, "إنما") + ["إ" + "x" * 1100 + "نما", "ن" * 300 + "إنما"]
    for k in (1, 2):
        got, want, fm = both("إنما", hs, pf=64, max_typos=k)
        assert_same(got, want, f"long k={k}")
        c = fm.last_counters()
        assert c["kept_by_exact_prefilter"] <= c["filter_survivors"]
        # the scalar automaton is exact up to the reference's multi-chunk deviations: all but a handful of its survivors are kept
        assert c["filter_survivors"] - c["kept_by_exact_prefilter"] <= max(5, c["filter_survivors"] // 1000), c
