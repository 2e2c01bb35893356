"""The ragged streaming filter over the interleaved filter view (k1_cdfa_view<SAN, G, NV>) and what follows it on a ragged list (classifier,
class launches, multi-chunk tail classes) against the oracle: every vector-count class of the view (longest haystack 33..64 / ..128 / ..256
bytes), both composition widths of the class-composite automaton (4 bytes per lookup for short needles, 2 for needles with many classes),
every automaton the filter runs (subsequence, LCS for typos, unicode), a needle with a NUL byte (bytes behind a haystack's end must be
sanitised), lists that end inside a tile and inside a 64-haystack group, empty haystacks, and lists too small for a tile."""
import random

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O
from test_gpu_parity import assert_same, both

pytestmark = pytest.mark.gpu


def _list(rng, n, max_len, needle, alphabet, plant=0.3):
    out = []
    for i in range(n):
        L = rng.choice([0, 1, max_len, max_len - 1]) if rng.random() < 0.05 else rng.randint(1, max_len)
        s = [rng.choice(alphabet) for _ in range(L)]
        if rng.random() < plant and L >= len(needle):
            for q, c in zip(sorted(rng.sample(range(L), len(needle))), needle):
                s[q] = c if rng.random() < 0.8 else c.swapcase()
        out.append("".join(s))
    out[rng.randrange(n)] = "".join(rng.choice(alphabet) for _ in range(max_len))  # the list's longest haystack decides the kernel's vector count
    return out


NEEDLES = [
    ("ab", dict()),
    ("deadbeef", dict()),
    ("abcdefghijklmn", dict()),                 # 15 byte classes: two bytes per composite lookup
    ("DeadBeef", dict()),                       # Smart casing: case-sensitive
    ("deadbe", dict(max_typos=1)),              # LCS automaton, then the lane-exact window kernel
    ("abcabc", dict(max_typos=2)),
]


@pytest.mark.parametrize("pf", [64, 32])
@pytest.mark.parametrize("max_len", [40, 64, 65, 128, 129, 256])
def test_ragged_lists_of_every_vector_class(max_len, pf):
    rng = random.Random(31 * max_len + pf)
    alphabet = "abcdefghijklmnDEAB_-/. 01xyz"
    for needle, cfg in NEEDLES:
        n = rng.choice([20000, 1024 * 3 + 517, 700])
        hs = _list(rng, n, max_len, needle, alphabet)
        got, want, _ = both(needle, hs, pf=pf, **cfg)
        assert len(want) > 0
        assert_same(got, want, f"{needle!r} {cfg} max_len={max_len} n={n} pf={pf}")


@pytest.mark.parametrize("max_len", [48, 100, 200])
def test_ragged_unicode_lists(max_len):
    rng = random.Random(977 + max_len)
    alphabet = list("abéÉñ人_ -/xyzüß")
    for needle in ("éa", "人b", "ñé_"):
        hs = []
        for i in range(6000):
            s = ""
            target = rng.randint(1, max_len)
            while True:
                c = rng.choice(alphabet)
                if len((s + c).encode()) > target:
                    break
                s += c
            if rng.random() < 0.3 and len(s) >= len(needle):
                lst = list(s)
                for q, c in zip(sorted(rng.sample(range(len(lst)), len(needle))), needle):
                    lst[q] = c
                cand = "".join(lst)
                if len(cand.encode()) <= max_len:
                    s = cand
            hs.append(s)
        hs[rng.randrange(len(hs))] = "x" * max_len
        got, want, _ = both(needle, hs, pf=64)
        assert len(want) > 0
        assert_same(got, want, f"unicode {needle!r} max_len={max_len}")


def test_needle_with_a_nul_byte_on_a_ragged_list():
    # the zero padding behind a haystack would match a NUL needle byte: the view kernel sanitises the last vector (SAN), the scorers take
    # their first forms (no closed-form padding)
    rng = random.Random(5)
    alphabet = "ab\x00cd_-"
    for needle in ("a\x00b", "\x00a"):
        hs = _list(rng, 5000, 100, needle, alphabet, plant=0.4)
        got, want, _ = both(needle, hs, pf=64)
        assert len(want) > 0
        assert_same(got, want, repr(needle))


def test_tiny_ragged_lists():
    rng = random.Random(9)
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025):
        hs = _list(rng, n, 90, "deadbeef", "deabfxyz_", plant=0.5)
        got, want, _ = both("deadbeef", hs, pf=64)
        assert_same(got, want, f"n={n}")
