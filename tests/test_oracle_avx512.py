"""The CPU-baseline build of the oracle (oracle/libfrizbee_oracle_native.so: -march=native, lane vectors in zmm
registers at the widths of the reference's AVX-512 backend) must agree with the portable lane-by-lane build on
everything: it is the same algorithm text over a different vector type.  Skipped on hosts without AVX-512 BW/VL/VBMI."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.skipif(O.simd_kind(True) != "avx512", reason="host CPU has no AVX-512 BW/VL/VBMI: the native build is the portable code")

ALPHA = b"abcABC_-/ 01xyzdeDEf.:"


def _rand_bytes(rng, n, alpha=ALPHA):
    return bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), n))


def test_portable_build_is_portable():
    assert O.simd_kind(False) == "portable"


def test_prefilter_and_sw_primitives_agree():
    rng = np.random.default_rng(7)
    for it in range(1500):
        needle = _rand_bytes(rng, int(rng.integers(1, 14)))
        hay = bytearray(_rand_bytes(rng, int(rng.choice([0, 1, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 300]))))
        if len(hay) > len(needle) and rng.random() < 0.6:  # plant the needle with holes
            pos = np.sort(rng.choice(len(hay), len(needle), replace=False))
            for p, c in zip(pos, needle):
                hay[p] = c
        hay = bytes(hay)
        cs = bool(rng.integers(0, 2))
        for typos in (0, 1, 2, 3):
            a = O.prefilter(needle, hay, typos, cs, False, 64)
            b = O.prefilter(needle, hay, typos, cs, False, 64, native=True)
            assert a == b, (needle, hay, typos, cs)
        for lanes, u8 in ((64, True), (32, False)):
            for prefix in (True, False):
                a = O.sw_score(needle, hay, None, cs, prefix, False, lanes, u8)
                b = O.sw_score(needle, hay, None, cs, prefix, False, lanes, u8, native=True)
                assert a == b, (needle, hay, lanes, u8, prefix)


def test_unicode_primitives_agree():
    rng = np.random.default_rng(8)
    pool = ["a", "B", "é", "É", "ن", "إ", "न", "😀", " ", "_", "x", "ß"]
    for it in range(600):
        needle = "".join(pool[int(i)] for i in rng.integers(0, len(pool), int(rng.integers(1, 6))))
        hay = "".join(pool[int(i)] for i in rng.integers(0, len(pool), int(rng.integers(0, 70))))
        cs = bool(rng.integers(0, 2))
        for typos in (0, 1, 2):
            assert O.prefilter(needle, hay, typos, cs, True, 64) == O.prefilter(needle, hay, typos, cs, True, 64, native=True), (needle, hay, typos)
        for lanes, u8 in ((64, True), (32, False)):
            assert O.sw_score(needle, hay, None, cs, True, True, lanes, u8) == O.sw_score(needle, hay, None, cs, True, True, lanes, u8, native=True), (needle, hay)


@pytest.mark.parametrize("typos", [0, 1, 2, None])
def test_match_list_agrees_end_to_end(typos):
    rng = np.random.default_rng(9 if typos is None else typos)
    hs = []
    for _ in range(20000):
        h = bytearray(_rand_bytes(rng, int(rng.integers(0, 140)), b"abcdefDEADBEEF0123_-/ xyz"))
        hs.append(bytes(h))
    for needle in ("deadbe", "dEadbeef", "a_b", "deadbeefdeadbeef"):  # the last one is the u16 class (32 lanes)
        kw = dict(max_typos=typos)
        a = O.Matcher(needle, **kw).match_list(hs)
        b = O.Matcher(needle, native=True, **kw).match_list(hs)
        assert a.tolist() == b.tolist(), (needle, typos)
        assert O.Matcher(needle, native=True, **kw).match_list_parallel(hs, 4).tolist() == a.tolist()
