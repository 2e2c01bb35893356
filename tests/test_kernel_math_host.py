"""The DEVICE arithmetic of the Smith-Waterman scorers (frizbee_amd/csrc/dp_body.h, dp_cf.h, dp_cfm.h, dp_unicode.h), compiled for the
host (tests/kernel_host: ROCm's clang++ with a stand-in hip_runtime.h) and compared with the oracle's score_haystack
(src/smith_waterman/algo/ascii.rs:10-158) - so the closed-form padding, the biased domain and the skipped last-row scan of dp_cf.h
are checked bit for bit without a GPU.  The GPU parity tests run the same headers through hipcc."""
import ctypes as C
import itertools
import random

import pytest

import kernel_host_lib as K
import oracle_lib as O

pytestmark = pytest.mark.skipif(not K.available(), reason="ROCm clang++ not installed")

DEF = [12, 6, 5, 1, 12, 4, 4, 8, 4]


def _fits(n, sc):
    return bool(O.lib().fzo_score_fits_in_u8(n, (C.c_uint16 * 9)(*sc)))


def _check(needle, hays, sc, cs, ips, swl, form, real):
    got = K.dp_batch(needle, hays, sc, cs, ips, swl, form, real)
    u8 = _fits(len(needle), sc)
    scc = (C.c_uint16 * 9)(*sc)
    f = O.lib().fzo_sw_score
    for h, ip, g in zip(hays, ips, got):
        want = f(needle, len(needle), h, len(h), scc, int(cs), int(ip), 0, swl, int(u8))
        assert g == want, (needle, h, sc, cs, ip, swl, form, real, int(g), want)


def _rnd(rng, n, alpha):
    return bytes(rng.choice(alpha) for _ in range(n))


def test_reference_known_answers_through_the_device_arithmetic():
    # src/smith_waterman/mod.rs:208-298 (scalar LANES = 8; the same at every width per backend/tests/parity.rs:95-124)
    cases = [(b"b", b"abc", 16), (b"a", b"abc", 28), (b"a", b"babc", 16), (b"abc", b"abc", 60), (b"b", b"a-b", 20), (b"a", b"-a--bc", 20),
             (b"test", b"Uteost", 59), (b"test", b"Uteoost", 58), (b"test", b"Utooooeoooosoooot", 40), (b"a", b"A", 24), (b"D", b"forDist", 20),
             (b"D", b"foRDist", 16), (b"D", b"FOR_DIST", 20), (b"foo", b"Ufo", 27), (b"foo", b"Uf", 10), (b"foo", b"U", 0)]
    for needle, hay, want in cases:
        for swl in (64, 32):
            nw = swl // 2
            for real in (nw // 2, 3 * nw // 4, nw):
                if len(hay) <= 2 * real:
                    cs = any(65 <= c <= 90 for c in needle)  # smart case, as the tests' `score` helper resolves it
                    assert K.dp_single(needle, hay, DEF, case_sensitive=cs, swl=swl, form=3, real=real) == want, (needle, hay, swl, real)


@pytest.mark.parametrize("swl", [64, 32, 16, 8])
def test_random_windows_all_forms_match_the_oracle(swl):
    rng = random.Random(1000 + swl)
    nw = swl // 2
    reals = sorted({max(1, nw // 4), nw // 2, 3 * nw // 4 if nw >= 4 else nw, nw})
    for it in range(400):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        needle = _rnd(rng, rng.randint(1, 12), alpha)
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.6:
            while True:
                sc = [rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 20), rng.randint(0, 6), rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12),
                      rng.randint(0, 20), rng.randint(0, 12)]
                if 2 * sc[3] <= sc[1]:
                    break
        real = rng.choice(reals)
        hays = [_rnd(rng, rng.randint(1, 2 * real), alpha) for _ in range(12)]
        ips = [rng.random() < 0.5 for _ in hays]
        _check(needle, hays, sc, cs, ips, swl, 3, real)
        if swl >= 16:  # the short-haystack kernel's set-up (LDS tables) in front of the same rows
            _check(needle, [h[: swl // 2] for h in hays], sc, cs, ips, swl, 4, 0)
        # the first form (dp_body.h) on the same inputs: biased scan, literal scan, padded half
        _check(needle, hays, sc, cs, ips, swl, 0, 0)
        _check(needle, hays, sc, cs, ips, swl, 1, 0)
        if real <= max(1, nw // 2) and swl >= 16:
            _check(needle, [h[: swl // 2] for h in hays], sc, cs, ips, swl, 2, 0)


def test_padding_entries_exhaustively_on_a_small_chunk():
    # 8-lane chunk, 4 computed lanes: every window over {a, b, q} up to 4 bytes x every needle over {a, b} up to 6 rows x gap scorings
    hays = [bytes(t) for n in range(1, 5) for t in itertools.product(b"abq", repeat=n)]
    needles = [bytes(t) for n in range(1, 7) for t in itertools.product(b"ab", repeat=n)]
    for e, o, x in ((0, 3, 0), (1, 4, 6), (1, 10, 2), (2, 0, 5), (0, 10, 3)):
        sc = [12, x, o + e, e, 6, 4, 4, 8, 4]
        for needle in needles:
            _check(needle, hays, sc, False, [True] * len(hays), 8, 3, 2)


@pytest.mark.parametrize("swl,real", [(64, 16), (64, 24), (64, 8), (32, 8), (32, 12), (16, 4), (16, 2), (16, 6)])
def test_full_windows_with_unmatched_needle_tails(swl, real):
    """The inputs on which the zero padding right of the computed lanes decides the score: the needle's head is matched near the
    end of a window that fills the computed lanes, its tail matches nothing, so the best alignment leaves the window through the
    padding (a variant of dp_cf.h without its padding term fails ~4 % of these)."""
    rng = random.Random(77 + swl + real)
    P = 2 * real
    for it in range(250):
        alpha = rng.choice([b"ab", b"abc", b"abcdeABC_"])
        sc = DEF
        if rng.random() < 0.6:
            while True:
                sc = [rng.randint(1, 30), rng.randint(0, 12), rng.randint(0, 30), rng.randint(0, 3), rng.randint(0, 20), rng.randint(0, 8), rng.randint(0, 8), rng.randint(0, 20),
                      rng.randint(0, 8)]
                if 2 * sc[3] <= sc[1]:
                    break
        for _ in range(6):
            t = rng.randint(1, 5)
            head = _rnd(rng, t, alpha)
            needle = head + _rnd(rng, rng.randint(1, 6), b"xyz")
            m = rng.randint(max(1, P - 2), P)
            body = bytearray(_rnd(rng, m, alpha + b"q"))
            pos = rng.choice([m - t, m - t, rng.randint(0, max(0, m - t))])
            if pos >= 0:
                body[pos:pos + t] = head
            for s in (1, 2, 4, 8, 16, 32):
                if rng.random() < 0.4 and 0 <= P - s < m:
                    body[P - s] = rng.choice(head)
            _check(needle, [bytes(body[:m])], sc, False, [rng.random() < 0.5], swl, 3, real)
            if real == swl // 4:
                _check(needle, [bytes(body[:m])], sc, False, [rng.random() < 0.5], swl, 4, 0)


def test_short_kernel_window_search_equals_the_first_form():
    """cf_window_first_last_regs (one merged flag word per needle byte) against window_first_last_regs, on haystacks of 0..32 bytes
    that contain the needle's first and last byte (what a survivor of the exact filter guarantees), both case modes."""
    rng = random.Random(4242)
    for it in range(20000):
        alpha = rng.choice([b"ab", b"abAB_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        needle = _rnd(rng, rng.randint(1, 8), alpha)
        cs = rng.random() < 0.3
        L = rng.randint(1, 32)
        hay = bytearray(_rnd(rng, L, alpha))
        hay[rng.randrange(L)] = needle[0]
        hay[rng.randrange(L)] = needle[-1]
        if needle[0] not in hay:
            hay[0] = needle[0]
        a, b = K.window(needle, bytes(hay), cs)
        assert a == b, (needle, bytes(hay), cs, a, b)


def test_short_kernel_typo_window_equals_the_reference_prefilter_window():
    """cf_window_typos_regs (what k2b_dp_short computes for an accepted haystack under max_typos >= 1) against the oracle's chunked
    multi-path prefilter (src/prefilter/algo/ascii_typos.rs) at all three lane widths, on every input the prefilter accepts."""
    rng = random.Random(99)
    checked = 0
    for it in range(12000):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefAB_- 01"])
        n = rng.randint(2, 9)
        needle = _rnd(rng, n, alpha)
        k = rng.randint(1, 3)
        if k >= n:
            continue
        L = rng.randint(1, 32)
        hay = bytearray(_rnd(rng, L, alpha))
        if rng.random() < 0.6 and L >= n:
            for q, c in zip(sorted(rng.sample(range(L), n)), needle):
                hay[q] = c
        hay = bytes(hay)
        cs = rng.random() < 0.4
        got = K.window_typos(needle, hay, k, cs)
        for lanes in (16, 32, 64):
            ok, ws, we = O.prefilter(needle, hay, k, cs, False, lanes)
            if ok:
                assert got == (ws, we), (needle, hay, k, cs, lanes, got, (ws, we))
                checked += 1
    assert checked > 8000


@pytest.mark.parametrize("swl", [64, 32, 16, 8])
def test_multi_chunk_windows_both_forms_match_the_oracle(swl):
    """windows wider than one chunk (dp_body.h's dp_multi_chunk and dp_cfm.h's dp_multi_chunk_t): the adjacent chunk's parked rows, the
    carry of the last lane across the chunk boundary, the last chunk's unpropagated last row - against score_haystack's multi-chunk
    loop (ascii.rs:91-158) with propagate_horizontal_gaps's adjacent vector (ascii_gap.rs:11-105)"""
    rng = random.Random(500 + swl)
    for it in range(500):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        n = rng.randint(1, 14)
        needle = _rnd(rng, n, alpha)
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.5:
            while True:
                sc = [rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 20), rng.randint(0, 6), rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12),
                      rng.randint(0, 20), rng.randint(0, 12)]
                if 2 * sc[3] <= sc[1]:  # LaunchCfg::cfm_ok
                    break
        m = rng.randint(swl + 1, min(1024, swl * rng.choice([2, 2, 3, 5, 16])))
        if it % 50 == 0:
            m = rng.choice([swl + 1, 2 * swl, 2 * swl + 1, min(1024, 16 * swl)])
        hay = _rnd(rng, m, alpha)
        if rng.random() < 0.5:
            for q, c in zip(sorted(rng.sample(range(m), min(n, m))), needle):
                hay = hay[:q] + bytes([c]) + hay[q + 1:]
        ip = rng.random() < 0.5
        u8 = _fits(n, sc)
        want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, lanes=swl, is_u8=u8)
        for form in (5, 6, 7, 8, 16, 17, 18):  # 16..18: 6..8 with the parked rows at the LDS layout's pitch
            got = K.dp_multi(needle, hay, sc, cs, ip, swl, form, u8)
            assert got == want, (needle, hay, sc, cs, ip, swl, form, got, want)


@pytest.mark.parametrize("swl", [64, 32, 16])
def test_long_needle_windows_both_forms_match_the_oracle(swl):
    """k2d_dp_long's two bodies (dp_multi_chunk and, round 6, dp_cfm.h's rows) with the needle's rows read through NeedleLongDev's pointers:
    needles of 64..230 bytes, windows of one to many chunks (a single-chunk window is a one-chunk walk there), default and random scorings
    inside LaunchCfg::cfm_ok, both score classes"""
    rng = random.Random(900 + swl)
    checked = 0
    for it in range(160):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        n = rng.choice([64, 65, 80, 100, 127, 128, 129, 200, 230]) if it % 3 else rng.randint(64, 230)
        needle = _rnd(rng, n, alpha)
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.4:
            while True:
                sc = [rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 20), rng.randint(0, 6), rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12),
                      rng.randint(0, 20), rng.randint(0, 12)]
                worst = n * (sc[0] + sc[4] + sc[5] + sc[6] + sc[8]) + sc[1] + (137 + n) * sc[3] + 64
                if 2 * sc[3] <= sc[1] and worst < 0x7C00:  # build_long_needle's cfm_ok (a bound on max_matrix_score is enough here)
                    break
        m = rng.choice([1, swl - 1, swl, swl + 1, 2 * swl, 3 * swl + 5]) if it % 8 == 0 else rng.randint(1, min(1024, swl * rng.choice([1, 2, 3, 5, 9])))
        hay = _rnd(rng, m, alpha)
        if rng.random() < 0.6:  # a good part of the needle, in order
            k = min(n, m)
            for q, c in zip(sorted(rng.sample(range(m), k)), needle[rng.randint(0, n - k):][:k]):
                hay = hay[:q] + bytes([c]) + hay[q + 1:]
        ip = rng.random() < 0.5
        u8 = _fits(n, sc)
        want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, lanes=swl, is_u8=u8)
        for form in (5, 6):
            got = K.dp_multi_long(needle, hay, sc, cs, ip, swl, form, u8)
            assert got == want, (needle, hay, sc, cs, ip, swl, form, got, want)
        checked += 1
    assert checked == 160


@pytest.mark.parametrize("swl", [64, 32])
def test_four_lanes_per_window_match_the_oracle(swl):
    """dp_quad.h - dp_cfm.h's arithmetic spread over four lanes, the reference's shift_right_padded as DPP row shifts whose out-of-row lanes keep the
    previous chunk's values - run by four host threads in lockstep (the shim's update_dpp): windows of 1..1024 bytes row by row with every
    chunk in registers (form 0: k2_classes_all, k2d_dp_long_quad up to 3-8 chunks), chunk by chunk with the rows parked in the LDS layout (1)
    and in the slab layout requested a row ahead (2); short needles by value and long ones through NeedleLongRows; default and random
    scorings inside LaunchCfg::cfm_ok; windows ending on chunk and quad-lane boundaries"""
    rng = random.Random(1300 + swl)
    for it in range(260):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        long_needle = it % 4 == 3
        n = rng.randint(64, 150) if long_needle else rng.randint(1, 20)
        needle = _rnd(rng, n, alpha)
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.45:
            while True:
                sc = [rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 20), rng.randint(0, 6), rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12),
                      rng.randint(0, 20), rng.randint(0, 12)]
                worst = n * (sc[0] + sc[4] + sc[5] + sc[6] + sc[8]) + sc[1] + (200 + n) * sc[3] + 64
                if 2 * sc[3] <= sc[1] and worst < 0x7C00:
                    break
        if it % 6 == 0:
            m = rng.choice([1, swl // 4, swl // 4 + 1, swl // 2, swl - 1, swl, swl + 1, 2 * swl, 2 * swl + swl // 4, 3 * swl, 4 * swl + 1, 1024])
        else:
            m = rng.randint(1, min(1024, swl * rng.choice([1, 2, 3, 5, 16])))
        hay = _rnd(rng, m, alpha)
        if rng.random() < 0.6:
            k = min(n, m)
            for q, c in zip(sorted(rng.sample(range(m), k)), needle[rng.randint(0, n - k):][:k]):
                hay = hay[:q] + bytes([c]) + hay[q + 1:]
        ip = rng.random() < 0.5
        u8 = _fits(n, sc)
        want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, lanes=swl, is_u8=u8)
        for form in (0, 1, 2):
            if form == 2 and n < 2:  # (the slab form is the long-needle kernel's; with ONE row the host threads have no exchange between a park and its read)
                continue
            got = K.dp_quad(needle, hay, sc, cs, ip, swl, form, u8, long_needle)
            assert got == want, (needle, hay, sc, cs, ip, swl, form, long_needle, got, want)


@pytest.mark.parametrize("swl", [64, 32, 16, 8])
def test_multi_chunk_last_chunk_padding_in_closed_form(swl):
    """dp_cfm.h with the last chunk's NUL lanes not computed (forms 7 / 8: the narrowest class of computed lanes that holds the window's tail,
    and one class wider): windows whose best path ENDS in the padding - the needle's head matched at the very end of the window (real lanes of
    the last chunk, or the adjacent half-chunk when the tail is short) and its remaining rows unmatched, so that the maximum of the last row
    sits in a padding lane reached by a gap step or the diagonal and decays down the padding column."""
    rng = random.Random(7700 + swl)
    checked = in_padding = 0
    for it in range(1500):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019"])
        n = rng.randint(2, 10)
        needle = _rnd(rng, n, alpha)
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.6:
            while True:
                sc = [rng.randint(0, 40), rng.randint(0, 20), rng.randint(0, 20), rng.randint(0, 6), rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12),
                      rng.randint(0, 20), rng.randint(0, 12)]
                if 2 * sc[3] <= sc[1]:
                    break
        nch = rng.choice([2, 2, 2, 3, 5])
        tail = rng.randint(1, swl)
        m = (nch - 1) * swl + tail
        filler = bytes([rng.choice(b"xyz.")])
        hay = bytearray(filler * m)
        # the needle's first k bytes planted just before the end of the window, the rest never occurs (filler is outside every alphabet)
        k = rng.randint(1, n)
        gap = rng.choice([0, 0, 1, 2, 3, 5])
        pos = m - k - gap - rng.choice([0, 0, 0, 1, 2, swl // 4, swl // 2])
        if pos < 0:
            continue
        for j in range(k):
            q = pos + j + (gap if j == k - 1 else 0)
            if q < m:
                hay[q] = needle[j]
        if rng.random() < 0.3:  # and some noise elsewhere
            for _ in range(rng.randint(1, 6)):
                hay[rng.randrange(m)] = rng.choice(needle)
        hay = bytes(hay)
        ip = rng.random() < 0.5
        u8 = _fits(n, sc)
        want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, lanes=swl, is_u8=u8)
        for form in (6, 7, 8, 17):
            got = K.dp_multi(needle, hay, sc, cs, ip, swl, form, u8)
            assert got == want, (needle, hay, sc, cs, ip, swl, form, got, want)
        checked += 1
        in_padding += want > 0
    assert checked > 1200 and in_padding > 300


def test_multi_chunk_closed_form_padding_where_gap_steps_into_the_padding_decide():
    """The gap-step entries into the last chunk's padding (dp_cf.h, 3b; dp_cfm.h adds the adjacent half-chunk's lanes as sources) are the
    maximum only when every lane a cheaper route could walk down is charged again and again: a free gap extension, an expensive gap opening,
    a long needle over a one- or two-letter alphabet and a tail of exactly a class's computed lanes.  A build without the real-lane entries
    differs on about 1 in 15 000 of these (none of the other tests' inputs), so this sweep is large and runs form 7 / 8 against form 6 (which
    the tests above hold to the oracle), with the oracle itself on a sample."""
    rng = random.Random(2)
    checked = 0
    for it in range(60000):
        swl = rng.choice([8, 16, 16, 32, 64])
        n = rng.randint(3, 14)
        alpha = rng.choice([b"a", b"ab", b"ab", b"abc", b"aA"])
        needle = bytes(rng.choice(alpha) for _ in range(n))
        e = rng.choice([0, 0, 0, 1, 1, 2])
        sc = [rng.randint(4, 40), rng.randint(2 * e, 20), rng.randint(max(2, e), 24), e, rng.randint(0, 30), rng.randint(0, 12), rng.randint(0, 12), rng.randint(0, 20),
              rng.randint(0, 12)]
        q = swl // 4
        m = (rng.choice([2, 2, 3]) - 1) * swl + rng.choice([q, q, 2 * q, 3 * q])
        hay = bytearray(b"x" * m)
        for j in range(m - rng.randint(1, min(m, swl)), m):
            hay[j] = rng.choice(alpha)
        hay = bytes(hay)
        cs = rng.random() < 0.3
        want = K.dp_multi(needle, hay, sc, cs, True, swl, 6, False)
        if it % 40 == 0:
            assert want == O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=True, lanes=swl, is_u8=False)
        for form in (7, 8):
            got = K.dp_multi(needle, hay, sc, cs, True, swl, form, False)
            assert got == want, (needle, hay, sc, cs, swl, form, got, want)
        checked += 1
    assert checked == 60000


def test_multi_chunk_biased_form_at_the_largest_values_it_is_configured_for():
    # cfm_ok's bound: 63 rows, the maximal default-scoring values, 1024-byte window of matches
    needle = b"a" * 63
    hay = b"a" * 1024
    for swl in (64, 32):
        u8 = _fits(63, DEF)
        want = O.sw_score(needle, hay, scoring=DEF, case_sensitive=False, include_prefix=True, lanes=swl, is_u8=u8)
        assert K.dp_multi(needle, hay, DEF, False, True, swl, 6, u8) == want
        assert K.dp_multi(needle, hay, DEF, False, True, swl, 5, u8) == want


def _rnd_utf8(rng, nbytes, alphabet):
    """a valid UTF-8 string of at most nbytes bytes from `alphabet` (a list of 1-3-byte characters)"""
    out = b""
    while True:
        c = rng.choice(alphabet).encode()
        if len(out) + len(c) > nbytes:
            return out
        out += c


@pytest.mark.parametrize("swl", [64, 32, 16])
def test_unicode_scorer_matches_the_oracle(swl):
    """dp_unicode.h's single-chunk scorer (rows = needle scalars, lanes = haystack bytes, continuation bytes as free transport lanes, the
    last row unpropagated, only the dwords that can hold a haystack byte computed) against score_haystack_unicode
    (src/smith_waterman/algo/unicode.rs:10-273) with propagate_horizontal_unicode_gaps (unicode_gap.rs:110-236)"""
    rng = random.Random(900 + swl)
    alphabets = [list("abcAB_ -"), list("abéÉñÑüÜß/_ "), list("aب人äÄé_. 語"), list("إنماÉé_-ab"), list("a😀é人_b")]
    checked = checked_t = 0
    for it in range(1200):
        alpha = rng.choice(alphabets)
        n = rng.randint(1, 6)
        needle = "".join(rng.choice(alpha) for _ in range(n))
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.4:
            sc = [rng.randint(0, 30), rng.randint(0, 16), rng.randint(0, 16), rng.randint(0, 5), rng.randint(0, 20), rng.randint(0, 10), rng.randint(0, 10),
                  rng.randint(0, 16), rng.randint(0, 10)]
        nbytes = rng.randint(1, swl) if it % 3 else rng.randint(1, swl // 2)
        hay = _rnd_utf8(rng, nbytes, alpha)
        if rng.random() < 0.6:  # plant the needle's characters in order
            chars = hay.decode()
            if len(chars) >= n:
                pos = sorted(rng.sample(range(len(chars)), n))
                lst = list(chars)
                for q, c in zip(pos, needle):
                    lst[q] = c if rng.random() < 0.7 else c.swapcase() if len(c.swapcase().encode()) == len(c.encode()) else c
                cand = "".join(lst).encode()
                if len(cand) <= swl:
                    hay = cand
        if not hay:
            continue
        if it % 5 == 4:  # not UTF-8: runs of continuation bytes (the scorers work on bytes; the gap scan's UTF-8 shortcut must not be taken)
            hb_ = bytearray(hay)
            for _ in range(rng.randint(1, 3)):
                q = rng.randrange(len(hb_))
                for t in range(q, min(len(hb_), q + rng.randint(4, 9))):
                    hb_[t] = rng.choice((0x80, 0x9F, 0xBF, 0xA9))
            hay = bytes(hb_)
        ip = rng.random() < 0.5
        rows = O.case_needle_unicode(needle, cs)
        u8 = _fits(len(rows), sc)
        want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, unicode=True, lanes=swl, is_u8=u8)
        reals = [swl // 2] + ([swl // 4] if len(hay) <= swl // 2 else [])
        for real in reals:
            got = K.dp_unicode(rows, hay, sc, ip, swl, real)
            assert got == want, (needle, hay, sc, cs, ip, swl, real, got, want)
            checked += 1
            if 2 * sc[3] <= sc[1]:  # the biased-throughout form's precondition (LaunchCfg::cfu_ok); small scorings fit 16 bits anyway
                for form in (1, 2):  # 1 = with the UTF-8 shortcut where the window allows it (as the kernel chooses), 2 = general steps
                    got_t = K.dp_unicode(rows, hay, sc, ip, swl, real, form=form)
                    assert got_t == want, ("T form", form, needle, hay, sc, cs, ip, swl, real, got_t, want)
                checked_t += 1
    assert checked > 700 and checked_t > 400, (checked, checked_t)


@pytest.mark.parametrize("swl", [64, 32, 16, 8])
def test_unicode_multi_chunk_scorer_matches_the_oracle(swl):
    """dp_unicode_multi_chunk (dp_unicode.h; windows of swl < m <= 1024 bytes, thread per haystack, the previous chunk's top half parked
    per needle row) against score_haystack_unicode over its chunks (src/smith_waterman/algo/unicode.rs:10-217, unicode_gap.rs:110-236):
    UTF-8 text of one- to four-byte scalars, runs of continuation bytes (not UTF-8: the scorers work on bytes), windows that end
    anywhere in their last chunk, both score classes' parking formats, random scorings."""
    rng = random.Random(4100 + swl)
    alphabets = [list("abcAB_ -"), list("abéÉñÑüÜß/_ "), list("aب人äÄé_. 語"), list("إنماÉé_-ab "), list("a😀é人_b")]
    checked = checked_t = 0
    for it in range(700):
        alpha = rng.choice(alphabets)
        n = rng.randint(1, 6)
        needle = "".join(rng.choice(alpha) for _ in range(n))
        cs = rng.random() < 0.3
        sc = DEF
        if rng.random() < 0.4:
            sc = [rng.randint(0, 30), rng.randint(0, 16), rng.randint(0, 16), rng.randint(0, 5), rng.randint(0, 20), rng.randint(0, 10), rng.randint(0, 10),
                  rng.randint(0, 16), rng.randint(0, 10)]
        hi = rng.choice([2 * swl, 3 * swl, min(1024, 6 * swl), min(1024, 17 * swl)])
        nbytes = rng.randint(swl + 1, max(swl + 1, hi))
        hay = _rnd_utf8(rng, nbytes, alpha)
        if rng.random() < 0.7:
            chars = hay.decode()
            if len(chars) >= n:
                lst = list(chars)
                for q, c in zip(sorted(rng.sample(range(len(chars)), n)), needle):
                    lst[q] = c if rng.random() < 0.7 else c.swapcase() if len(c.swapcase().encode()) == len(c.encode()) else c
                cand = "".join(lst).encode()
                if swl < len(cand) <= 1024:
                    hay = cand
        if it % 5 == 4:
            hb_ = bytearray(hay)
            for _ in range(rng.randint(1, 4)):
                q = rng.randrange(len(hb_))
                for t in range(q, min(len(hb_), q + rng.randint(4, 12))):
                    hb_[t] = rng.choice((0x80, 0x9F, 0xBF, 0xA9))
            hay = bytes(hb_)
        if len(hay) <= swl:
            continue
        ip = rng.random() < 0.5
        rows = O.case_needle_unicode(needle, cs)
        for u8 in {_fits(len(rows), sc), False}:  # the class the matcher would choose, and the u16 class (unpacked parking) anyway
            want = O.sw_score(needle, hay, scoring=sc, case_sensitive=cs, include_prefix=ip, unicode=True, lanes=swl, is_u8=u8)
            got = K.dp_unicode_multi(rows, hay, sc, ip, swl, is_u8=u8)
            assert got == want, (needle, hay, sc, cs, ip, swl, u8, got, want)
            checked += 1
            if 2 * sc[3] <= sc[1]:  # the biased-throughout form's precondition (LaunchCfg::cfu_ok)
                for form in (1, 2):
                    got_t = K.dp_unicode_multi(rows, hay, sc, ip, swl, is_u8=u8, form=form)
                    assert got_t == want, ("T form", form, needle, hay, sc, cs, ip, swl, u8, got_t, want)
                checked_t += 1
    assert checked > 800 and checked_t > 500, (checked, checked_t)


def test_ascii_window_of_any_length_equals_the_reference_prefilter_window():
    """window_first_last (dp_body.h: 32-byte blocks, one compare per case-folded needle byte, positions masked behind the haystack's end)
    against the window the reference's ASCII prefilter returns for an accepted haystack (src/prefilter/algo/ascii.rs:6-72), haystacks of
    33 - 700 bytes with the next haystack's bytes behind them"""
    rng = random.Random(515)
    accepted = 0
    for it in range(4000):
        alpha = rng.choice([b"ab", b"abcA_", b"abcdefABCDEF_-/ 019", bytes(range(33, 127))])
        n = rng.randint(1, 8)
        needle = _rnd(rng, n, alpha)
        cs = rng.random() < 0.3
        L = rng.choice([rng.randint(33, 70), rng.randint(33, 160), rng.randint(120, 700)])
        hay = bytearray(_rnd(rng, L, alpha + b"xyzw" * 3))
        if rng.random() < 0.7:
            for q, c in zip(sorted(rng.sample(range(L), n)), needle):
                hay[q] = c
        hay = bytes(hay)
        ok, ws, we = O.prefilter(needle, hay, 0, cs, False, 64)
        if not ok:
            continue
        accepted += 1
        got, _ = K.window(needle, hay, cs)
        assert got == (ws, we), (needle, hay, cs, got, (ws, we))
    assert accepted > 1500, accepted


@pytest.mark.parametrize("swl", [64, 32, 16])
def test_unicode_register_path_of_the_short_corpus_kernel(swl):
    """k2u_dp_unicode_half's register path (haystacks of at most swl/2 <= 32 bytes, two 16-byte vectors): the SWAR window search
    (unicode_window_regs) against the reference prefilter's window and against the byte-wise search, the window's bytes shifted out of the
    vectors (whatever follows the haystack in them - the next haystack - must not leak in), the set-up that also decides the UTF-8 shortcut,
    and the rows - against score_haystack_unicode over the trimmed window"""
    rng = random.Random(4400 + swl)
    alphabets = [list("abcAB_ -"), list("abéÉñÑüÜß/_ "), list("aب人äÄé_. 語"), list("إنماÉé_-ab"), list("a😀é人_b")]
    accepted = 0
    for it in range(4000):
        alpha = rng.choice(alphabets)
        n = rng.randint(1, 5)
        needle = "".join(rng.choice(alpha) for _ in range(n))
        cs = rng.random() < 0.3
        hay = _rnd_utf8(rng, rng.randint(1, swl // 2), alpha)
        chars = hay.decode()
        if len(chars) >= n and rng.random() < 0.8:
            pos = sorted(rng.sample(range(len(chars)), n))
            lst = list(chars)
            for q, c in zip(pos, needle):
                lst[q] = c if rng.random() < 0.7 else c.swapcase() if len(c.swapcase().encode()) == len(c.encode()) else c
            cand = "".join(lst).encode()
            if 1 <= len(cand) <= swl // 2:
                hay = cand
        if not hay:
            continue
        ok, ws, we = O.prefilter(needle, hay, max_typos=0, case_sensitive=cs, unicode=True, lanes=swl)
        if not ok:
            continue
        sc = DEF
        if rng.random() < 0.4:
            while True:
                sc = [rng.randint(0, 30), rng.randint(0, 16), rng.randint(0, 16), rng.randint(0, 5), rng.randint(0, 20), rng.randint(0, 10), rng.randint(0, 10),
                      rng.randint(0, 16), rng.randint(0, 10)]
                if 2 * sc[3] <= sc[1]:  # LaunchCfg::cfu_ok
                    break
        rows = O.case_needle_unicode(needle, cs)
        w_regs, w_mem, s_regs, s_mem = K.unicode_regs(rows, hay, sc, swl)
        assert w_regs == (ws, we) and w_mem == (ws, we), (needle, hay, cs, w_regs, w_mem, (ws, we))
        sp = ws - 1 if ws else 0
        want = O.sw_score(needle, hay[sp:we], scoring=sc, case_sensitive=cs, include_prefix=(sp == 0), unicode=True, lanes=swl, is_u8=_fits(len(rows), sc))
        assert s_regs == want and s_mem == want, (needle, hay, sc, cs, swl, s_regs, s_mem, want)
        accepted += 1
    assert accepted > 1200, accepted


def test_unicode_window_equals_the_reference_prefilter_window():
    """unicode_window_first_last (dp_unicode.h) against the window the reference's unicode prefilter returns for an accepted haystack
    (src/prefilter/algo/unicode.rs:118-219), at the three lane widths (the window does not depend on the width)"""
    rng = random.Random(77)
    alphabets = [list("abcAB_ -"), list("abéÉñÑüÜß/_ "), list("aب人äÄé_. 語"), list("إنماÉé_-ab")]
    accepted = 0
    for it in range(3000):
        alpha = rng.choice(alphabets)
        n = rng.randint(1, 5)
        needle = "".join(rng.choice(alpha) for _ in range(n))
        cs = rng.random() < 0.3
        hay = _rnd_utf8(rng, rng.randint(1, 120), alpha)
        chars = hay.decode()
        if len(chars) >= n and rng.random() < 0.7:
            pos = sorted(rng.sample(range(len(chars)), n))
            lst = list(chars)
            for q, c in zip(pos, needle):
                lst[q] = c
            hay = "".join(lst).encode()
        ok, ws, we = O.prefilter(needle, hay, max_typos=0, case_sensitive=cs, unicode=True, lanes=64)
        for lanes in (32, 16):
            assert O.prefilter(needle, hay, max_typos=0, case_sensitive=cs, unicode=True, lanes=lanes) == (ok, ws, we)
        if not ok:
            continue
        accepted += 1
        assert K.unicode_window(O.case_needle_unicode(needle, cs), hay) == (ws, we), (needle, hay, cs)
    assert accepted > 1000
