"""Bulk differential fuzz of the scorers AS COMPILED FOR gfx950: >= 2 M random (needle, haystack, scoring) triples per lane width through
`fzb_match_list_into` against the oracle.  The host harness (tests/kernel_host) fuzzes the same headers compiled for x86; this is the ISA
hipcc emits.  The configurations sit on BOTH sides of every kernel-selection predicate of fzb_matcher_create / run_pipeline:
  cf_ok   (dp_cf.h: no NUL in the needle, biased values fit 16 bits, 2 * gap_extend <= mismatch)  vs  dp_body.h's first form
  bias_ok (biased gap scan)  vs  the literal three-operation form
  cfm_ok  (dp_cfm.h multi-chunk scorer)  vs  its first form
  u8 / u16 score class, short corpus (k2b_dp_short) / classified launches / multi-chunk / unicode half / full
  typo fast path (single prefilter chunk, marginal survivors re-decided)  vs  the lane-exact window pass
Index order (no sort in the way), every record compared."""
import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

pytestmark = pytest.mark.gpu

LANE_TRIPLES = {64: (64, 64, 32), 32: (32, 32, 16), 16: (16, 16, 8)}
DEFAULT = [12, 6, 5, 1, 12, 4, 4, 8, 4]
# (scoring, what it is for): match, mismatch, gap_open, gap_extend, prefix, capitalization, matching_case, exact, delimiter
SCORINGS = [
    (DEFAULT, "default: cf_ok, cfm_ok, u8 class for short needles"),
    ([12, 6, 5, 3, 12, 4, 4, 8, 4], "2 * gap_extend = mismatch: the cf_ok boundary, inside"),
    ([12, 6, 5, 4, 12, 4, 4, 8, 4], "2 * gap_extend > mismatch: dp_body.h first form (biased), dp_multi first form"),
    ([12, 900, 450, 400, 12, 4, 4, 8, 4], "130 * gex fits, 200 * gex does not: cf_ok but not cfm_ok"),
    ([12, 1100, 600, 500, 12, 4, 4, 8, 4], "130 * gex does not fit 16 bits: bias_ok false, literal gap scan"),
    ([1, 1, 1, 0, 0, 0, 0, 0, 0], "minimal: zero gap_extend, u8 class for every needle"),
    ([40, 13, 17, 2, 30, 11, 9, 21, 10], "heavy bonuses: u16 class already for short needles"),
    ([12, 6, 0, 0, 12, 4, 4, 8, 4], "free gaps"),
    ([0, 0, 0, 0, 0, 0, 0, 0, 0], "all zero"),
    ([7, 3, 9, 1, 0, 13, 0, 5, 2], "gap_open >> bonuses, no prefix / case bonus"),
]
NEEDLES = [b"deadbe", b"ab", b"a", b"fBr", b"deadbeefcafe", b"abcdefghijklmn", b"x_y", b"Aa0", b"de\0d", b"abcdefghijklmnopqrstuvwxyz012345"]


def make_list(rng, needle, n, len_pool):
    """n haystacks over the needle's letters (both cases) + a few fillers; half of them carry a random in-order subset of the needle
    (30 % of those all of it) at sorted random positions - a position drawn twice simply carries the later needle byte"""
    nn = len(needle)
    letters = sorted(set(needle) | {c ^ 0x20 for c in needle if chr(c).isalpha()})
    alpha = np.array(letters + list(b"_-/ z0Q") + ([0] if 0 in needle else []), np.uint8)
    L = rng.choice(len_pool, n).astype(np.int64)
    W = max(int(L.max()) if n else 1, 1)
    ar = np.arange(W)
    rows = alpha[rng.integers(0, len(alpha), (n, W))]
    keep = rng.random((n, nn)) > 0.15
    keep[rng.random(n) < 0.3] = True
    keep &= (rng.random(n) < 0.5)[:, None] & (L > 0)[:, None]
    pos = np.minimum((np.sort(rng.random((n, nn)), axis=1) * L[:, None]).astype(np.int64), np.maximum(L - 1, 0)[:, None])
    ri, ci = np.nonzero(keep)
    rows[ri, pos[ri, ci]] = np.frombuffer(needle, np.uint8)[ci]
    flip = (rng.random((n, W)) < 0.07) & (((rows | 0x20) >= ord("a")) & ((rows | 0x20) <= ord("z")))
    rows = np.where(flip, rows ^ 0x20, rows).astype(np.uint8)
    mask = ar[None, :] < L[:, None]
    data = np.concatenate([rows[mask], np.zeros(64, np.uint8)])
    ends = np.cumsum(L).astype(np.uint64)
    return data, ends


def check(needle, data, ends, lanes, tag, **cfg):
    pf, sw8, sw16 = LANE_TRIPLES[lanes]
    om = O.Matcher(needle, lanes=(pf, sw8, sw16), sort="IndexAsc", **cfg)
    fc = F.Config(max_typos=cfg.get("max_typos", 0), scoring=F.Scoring(*cfg.get("scoring", DEFAULT)), pf_lanes=pf,
                  unicode=F.UnicodeMatching[cfg.get("unicode", "Smart")], casing=F.CaseMatching[cfg.get("casing", "Smart")])
    fm = F.Matcher(needle, fc)
    assert fm.info()["use_u8"] == om.info()["use_u8"] and fm.info()["sw_lanes"] == om.info()["sw_lanes"], tag
    want = om.match_packed(data, ends)
    got = fm.match_list_into(F.Corpus(packed=(data, ends)))
    if got.tolist() != want.tolist():
        bad = next((i for i in range(min(len(got), len(want))) if got[i].tolist() != want[i].tolist()), min(len(got), len(want)))
        idx = int(want[bad]["index"]) if bad < len(want) else int(got[bad]["index"])
        lo = int(ends[idx - 1]) if idx else 0
        raise AssertionError((tag, "records", len(got), len(want), "first difference at", bad, got[bad : bad + 1].tolist(), want[bad : bad + 1].tolist(),
                              "haystack", bytes(data[lo : int(ends[idx])])))
    return len(ends)


@pytest.mark.parametrize("lanes", [64, 32, 16])
def test_two_million_random_triples_per_lane_width_ascii(lanes):
    rng = np.random.default_rng(1000 + lanes)
    pools = {
        "short": np.array([0, 1, 3, 6, 8, 12, 16, 20, 27, 31, 32]),                      # every haystack fits half a chunk: k2b_dp_short, typo fast path
        "chunk": np.array([0, 5, 16, 31, 32, 33, 40, 47, 48, 49, 63, 64]),                # single-chunk windows of three classes
        "ragged": np.array([2, 9, 30, 33, 64, 65, 70, 100, 127, 128, 129, 200, 300]),     # + multi-chunk windows (dp_multi / dp_cfm)
    }
    total = 0
    n = 26_000
    for si, (sc, what) in enumerate(SCORINGS):
        for ni, needle in enumerate(NEEDLES):
            if (si + ni) % 2 and si > 1:  # the full cross product is 3x the budget: default + boundary scorings get every needle
                continue
            for typos in (0, 1, 2, None):
                if typos in (1, 2) and len(needle) <= typos:
                    continue
                pool = ("short", "chunk", "ragged")[(si + ni + (typos or 0)) % 3]
                data, ends = make_list(rng, needle, n if typos is not None else n // 2, pools[pool])
                casing = "Respect" if (si + ni) % 5 == 4 else "Smart"
                total += check(needle, data, ends, lanes, (lanes, what, needle, typos, pool, casing), max_typos=typos, scoring=sc, casing=casing)
    assert total >= 2_000_000, total


def make_unicode_list(rng, needle, n, max_chars):
    chars = sorted(set(needle) | set(needle.upper()) | set(needle.lower()) | set("éЖ中_ a😀"))
    enc = [c.encode() for c in chars]
    table = np.zeros((len(chars), 4), np.uint8)
    lens = np.array([len(e) for e in enc])
    for i, e in enumerate(enc):
        table[i, : len(e)] = np.frombuffer(e, np.uint8)
    C = rng.integers(0, max_chars + 1, n)
    W = int(C.max()) if n else 1
    idx = rng.integers(0, len(chars), (n, W))
    # carry the needle's scalars in order in half of the rows
    nidx = np.array([chars.index(c) for c in needle])
    nn = len(nidx)
    keep = rng.random((n, nn)) > 0.2
    k = np.minimum(keep.sum(1), C)
    order = np.argsort(~keep, axis=1, kind="stable")
    r = rng.random((n, W))
    ar = np.arange(W)
    r[ar[None, :] >= C[:, None]] = 2.0
    pos = r.argsort(1).argsort(1) < k[:, None]
    slot = np.clip(np.cumsum(pos, 1) - 1, 0, nn - 1)
    emb = np.take_along_axis(nidx[order], slot, 1)
    carry = rng.random(n) < 0.5
    idx = np.where(pos & carry[:, None], emb, idx)
    cmask = ar[None, :] < C[:, None]
    flat = idx[cmask]
    b = table[flat]
    bmask = np.arange(4)[None, :] < lens[flat][:, None]
    data = np.concatenate([b[bmask], np.zeros(64, np.uint8)])
    row_bytes = np.zeros(n, np.int64)
    np.add.at(row_bytes, np.nonzero(cmask)[0], lens[flat])
    return data, np.cumsum(row_bytes).astype(np.uint64)


@pytest.mark.parametrize("lanes", [64, 32, 16])
def test_random_triples_unicode_scorer(lanes):
    rng = np.random.default_rng(2000 + lanes)
    total = 0
    for sc, what in (SCORINGS[0], SCORINGS[1], SCORINGS[2], SCORINGS[4], SCORINGS[5], SCORINGS[6]):
        for needle in ("إنما", "éa", "中文字", "aЖ", "😀é", "é"):
            for typos, max_chars in ((0, 14), (0, 60), (1, 14), (None, 30)):
                if typos and len(needle) <= typos:
                    continue
                data, ends = make_unicode_list(rng, needle, 9000, max_chars)
                total += check(needle, data, ends, lanes, (lanes, what, needle, typos, max_chars), max_typos=typos, scoring=sc)
    assert total >= 1_000_000, total


@pytest.fixture
def forced_unicode_multi():
    """The thread-per-haystack multi-chunk unicode scorer, which the lists below are too small to reach by default (FZB_UNICODE_MULTI=1: shipped
    from 32 768 queued windows on)."""
    import os
    os.environ["FZB_UNICODE_MULTI"] = "1"
    F.lib().fzb_debug_reload_knobs()
    yield
    os.environ.pop("FZB_UNICODE_MULTI", None)
    F.lib().fzb_debug_reload_knobs()


@pytest.mark.parametrize("lanes", [64, 32, 16])
def test_random_triples_through_the_view_filter_and_the_unicode_multi_chunk_scorer(lanes, forced_unicode_multi):
    rng = np.random.default_rng(3000 + lanes)
    total = 0
    # lists whose longest haystack is 33..256 bytes get the interleaved view (every second haystack carries the needle)
    view_pool = np.array([0, 2, 9, 30, 33, 48, 64, 65, 70, 100, 127, 128, 129, 200, 255, 256])
    for si, (sc, what) in enumerate(SCORINGS):
        for ni, needle in enumerate(NEEDLES):
            if (si + ni) % 3:
                continue
            data, ends = make_list(rng, needle, 30_000, view_pool)
            casing = "Respect" if (si + ni) % 5 == 4 else "Smart"
            total += check(needle, data, ends, lanes, (lanes, "view filter", what, needle, casing), max_typos=0, scoring=sc, casing=casing)
    assert total >= 900_000, total
    utotal = 0
    for sc, what in (SCORINGS[0], SCORINGS[1], SCORINGS[2], SCORINGS[4], SCORINGS[5], SCORINGS[6]):
        for needle in ("إنما", "éa", "中文字", "aЖ"):
            for typos, max_chars in ((0, 120), (None, 90), (1, 200)):
                data, ends = make_unicode_list(rng, needle, 4000, max_chars)
                utotal += check(needle, data, ends, lanes, (lanes, "unicode multi-chunk", what, needle, typos, max_chars), max_typos=typos, scoring=sc)
    assert utotal >= 250_000, utotal
