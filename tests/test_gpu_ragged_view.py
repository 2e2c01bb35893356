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


def _padded16(hs):
    """list[bytes] -> (uint8 array in the library's device layout, u32 exclusive ends inside it)"""
    buf = bytearray()
    ends = []
    for h in hs:
        buf += b"\0" * (-len(buf) % 16)
        buf += h
        ends.append(len(buf))
    buf += b"\0" * (-len(buf) % 16) + b"\0" * 96
    return np.frombuffer(bytes(buf), np.uint8).copy(), np.array(ends, np.uint32)


@pytest.mark.parametrize("max_len", [100, 200])
def test_borrowed_corpus_gets_the_view_and_ignores_a_wrong_length_hint(max_len):
    # fzb_corpus_build_view on borrowed device memory (what bench.py / a torch caller holds): the same kernel as on an uploaded list.  The
    # view's vector count comes from the lengths the builder READ - a corpus whose max_len was cleared (fzb_corpus_set_max_len(c, 0) =
    # "unknown") still scans every vector of its 129..256-byte haystacks (round 3 picked 8 vectors for max_len == 0 and dropped matches).
    import torch
    rng = random.Random(71 + max_len)
    hs = [h.encode() for h in _list(rng, 30000, max_len, "deadbeef", "deabfxyz_-/ 01")]
    hs[123] = b"x" * (max_len - 8) + b"deadbeef"  # a match that ends in the longest haystack's last vector
    data, ends = _padded16(hs)
    dev = torch.device("cuda", 0)
    d_bytes, d_ends = torch.from_numpy(data).to(dev), torch.from_numpy(ends.astype(np.int64)).to(dev).to(torch.int32)
    want = O.Matcher("deadbeef", lanes=(64, 64, 32)).match_list(hs)
    assert 123 in want["index"].tolist()
    for hint in (0, max_len, 4096):
        cp = F.Corpus.from_device(d_bytes.data_ptr(), d_ends.data_ptr(), len(hs), d_bytes.numel(), keep=(d_bytes, d_ends), max_len=hint)
        m = F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64))
        before = m.match_list(cp)
        assert cp.build_view() is True
        if hint:
            assert F.lib().fzb_corpus_set_max_len(cp.h, 0) == 0  # "unknown" again: must not change what the view kernel scans
        after = m.match_list(cp)
        assert before.tolist() == want.tolist() and after.tolist() == want.tolist(), hint
    up = F.Corpus(hs)  # an uploaded corpus measured its lengths: a smaller bound is refused, a looser one ignored
    assert F.lib().fzb_corpus_set_max_len(up.h, 32) == 1 and F.lib().fzb_corpus_set_max_len(up.h, 0) == 0 and F.lib().fzb_corpus_set_max_len(up.h, 9999) == 0
    assert F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64)).match_list(up).tolist() == want.tolist()
    short = F.Corpus.from_device(d_bytes.data_ptr(), d_ends.data_ptr(), 5, d_bytes.numel(), keep=(d_bytes, d_ends))
    assert isinstance(short.build_view(), bool)


def _plant(rng, L, needle, alphabet, where=None):
    s = [rng.choice(alphabet) for _ in range(L)]
    lo, hi = where if where else (0, L)
    for q, c in zip(sorted(rng.sample(range(lo, hi), len(needle))), needle):
        s[q] = c
    return "".join(s)


@pytest.mark.parametrize("n_out", [1, 40])
def test_a_few_haystacks_beyond_256_bytes_do_not_cost_the_list_its_view(n_out):
    """The view holds haystacks up to 256 bytes; a list with a FEW longer ones (up to n/256 + 64) keeps it - they are listed as outliers
    and decided by k1_cdfa_outliers over the canonical layout.  Outliers at tile and group boundaries, matching only beyond byte 256,
    matching nowhere, beyond 1024 bytes (the generic scorer's territory), under every automaton; sub-range queries see only their own."""
    rng = random.Random(4242 + n_out)
    alphabet = "abcdefghijklmnDEAB_-/. 01xyz"
    for needle, cfg in NEEDLES + [("é人", dict())]:
        abc = alphabet.replace(needle[0].lower(), "").replace(needle[0].upper(), "")  # filler that cannot complete the needle by chance
        n = 1024 * 5 + 333
        hs = _list(rng, n, 120, needle, alphabet)
        spots = [0, 1023, 1024, 2047 + 64, n - 1][:n_out] + rng.sample(range(n), max(0, n_out - 5))
        for k, i in enumerate(spots):
            L = rng.choice([257, 300, 1000, 1025, 3000])
            kind = k % 3
            if kind == 0:
                hs[i] = _plant(rng, L, needle, abc, where=(257, L)) if L - 257 >= len(needle) else _plant(rng, L, needle, abc)
            elif kind == 1:
                hs[i] = "".join(rng.choice(abc) for _ in range(L))  # cannot match
            else:
                hs[i] = _plant(rng, L, needle, alphabet)
        assert F.Corpus(hs).build_view(), "the list lost its view to a few outliers"
        got, want, _ = both(needle, hs, **cfg)
        assert len(want) > 0
        assert_same(got, want, f"{needle!r} {cfg} outliers={n_out}")
        outl = set(spots)
        assert any(int(i) in outl for i in want["index"]) or n_out == 1
        if not cfg:  # tile-aligned sub-ranges of the same corpus (the view's precondition): each sees only its own outliers
            cp = F.Corpus(hs)
            fm = F.Matcher(needle, F.Config(sort=F.SortStrategy.IndexAsc))
            whole = fm.match_list(cp)
            for first, cnt in ((0, 1024), (1024, 2048), (2048, n - 2048)):
                part = fm.match_list_into(cp, first=first, count=cnt, index_offset=first)
                sl = whole[(whole["index"] >= first) & (whole["index"] < first + cnt)]
                assert sorted(part.tolist()) == sorted(sl.tolist()), (needle, first)


def test_a_list_of_mostly_long_haystacks_gets_no_view():
    rng = random.Random(7)
    hs = ["".join(rng.choice("abcdef") for _ in range(rng.randint(200, 400))) for _ in range(3000)]
    assert not F.Corpus(hs).build_view()
    got, want, _ = both("fade", hs)
    assert_same(got, want, "long list")


def test_a_list_of_more_tiles_than_the_short_list_form_takes():
    """below two tiles per CU (512 on MI355X) the view filter runs as 1024-thread workgroups, one 64-haystack group per wave - every other list in
    this file; this one has 540 000 haystacks of 8..128 bytes (528 tiles): the 256-thread form (four groups in a row per wave, what BASELINE
    config 4 runs), and the thread-per-window scorers above 49 152 multi-chunk windows are not reached but the class bodies above 32 768 are"""
    rng = np.random.default_rng(4242)
    n = 540_000
    lens = rng.integers(8, 129, n)
    ends = np.cumsum(lens).astype(np.uint64)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz_-/. 0123456789", dtype=np.uint8)
    data = alpha[rng.integers(0, len(alpha), int(ends[-1]))].copy()
    needle = b"deadbeef"
    starts = np.concatenate([[0], ends[:-1]]).astype(np.int64)
    for i in rng.choice(n, n // 12, replace=False):  # the needle as an ordered subsequence of about one haystack in twelve
        L = int(lens[i])
        pos = np.sort(rng.choice(L, len(needle), replace=False))
        data[starts[i] + pos] = np.frombuffer(needle, dtype=np.uint8)
    got, want, fm = both("deadbeef", None, pf=64, packed=(data, ends))
    assert len(want) > 40_000
    assert fm.last_counters()["multi_chunk_scored"] > 10_000
    assert_same(got, want, "540 k ragged haystacks")


@pytest.mark.parametrize("n", [40_000, 700_000])
def test_survivors_that_cluster_in_a_few_tiles(n):
    """Every match of the list inside ONE stretch of haystacks (all files below one directory): on a small list the compaction launch classifies its
    workgroup's survivors itself (k_compact1_classify) - here whole tiles of them, several rounds per workgroup, windows of every class; a list beyond two
    tiles per CU takes k_compact1 + k2w_classify, which spread any survivor list evenly.  Both against the oracle, with and without the fused form."""
    import os
    rng = random.Random(n)
    alphabet = "ghijkxyz_/."
    hs = []
    lo, hi = n // 3, n // 3 + 5000  # ~ five tiles of survivors
    for i in range(n):
        L = rng.randint(8, 140)
        s = [rng.choice(alphabet) for _ in range(L)]
        if lo <= i < hi and rng.random() < 0.95:
            for q, c in zip(sorted(rng.sample(range(L), 8)), "deadbeef"):
                s[q] = c
        hs.append("".join(s))
    cp = F.Corpus(hs)
    want = O.Matcher("deadbeef", lanes=(64, 64, 32)).match_list(hs)
    assert 4000 < len(want) < 5000
    for fused in (True, False):
        if not fused:
            os.environ["FZB_NO_FUSED_CLASSIFY"] = "1"
        F.lib().fzb_debug_reload_knobs()
        try:
            got = F.Matcher("deadbeef", F.Config(pf_lanes=64, sw_lanes=64)).match_list(cp)
        finally:
            os.environ.pop("FZB_NO_FUSED_CLASSIFY", None)
            F.lib().fzb_debug_reload_knobs()
        assert got.tolist() == want.tolist(), (n, fused)
