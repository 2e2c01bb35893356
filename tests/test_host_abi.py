"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/frizbee_hip.h declares,
and the host logic (class selection, Smart casing/unicode, guards, needle tables, sort/merge helpers) agrees with the oracle.
No compute entry point is exercised here (no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import frizbee_amd as F
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "frizbee_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fzb_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(F.SYMBOLS), (declared, F.SYMBOLS)
    l = F.lib()
    for s in declared:
        assert hasattr(l, s), s


def test_struct_layouts_match_header():
    assert C.sizeof(F._CScoring) == 18 and C.sizeof(F._CConfig) == 4 * 4 + 18 + 4 + 2 + 4  # 2 bytes of padding before the trailing int32 `matching`
    assert F._CConfig.matching.offset == 40 and C.sizeof(F._CPattern) == 8 + 8 + 6 * 4 + 18 + 2 + 4
    assert F.MATCH_DTYPE.itemsize == 8 and O.MATCH_DTYPE == F.MATCH_DTYPE
    c = F._CConfig()
    F.lib().fzb_config_default(C.byref(c))
    assert (c.max_typos, c.casing, c.unicode, c.sort, c.matching) == (0, 1, 1, 0, 0)
    assert [getattr(c.scoring, f[0]) for f in F._CScoring._fields_] == O.DEFAULT_SCORING


@pytest.mark.parametrize("needle,cfg", [
    ("deadbe", {}), ("fBr", {}), ("a" * 13, {}), ("a" * 14, {}), ("إنما", {}), ("é", dict(unicode="Ignore")), ("abc", dict(unicode="Always")),
    ("FoO", dict(casing="Ignore")), ("foo", dict(casing="Respect")), ("Éa", {}), ("ßx", {}), ("x", dict(scoring=[12, 260, 5, 1, 12, 4, 4, 8, 4])),
    ("BBBB", dict(scoring=[40, 0, 0, 0, 0, 40, 0, 0, 0])), ("abcd", dict(scoring=[12, 6, 5, 8, 12, 4, 4, 8, 4])),
])
def test_matcher_new_resolution_matches_oracle(needle, cfg):
    kw = dict(cfg)
    fc = F.Config(casing=F.CaseMatching[kw.pop("casing", "Smart")], unicode=F.UnicodeMatching[kw.pop("unicode", "Smart")],
                  scoring=F.Scoring(*kw.pop("scoring", O.DEFAULT_SCORING)), pf_lanes=64, sw_lanes=0)
    # explicit lanes: ask for the AVX-512 pair of the needle's class
    u8 = O.score_fits_in_u8(len(needle.encode()), fc.scoring.as_list())
    fc.sw_lanes = 64 if u8 else 32
    info = F.Matcher(needle, fc).info()
    want = O.Matcher(needle, lanes=(64, 64, 32), **cfg).info()
    assert (info["pf_lanes"], info["sw_lanes"], info["use_u8"]) == (want["pf_lanes"], want["sw_lanes"], want["use_u8"])
    casing = cfg.get("casing", "Smart")
    assert info["case_sensitive"] == O.respects_case_for(casing, needle)
    uni = cfg.get("unicode", "Smart")
    assert info["unicode"] == (uni == "Always" or (uni == "Smart" and not needle.isascii()))
    assert info["rows"] == (len(needle) if info["unicode"] else len(needle.encode()))


def test_auto_lanes_follow_host_cpu():
    flags = open("/proc/cpuinfo").read()
    info8 = F.Matcher("deadbe").info()
    info16 = F.Matcher("a" * 20).info()
    has = lambda f: re.search(r"\b%s\b" % f, flags) is not None
    if has("avx512f") and has("avx512bw") and has("bmi1") and has("bmi2"):
        assert (info16["pf_lanes"], info16["sw_lanes"]) == (64, 32)
        if has("avx512vbmi"):
            assert (info8["pf_lanes"], info8["sw_lanes"]) == (64, 64)
    elif has("avx2"):
        assert (info8["pf_lanes"], info8["sw_lanes"], info16["sw_lanes"]) == (32, 32, 16)


def test_guards_raise_with_reference_panic_text():
    with pytest.raises(F.PanicError, match="needle too long and could overflow the u16 score"):
        F.Matcher("f", F.Config(scoring=F.Scoring(capitalization_bonus=60000, matching_case_bonus=40000)))
    with pytest.raises(F.FrizbeeError):
        F.Matcher(b"\xff\xfe")  # not UTF-8 (a Rust &str cannot hold this)
    # long needles are accepted up to the reference's own bound: `guard_against_score_overflow` (src/lib.rs:506-527) lets 3 639 rows
    # through with the default scoring ((65535 - 12 - 8 - 6 - 2) / 18; the documented `max_needle_len()` = 10 922 of :483-485 divides by
    # the per-char BONUS only and is never what panics), in the u16 class; one more row panics with the reference's text
    for n in (65, 200, 1000, 3639):
        info = F.Matcher("a" * n).info()
        assert info["rows"] == n and not info["use_u8"] and O.Matcher("a" * n).info()["use_u8"] is False
    assert O.max_needle_len() == 10922
    with pytest.raises(F.PanicError, match=r"needle too long and could overflow the u16 score: 3640 > 3639"):
        F.Matcher("a" * 3640)
    with pytest.raises(RuntimeError, match=r"needle too long and could overflow the u16 score: 3640 > 3639"):
        O.Matcher("a" * 3640)
    zero = F.Scoring(match_score=0, mismatch_penalty=0, gap_open_penalty=0, gap_extend_penalty=0, prefix_bonus=0, capitalization_bonus=0, matching_case_bonus=0, exact_match_bonus=0, delimiter_bonus=0)
    assert F.Matcher("a" * 20000, F.Config(scoring=zero)).info()["rows"] == 20000  # "a zero per-char score can never overflow regardless of needle length"
    assert F.Matcher("é" * 64, F.Config(unicode=F.UnicodeMatching.Always)).info()["rows"] == 64  # 128 bytes, 64 scalar rows: long as well
    # a scoring under which a 100-byte needle still fits the u8 class (src/smith_waterman/mod.rs:92-116)
    tiny = F.Scoring(match_score=1, mismatch_penalty=1, gap_open_penalty=1, gap_extend_penalty=0, prefix_bonus=0, capitalization_bonus=0, matching_case_bonus=0, exact_match_bonus=0, delimiter_bonus=0)
    assert F.Matcher("a" * 100, F.Config(scoring=tiny)).info()["use_u8"] is True
    # unicode rows are counted in chars for the guard (src/matcher/algo.rs:383-393)
    F.Matcher("一二三四五六七八", F.Config(scoring=F.Scoring(capitalization_bonus=4000)))


def test_radix_sort_and_k_merge_match_oracle():
    rng = np.random.default_rng(42)
    n = 1 << 16
    arr = np.zeros(n, F.MATCH_DTYPE)
    arr["index"] = np.arange(n)
    arr["score"] = rng.integers(0, 1 << 16, n)
    assert F.radix_sort_matches(arr).tolist() == O.radix_sort(arr).tolist()
    for order in ("ScoreThenIndexAsc", "ScoreThenIndexDesc", "IndexAsc", "IndexDesc"):
        runs = []
        for k in range(5):
            a = np.zeros(1000 + k, F.MATCH_DTYPE)
            a["index"] = rng.choice(1 << 20, len(a), replace=False) * 5 + k
            a["score"] = rng.integers(0, 300, len(a))
            key = {"ScoreThenIndexAsc": (a["index"], -a["score"].astype(np.int64)), "ScoreThenIndexDesc": (-a["index"].astype(np.int64), -a["score"].astype(np.int64)),
                   "IndexAsc": (a["index"],), "IndexDesc": (-a["index"].astype(np.int64),)}[order]
            runs.append(a[np.lexsort(key)])
        assert F.k_merge_matches(F.SortStrategy[order], runs).tolist() == O.k_merge(order, runs).tolist(), order


def test_scoring_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(F.FrizbeeError):
        F.Matcher("abc").match_list(["abc"])


def test_parse_query_matches_the_references_known_answers():
    # Pattern::parse / parse_query (src/pattern.rs:307-382), host logic of the boundary (no GPU involved)
    import json
    lt = json.load(open(os.path.join(ROOT, "tests", "golden", "literal.json")))
    for atom, needle, matching, negated in lt["parse_atoms"]:
        got = F.parse_query(atom.replace(" ", "\\ ") if " " in atom and "\\ " not in atom else atom)
        assert len(got) == 1
        assert (got[0].needle, None if got[0].matching is None else got[0].matching.name, got[0].negated) == (needle, matching, negated), atom
    for query, needles, ref in lt["parse_queries"]:
        assert [p.needle for p in F.parse_query(query)] == needles, ref
    for query in ("foo !^bar 'lit baz$ ^ex$ !neg a\\ b \\!x ! \\", "다나 !é$ ^\\^x"):
        got, want = F.parse_query(query), O.parse_query(query)
        assert [(p.needle, p.negated, None if p.matching is None else p.matching.name) for p in got] == [(p["needle"], p["negated"], p["matching"]) for p in want]


def test_parse_query_differential_against_the_oracle_on_random_queries():
    # two separately written parsers (host side of the product, oracle) over random strings of the syntax's alphabet
    rng = np.random.default_rng(77)
    alpha = ["a", "b", "Z", "é", "다", " ", "  ", "\t", "\\", "\\\\", "!", "^", "$", "'", "\\ ", "\\!", "\\^", "\\$", "\\'", "　", "!^", "$ "]
    for _ in range(3000):
        q = "".join(alpha[int(i)] for i in rng.integers(0, len(alpha), int(rng.integers(0, 12))))
        got = [(p.needle, p.negated, None if p.matching is None else p.matching.name) for p in F.parse_query(q)]
        want = [(p["needle"], p["negated"], p["matching"]) for p in O.parse_query(q)]
        assert got == want, repr(q)


def test_matcher_new_differential_on_random_scorings():
    # class selection (score_fits_in_u8), Smart case / unicode resolution and the overflow guards, host side vs oracle
    rng = np.random.default_rng(1234)
    pool = ["a", "B", "0", "_", "é", "다", "ß", "Z"]
    agree_panics = ok = 0
    for _ in range(1500):
        needle = "".join(pool[int(i)] for i in rng.integers(0, len(pool), int(rng.integers(1, 20))))
        if len(needle.encode()) > 64:
            continue
        big = rng.random() < 0.15
        sc = [int(rng.integers(0, 60000 if big else 40)) for _ in range(9)]
        casing = ["Ignore", "Smart", "Respect"][int(rng.integers(0, 3))]
        uni = ["Ignore", "Smart", "Always"][int(rng.integers(0, 3))]
        matching = ["Fuzzy", "Substring"][int(rng.integers(0, 2))]
        try:
            want = O.Matcher(needle, lanes=(64, 64, 32), scoring=sc, casing=casing, unicode=uni, matching=matching).info()
            oerr = None
        except RuntimeError as e:
            want, oerr = None, str(e)
        fc = F.Config(casing=F.CaseMatching[casing], unicode=F.UnicodeMatching[uni], scoring=F.Scoring(*sc), matching=F.Matching[matching], pf_lanes=64)
        try:
            got = F.Matcher(needle, fc).info()
            ferr = None
        except F.PanicError as e:
            got, ferr = None, str(e)
        assert (oerr is None) == (ferr is None), (needle, sc, oerr, ferr)
        if oerr is not None:
            assert ferr == oerr, (needle, sc)
            agree_panics += 1
        elif matching == "Fuzzy":
            assert (got["pf_lanes"], got["sw_lanes"], got["use_u8"]) == (want["pf_lanes"], want["sw_lanes"], want["use_u8"]), (needle, sc)
            ok += 1
    assert agree_panics > 20 and ok > 300


def test_needles_a_rust_str_cannot_hold_are_refused():
    # overlong encodings, UTF-16 surrogates, scalars above U+10FFFF, truncated sequences: `&str` is always valid UTF-8
    for bad in (b"\xc0\xaf", b"\xe0\x80\xaf", b"\xf0\x80\x80\xaf", b"\xed\xa0\x80", b"\xed\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf8\x88\x80\x80\x80", b"a\xc3", b"\x80"):
        with pytest.raises(Exception, match="not valid UTF-8"):
            F.Matcher(bad)
    for good in ("\u00e9", "\ud7ff", "\ue000", "\U0010ffff", "\u07ff\u0800"):
        F.Matcher(good)


def test_k_merge_rejects_null_buffers():
    import ctypes as C
    lens = (C.c_size_t * 1)(3)
    assert F.lib().fzb_k_merge_matches(0, None, lens, 1, None) != 0


def test_unicode_dfa_is_the_unicode_prefilter():
    """The byte-level DFA that replaces superset filter + lane-exact window pass on the unicode path with 0 typos (host.hip) accepts
    exactly what the reference's unicode prefilter accepts (src/prefilter/algo/unicode.rs:118-219, oracle at 16 / 32 / 64 lanes),
    both case modes, scalars of 1-4 bytes, also on byte strings that are not valid UTF-8."""
    import random
    rng = random.Random(12)
    alphabets = ["abéÉüÜ_ ", "aéñ中文😀b ", "إنماab ", "ΑαΒβΓγ xyz", "ßẞss", "éÉeE", "𐐀𐐨a𝒳"]
    checked = accepted = 0
    for it in range(6000):
        al = rng.choice(alphabets)
        needle = "".join(rng.choice(al) for _ in range(rng.randint(1, 6)))
        casing = rng.choice([F.CaseMatching.Ignore, F.CaseMatching.Respect, F.CaseMatching.Smart])
        m = F.Matcher(needle, F.Config(max_typos=0, casing=casing, unicode=F.UnicodeMatching.Always, pf_lanes=64))
        cs = O.respects_case_for(casing.name, needle)
        for _ in range(6):
            hay = "".join(rng.choice(al) for _ in range(rng.randint(0, 40))).encode()
            if rng.random() < 0.2 and hay:  # cut inside a scalar / repeat a lead byte: not valid UTF-8 any more
                cut = rng.randrange(len(hay))
                hay = hay[:cut] + hay[cut:cut + 1] + hay[cut:]
            got = F.lib().fzb_debug_unicode_dfa_accepts(m.h, hay, len(hay))
            assert got in (0, 1), (needle, got)
            for lanes in (16, 32, 64):
                want = O.prefilter(needle, hay, 0, cs, True, lanes)[0]
                assert bool(got) == want, (needle, hay, casing, lanes, got, want)
            checked += 1
            accepted += got
    assert checked > 30000 and accepted > 5000
    assert F.lib().fzb_debug_unicode_dfa_accepts(F.Matcher("abc", F.Config(max_typos=0)).h, b"abc", 3) == -1  # ASCII path: no such DFA


def test_shard_ranges_of_the_c_abi_are_the_python_ones():
    # fzb_shard_ranges (host arithmetic of fzb_corpus_upload_sharded) == frizbee_amd.distributed.shard_range / shard_ranges_by_bytes
    from frizbee_amd.distributed import shard_range, shard_ranges_by_bytes
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 9, 1000, 100_003):
        ends = np.cumsum(rng.integers(0, 129, n).astype(np.uint64), dtype=np.uint64)
        for w in (1, 2, 3, 8):
            assert F.shard_ranges(ends, w) == [shard_range(n, k, w) for k in range(w)]
            r = F.shard_ranges(ends, w, by_bytes=True)
            assert r == shard_ranges_by_bytes(ends, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
    # hand-computed: ends 10,20,30,40 in two shards by bytes -> the byte target 20 is where haystack 2 starts
    assert F.shard_ranges(np.array([10, 20, 30, 40], np.uint64), 2, by_bytes=True) == [(0, 2), (2, 4)]
    assert F.shard_ranges(np.array([10, 25, 30, 40], np.uint64), 2, by_bytes=True) == [(0, 2), (2, 4)]  # the straddling haystack stays left
    with pytest.raises(F.FrizbeeError):
        F.shard_ranges(np.array([1], np.uint64), 0)


def test_multi_device_entry_points_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    with pytest.raises(F.FrizbeeError) as e:
        F.ShardedCorpus(["a", "b"], ndev=2)
    assert e.value.code == 4
    with pytest.raises(F.FrizbeeError) as e:
        F.Corpus(["a", "b"])
    assert e.value.code == 4  # no CPU fallback anywhere on the product path


def test_lcs_automaton_is_the_lcs_criterion():
    # typo configurations: the streaming filter's accept test as a DFA over the reachable bit-vector states (fzb_matcher_create) must be
    # `LCS(needle, haystack) >= rows - max_typos` with case folding (src/prefilter/mod.rs:1013-1084) - checked against a plain DP
    import ctypes as C

    def lcs(a, b):
        prev = [0] * (len(b) + 1)
        for x in a:
            cur = [0]
            for j, y in enumerate(b):
                cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
            prev = cur
        return prev[-1]

    rng = np.random.default_rng(12)
    seen_states = []
    for needle, k in (("deadbe", 1), ("deadbe", 2), ("aab", 1), ("abcabc", 3), ("x_Y-z", 2), ("abcdefghijkl", 2), ("aaaaaaaa", 3), ("ab", 1)):
        m = F.Matcher(needle, F.Config(max_typos=k, casing=F.CaseMatching.Ignore))
        ns = C.c_int32()
        alpha = (needle + needle.upper() + "q_ 0").encode()
        for _ in range(400):
            h = bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(0, 40))))
            got = F.lib().fzb_debug_lcs_dfa_accepts(m.h, h, len(h), C.byref(ns))
            if ns.value == 0:
                assert got == -1  # more reachable states than the table holds: the bit-vector kernel keeps this needle
                break
            assert got == int(lcs(needle.lower(), h.decode().lower()) + k >= len(needle)), (needle, k, h)
        seen_states.append(ns.value)
    assert all(n <= 226 for n in seen_states) and sum(n > 0 for n in seen_states) >= 6, seen_states
    print("LCS automaton states:", seen_states)
    # no automaton: 0 typos, no prefilter, and a needle whose reachable states exceed the table
    assert F.lib().fzb_debug_lcs_dfa_accepts(F.Matcher("deadbe").h, b"x", 1, None) == -1
    assert F.lib().fzb_debug_lcs_dfa_accepts(F.Matcher("deadbe", F.Config(max_typos=None)).h, b"x", 1, None) == -1
    assert F.lib().fzb_debug_lcs_dfa_accepts(F.Matcher("abcdefghijklmnopqrstuvwxyz012345", F.Config(max_typos=3)).h, b"x", 1, None) == -1


def test_scalar_lcs_automaton_is_the_unicode_typo_prefilter_on_single_chunk_haystacks():
    # unicode typo configurations (round 6): the streaming filter's automaton over (reachable LCS bit-vector, bytes of the scalar being read) -
    # fzb_matcher_create, build_scalar_lcs_dfa - against (1) LCS(needle scalars, the haystack's scalar occurrences) + k >= n computed directly and
    # (2) the oracle's unicode typo prefilter at a lane width the haystack fits in ONE chunk of; valid UTF-8 and arbitrary bytes
    import ctypes as C
    import pf_second_transcription as P2
    from test_oracle_reference_properties import scalar_lcs
    rng = np.random.default_rng(21)
    alpha = ["a", "B", "_", "é", "É", "ж", "다", "😀", "ن", "إ", "م", "ا", " "]
    seen, marginal = [], 0
    for needle, k, casing in (("إنما", 1, "Smart"), ("إنما", 2, "Smart"), ("إن", 1, "Smart"), ("éa", 1, "Ignore"), ("aÉжb", 2, "Ignore"), ("다😀a다", 1, "Respect"), ("жжжж", 2, "Smart"), ("aébécé", 3, "Smart")):
        m = F.Matcher(needle, F.Config(max_typos=k, casing=F.CaseMatching[casing]))
        cs = casing == "Respect" or (casing == "Smart" and any(c.isupper() for c in needle))
        chars = P2.case_needle_unicode(needle, cs)
        ns = C.c_int32()
        pool = [c.encode() for c in alpha] + [a for a, _ in chars] + [b for _, b in chars]
        for it in range(500):
            ln = int(rng.integers(0, 65))
            if it % 4 == 3:  # arbitrary bytes: truncated scalars, stray continuation and lead bytes
                raw = pool + [bytes([int(rng.integers(0, 256))]) for _ in range(4)] + [b"\x80", b"\xd8", b"\xf0\x9f"]
                h = b"".join(raw[int(rng.integers(0, len(raw)))] for _ in range(ln))[:ln]
            else:
                h = b""
                while True:
                    c = pool[int(rng.integers(0, len(pool)))]
                    if len(h) + len(c) > ln:
                        break
                    h += c
            got = F.lib().fzb_debug_lcs_dfa_accepts(m.h, h, len(h), C.byref(ns))
            assert ns.value > 0 and got in (0, 1), (needle, k, ns.value)
            slack = scalar_lcs(chars, h) + k - len(chars)
            marginal += slack == 0
            assert got == int(slack >= 0), (needle, k, h, slack)
            assert got == int(O.prefilter(needle, h, k, cs, True, 64)[0]), (needle, k, h)
        seen.append(ns.value)
    assert all(0 < n <= 226 for n in seen) and marginal > 300, (seen, marginal)
    print("scalar LCS automaton states:", seen)


def test_class_composite_automaton_equals_the_byte_automaton():
    # the ragged filter's table (G byte transitions composed over the K byte classes, fzb_matcher_create) must decide exactly like the
    # byte-level automaton it was built from: ordered subsequence (0 typos), the LCS criterion (typos), the unicode prefilter, KMP (substring)
    import ctypes as C

    def subseq(n, h):
        it = iter(h)
        return all(c in it for c in n)

    rng = np.random.default_rng(21)
    kg = (C.c_int32 * 2)()
    seen = []
    for needle, cfg in (("deadbeef", dict()), ("linux", dict()), ("a", dict()), ("x_Y-z", dict()), ("deadbe", dict(max_typos=2)), ("abcabc", dict(max_typos=1)),
                        ("إنما", dict()), ("éa", dict()), ("dea", dict(matching=F.Matching.Substring)), ("abcdefghijklmnop", dict())):
        m = F.Matcher(needle, F.Config(**cfg))
        info = m.info()
        alpha = (needle + needle.upper() + needle.lower() + "q_ 0é").encode()
        for _ in range(600):
            h = bytes(alpha[int(x)] for x in rng.integers(0, len(alpha), int(rng.integers(0, 45))))
            got = F.lib().fzb_debug_cdfa_state(m.h, h, len(h), kg)
            if got < 0:
                break
            if cfg.get("max_typos"):
                want = F.lib().fzb_debug_lcs_dfa_accepts(m.h, h, len(h), None)
            elif info["unicode"]:
                want = F.lib().fzb_debug_unicode_dfa_accepts(m.h, h, len(h))
            elif cfg.get("matching"):
                want = int(needle.lower().encode() in h.lower()) if not info["case_sensitive"] else int(needle.encode() in h)
            else:
                hh, nn = (h, needle.encode()) if info["case_sensitive"] else (h.lower(), needle.lower().encode())
                want = int(subseq(nn, hh))
            assert got == want, (needle, cfg, h, got, want)
        seen.append((needle, kg[0], kg[1]))
    print("class-composite automata (needle, K, G):", seen)
    assert sum(1 for _, k, g in seen if g == 4) >= 3 and sum(1 for _, k, g in seen if g == 2) >= 1


def test_shard_comm_argument_checks_and_loud_failure_without_a_gpu():
    """fzb_shard_comm_* (csrc/host_rccl.hip), host side: bad arguments are FZB_ERR_INVALID before RCCL is touched; without a GPU the
    communicator cannot exist and says why (FZB_ERR_HIP) - there is no other transport to fall back to."""
    import ctypes as C
    import torch
    l = F.lib()
    out = C.c_void_p()
    uid = bytes(128)
    assert l.fzb_rccl_unique_id(None) == 1
    assert l.fzb_shard_comm_create(None, 0, 1, C.byref(out)) == 1 and l.fzb_shard_comm_create(C.c_char_p(uid), 0, 1, None) == 1
    for rank, world in ((-1, 2), (2, 2), (0, 0)):
        assert l.fzb_shard_comm_create(C.c_char_p(uid), rank, world, C.byref(out)) == 1 and not out.value
        assert b"outside a world" in l.fzb_last_error()
    assert l.fzb_shard_comm_rank(None) == -1 and l.fzb_shard_comm_world(None) == 0
    l.fzb_shard_comm_free(None)
    n, res = C.c_size_t(), C.c_void_p()
    assert l.fzb_match_list_parallel_rccl(None, None, 0, None, 0, C.byref(res), C.byref(n)) == 1
    b = (C.c_uint64 * 2)()
    assert l.fzb_shard_comm_last_exchange(None, b) == 1
    if not torch.cuda.is_available():
        assert l.fzb_shard_comm_create(C.c_char_p(uid), 0, 1, C.byref(out)) == 4 and not out.value
        assert l.fzb_last_error()
