"""The Unicode case table behind `case_needle_unicode` (src/prefilter/mod.rs:71-96: a scalar's flip = its single-scalar lower / upper
mapping of the same UTF-8 width) and `CaseMatching::Smart` (`char::is_uppercase`, src/lib.rs:370-376), Unicode 16.0 (Rust >= 1.89,
Cargo.toml:10).  Product (frizbee_amd/csrc/unicode_case_table.inc, generated from Python's unicodedata + a hand-written 14.0-16.0
delta) and oracle (oracle/unicode_case_table.inc, generated from ICU 70 + the `regex` module's tables) carry one copy each, made by
different routes (tools/gen_unicode_case.py, tools/gen_unicode_case_icu.py); this file is what holds them together:
  * the two committed tables are identical, entry for entry;
  * both equal a THIRD derivation written here (unicodedata's properties + the delta, rule restated);
  * the `regex` module's \\p{Uppercase} (its own database) agrees on the Smart-casing set for every scalar it assigns up to 16.0;
  * the compiled product library answers Smart casing (`fzb_matcher_info`: case_sensitive) from that set."""
import os
import re
import sys
import unicodedata

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _parse(path):
    text = open(path).read()
    flip_part, upper_part = text.split("FZB_UPPER_RANGES[][2]")
    pairs = lambda t: [(int(a, 16), int(b, 16)) for a, b in re.findall(r"\{0x([0-9A-F]+),0x([0-9A-F]+)\}", t)]  # noqa: E731
    flips, ranges = pairs(flip_part), pairs(upper_part)
    assert len(flips) == int(re.search(r"FZB_CASE_FLIP_LEN = (\d+)", text).group(1))
    assert len(ranges) == int(re.search(r"FZB_UPPER_RANGES_LEN = (\d+)", text).group(1))
    return flips, ranges


PRODUCT = os.path.join(ROOT, "frizbee_amd", "csrc", "unicode_case_table.inc")
ORACLE = os.path.join(ROOT, "oracle", "unicode_case_table.inc")


def test_product_and_oracle_tables_are_identical():
    pf, pr = _parse(PRODUCT)
    of, orr = _parse(ORACLE)
    assert pf == of and pr == orr
    assert pf == sorted(pf) and len({a for a, _ in pf}) == len(pf)          # the lookup is a binary search: sorted, one entry per scalar
    assert all(a <= b for a, b in pr) and all(pr[i][1] + 1 < pr[i + 1][0] for i in range(len(pr) - 1))  # maximal, disjoint, ascending ranges


def _utf8len(cp):
    return 1 if cp < 0x80 else 2 if cp < 0x800 else 3 if cp < 0x10000 else 4


def _third_derivation():
    """The rule restated over unicodedata's DERIVED properties (str.isupper / islower on one scalar = Uppercase / Lowercase, str.lower /
    upper = the full mappings Rust's to_lowercase / to_uppercase yield), then the scalars Unicode 14.0-16.0 added."""
    import gen_unicode_case as G  # the delta list (data, cited line by line from UnicodeData.txt 16.0); the rule is NOT imported

    assert unicodedata.unidata_version == "13.0.0"
    flips, upper = {}, set()
    for cp in range(0x80, 0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        c = chr(cp)
        if c.isupper():
            upper.add(cp)
            other = c.lower()
        elif c.islower():
            other = c.upper()
        else:
            continue
        if len(other) == 1 and other != c and _utf8len(ord(other)) == _utf8len(cp):
            flips[cp] = ord(other)
    for u, l in G.PAIRS_14_TO_16:
        assert u not in flips and l not in flips and _utf8len(u) == _utf8len(l)
        flips[u], flips[l] = l, u
        upper.add(u)
    upper.update(G.UPPER_ONLY_14_TO_16)
    return flips, upper


def _expand(ranges):
    out = set()
    for a, b in ranges:
        out.update(range(a, b + 1))
    return out


def test_tables_equal_a_third_derivation():
    flips, upper = _third_derivation()
    pf, pr = _parse(PRODUCT)
    assert dict(pf) == flips
    assert _expand(pr) == upper
    # spot checks a reader can verify by eye: same-width pairs, and the classic exclusions
    d = dict(pf)
    assert d[0xE9] == 0xC9 and d[0xC9] == 0xE9 and d[0x3B1] == 0x391 and d[0x10D50] == 0x10D70
    assert 0xDF not in d            # ß: uppercase is "SS" (two scalars)
    assert 0x130 not in d           # İ: lowercase is two scalars
    assert 0x212A not in d          # KELVIN SIGN: lowercase k is one byte, not three
    assert 0x1E9E not in d          # ẞ -> ß changes the UTF-8 width (3 -> 2)
    assert 0x130 in upper and 0x212A in upper and 0x1E9E in upper and 0xDF not in upper


def test_regex_modules_uppercase_property_agrees_on_the_smart_casing_set():
    regex = pytest.importorskip("regex")
    _, pr = _parse(PRODUCT)
    ours = _expand(pr)
    try:
        import gen_unicode_case_icu as I  # only its list of scalars that exist since Unicode 17.0 (newer than the reference's toolchain)
    except (OSError, AssertionError) as e:  # the module binds ICU 70 when imported
        pytest.skip(f"ICU 70 not loadable here: {e}")
    pat = regex.compile(r"\p{Uppercase}")
    theirs = {cp for cp in range(0x80, 0x110000) if not 0xD800 <= cp <= 0xDFFF and cp not in I.UNICODE_17_ONLY and pat.fullmatch(chr(cp))}
    assert theirs == ours, (sorted(theirs - ours)[:10], sorted(ours - theirs)[:10])


def test_compiled_library_answers_smart_casing_from_the_table():
    import frizbee_amd as F

    try:
        F.lib()
    except F.FrizbeeError:
        pytest.skip("libfrizbee_hip.so not built")
    flips, upper = _third_derivation()
    sample = sorted(upper)[::37] + [0x10D50, 0xA7DC, 0x1C89]
    for cp in sample:
        assert F.Matcher("a" + chr(cp), F.Config(casing=F.CaseMatching.Smart, unicode=F.UnicodeMatching.Always)).info()["case_sensitive"] is True, hex(cp)
    for cp in [flips[c] for c in sample if c in flips] + [0xDF, 0x4E2D, 0x10D70]:
        assert F.Matcher("a" + chr(cp), F.Config(casing=F.CaseMatching.Smart, unicode=F.UnicodeMatching.Always)).info()["case_sensitive"] is False, hex(cp)
