"""ctypes loader for oracle/libfrizbee_oracle.so (TEST INFRASTRUCTURE: the CPU restatement of the
reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DEFAULT_SCORING = [12, 6, 5, 1, 12, 4, 4, 8, 4]
CASING = {"Ignore": 0, "Smart": 1, "Respect": 2}
UNICODE = {"Ignore": 0, "Smart": 1, "Always": 2}
SORT = {"ScoreThenIndexAsc": 0, "ScoreThenIndexDesc": 1, "IndexAsc": 2, "IndexDesc": 3}
MATCHING = {"Fuzzy": 0, "Exact": 1, "Prefix": 2, "Suffix": 3, "Substring": 4}

MATCH_DTYPE = np.dtype([("index", "<u4"), ("score", "<u2"), ("exact", "u1"), ("_pad", "u1")])


class FzoPattern(C.Structure):
    _fields_ = [("needle", C.c_char_p), ("needle_len", C.c_size_t), ("negated", C.c_int32), ("has_max_typos", C.c_int32), ("max_typos", C.c_int32),
                ("casing", C.c_int32), ("unicode", C.c_int32), ("has_scoring", C.c_int32), ("scoring", C.c_uint16 * 9), ("matching", C.c_int32)]


class FzoConfig(C.Structure):
    _fields_ = [("max_typos", C.c_int32), ("casing", C.c_int32), ("unicode", C.c_int32), ("sort", C.c_int32), ("scoring", C.c_uint16 * 9), ("matching", C.c_int32)]


def build(native=False, force=False):
    """Portable build (the parity checker): libfrizbee_oracle.so.  native=True: libfrizbee_oracle_native.so, compiled
    -march=native with the AVX-512 lane-vector types when the host has them (the CPU-baseline build; host-specific, so
    it is listed in .gpurunignore and bench.py rebuilds it with force=True on the machine it times)."""
    name = "libfrizbee_oracle_native.so" if native else "libfrizbee_oracle.so"
    so = os.path.join(ORACLE_DIR, name)
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle_capi.cpp", "frizbee_oracle.hpp", "unicode_case_table.inc", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR] + (["-B"] if force else []) + [name], stdout=subprocess.DEVNULL)
    return so


_libs = {}


def lib(native=False):
    l = _libs.get(bool(native))
    if l is None:
        l = C.CDLL(build(native))
        l.fzo_last_error.restype = C.c_char_p
        l.fzo_simd_kind.restype = C.c_char_p
        l.fzo_prefilter.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        l.fzo_sw_score.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.fzo_sw_indices.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint16), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]
        l.fzo_sw_score_typos.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_int, C.c_int]
        l.fzo_match_list_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        l.fzo_multi_match_list_indices.argtypes = l.fzo_match_list_indices.argtypes
        l.fzo_greedy.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint16), C.c_int, C.c_int]
        l.fzo_score_fits_in_u8.argtypes = [C.c_size_t, C.POINTER(C.c_uint16)]
        l.fzo_max_needle_len.argtypes = [C.POINTER(C.c_uint16)]
        l.fzo_matcher_create.restype = C.c_void_p
        l.fzo_matcher_create.argtypes = [C.POINTER(FzoConfig), C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        l.fzo_matcher_free.argtypes = [C.c_void_p]
        l.fzo_matcher_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.fzo_match_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_long, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzo_match_list_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_long, C.POINTER(C.c_size_t)]
        l.fzo_score_count_unordered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_long, C.POINTER(C.c_size_t)]
        l.fzo_multi_create.restype = C.c_void_p
        l.fzo_multi_create.argtypes = [C.POINTER(FzoConfig), C.POINTER(FzoPattern), C.c_size_t, C.c_int, C.c_int, C.c_int]
        l.fzo_multi_free.argtypes = [C.c_void_p]
        l.fzo_multi_match_list.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzo_parse_query.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        l.fzo_free.argtypes = [C.c_void_p]
        l.fzo_radix_sort.argtypes = [C.c_void_p, C.c_size_t]
        l.fzo_k_merge.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.fzo_respects_case_for.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        l.fzo_case_needle_unicode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
        _libs[bool(native)] = l
    return l


def simd_kind(native=False):
    return lib(native).fzo_simd_kind().decode()


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("utf-8")


def _scoring(s):
    return (C.c_uint16 * 9)(*(s or DEFAULT_SCORING))


def prefilter(needle, haystack, max_typos=0, case_sensitive=False, unicode=False, lanes=64, native=False):
    out = (C.c_uint64 * 3)()
    n, h = _b(needle), _b(haystack)
    rc = lib(native).fzo_prefilter(n, len(n), h, len(h), max_typos, int(case_sensitive), int(unicode), lanes, out)
    if rc:
        raise RuntimeError(lib().fzo_last_error().decode())
    return (bool(out[0]), int(out[1]), int(out[2]))


def sw_score(needle, haystack, scoring=None, case_sensitive=False, include_prefix=True, unicode=False, lanes=8, is_u8=False, native=False):
    n, h = _b(needle), _b(haystack)
    r = lib(native).fzo_sw_score(n, len(n), h, len(h), _scoring(scoring), int(case_sensitive), int(include_prefix), int(unicode), lanes, int(is_u8))
    if r < 0:
        raise RuntimeError(lib().fzo_last_error().decode())
    return r


def sw_indices(needle, haystack, start_pos=0, unicode=False, max_typos=None, scoring=None, case_sensitive=False, lanes=8, is_u8=False, native=False):
    """score_haystack[_unicode]_indices (smith_waterman/algo/mod.rs:49-152) -> (score, matched byte positions in reverse order)"""
    n, h = _b(needle), _b(haystack)
    cap = 4 * len(n) + 16
    out = (C.c_uint32 * cap)()
    cnt = C.c_size_t()
    r = lib(native).fzo_sw_indices(n, len(n), h, len(h), _scoring(scoring), int(case_sensitive), start_pos, int(unicode), lanes, int(is_u8), -1 if max_typos is None else max_typos, out, cap, C.byref(cnt))
    if r < 0:
        raise RuntimeError(lib(native).fzo_last_error().decode())
    return r, list(out[: cnt.value])


def sw_score_typos(needle, haystack, max_typos, scoring=None, case_sensitive=False, lanes=8, is_u8=False):
    """get_score_typos of the reference's tests: the score if an alignment path within the typo budget exists, else None"""
    n, h = _b(needle), _b(haystack)
    r = lib().fzo_sw_score_typos(n, len(n), h, len(h), _scoring(scoring), int(case_sensitive), lanes, int(is_u8), max_typos)
    if r < -1:
        raise RuntimeError(lib().fzo_last_error().decode())
    return None if r < 0 else r


def greedy(needle, haystack, scoring=None, case_sensitive=False, include_prefix=True):
    n, h = _b(needle), _b(haystack)
    return lib().fzo_greedy(n, len(n), h, len(h), _scoring(scoring), int(case_sensitive), int(include_prefix))


def score_fits_in_u8(needle_len, scoring=None):
    return bool(lib().fzo_score_fits_in_u8(needle_len, _scoring(scoring)))


def max_needle_len(scoring=None):
    return lib().fzo_max_needle_len(_scoring(scoring))


def pack(haystacks):
    """list[str|bytes] -> (uint8 bytes array (padded 64 zero bytes), uint64 exclusive end offsets)"""
    bs = [_b(h) for h in haystacks]
    ends = np.cumsum(np.fromiter((len(b) for b in bs), dtype=np.uint64, count=len(bs)), dtype=np.uint64) if bs else np.zeros(0, np.uint64)
    data = np.frombuffer(b"".join(bs) + b"\0" * 64, dtype=np.uint8).copy()
    return data, ends


def make_config(max_typos=0, casing="Smart", unicode="Smart", sort="ScoreThenIndexAsc", scoring=None, matching="Fuzzy"):
    cfg = FzoConfig()
    cfg.max_typos = -1 if max_typos is None else int(max_typos)
    cfg.casing = CASING[casing] if isinstance(casing, str) else int(casing)
    cfg.unicode = UNICODE[unicode] if isinstance(unicode, str) else int(unicode)
    cfg.sort = SORT[sort] if isinstance(sort, str) else int(sort)
    for i, v in enumerate(scoring or DEFAULT_SCORING):
        cfg.scoring[i] = v
    cfg.matching = MATCHING[matching] if isinstance(matching, str) else int(matching)
    return cfg


class Matcher:
    """Oracle `Matcher` emulating the backend pair an ISA would select (matcher/mod.rs:448-498):
    (pf_lanes, sw_lanes_u8, sw_lanes_u16) = (64,64,32) AVX-512+VBMI, (32,32,16) AVX2, (16,16,8) SSE/NEON/scalar."""

    def __init__(self, needle, lanes=(64, 64, 32), native=False, **cfg):
        self.cfg = make_config(**cfg)
        self.lib = lib(native)
        n = _b(needle)
        self.h = self.lib.fzo_matcher_create(C.byref(self.cfg), n, len(n), *lanes)
        if not self.h:
            raise RuntimeError(self.lib.fzo_last_error().decode())

    def info(self):
        out = (C.c_int * 3)()
        self.lib.fzo_matcher_info(self.h, out)
        return dict(pf_lanes=out[0], sw_lanes=out[1], use_u8=bool(out[2]))

    def match_packed(self, data, ends, threads=-1):
        out = C.c_void_p()
        n = C.c_size_t()
        rc = self.lib.fzo_match_list(self.h, data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), threads, C.byref(out), C.byref(n))
        if rc:
            raise RuntimeError(self.lib.fzo_last_error().decode())
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 8,))[: n.value * 8].view(MATCH_DTYPE).copy()
        self.lib.fzo_free(out)
        return arr

    def count_packed(self, data, ends, threads=-1):
        n = C.c_size_t()
        rc = self.lib.fzo_match_list_count(self.h, data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), threads, C.byref(n))
        if rc:
            raise RuntimeError(self.lib.fzo_last_error().decode())
        return n.value

    def score_count_unordered(self, data, ends, threads):
        """timing aid: the parallel scoring loop without the ordering step; returns the number of matches"""
        n = C.c_size_t()
        rc = self.lib.fzo_score_count_unordered(self.h, data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), threads, C.byref(n))
        if rc:
            raise RuntimeError(self.lib.fzo_last_error().decode())
        return n.value

    def match_list(self, haystacks):
        return self.match_packed(*pack(haystacks))

    def match_list_indices(self, haystacks):
        """`Matcher::match_list_indices` for one pattern, index order: (records, list of index lists in reverse byte order)"""
        return _indices_call(self.lib, self.lib.fzo_match_list_indices, self.h, haystacks)

    def match_list_indices_ordered(self, haystacks):
        """`Matcher::match_list_indices` with its ordering step: list of (index, score, exact, indices)."""
        return _order_indices(*self.match_list_indices(haystacks), self.cfg.sort)

    def match_list_parallel(self, haystacks, threads):
        return self.match_packed(*pack(haystacks), threads=threads)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.fzo_matcher_free(self.h)
                self.h = None
        except Exception:
            pass


def _indices_call(l, fn, handle, haystacks):
    data, ends = pack(haystacks)
    out, n, oi, oo = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_void_p()
    rc = fn(handle, data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), C.byref(out), C.byref(n), C.byref(oi), C.byref(oo))
    if rc:
        raise RuntimeError(l.fzo_last_error().decode())
    recs = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 8,))[: n.value * 8].view(MATCH_DTYPE).copy()
    offs = np.ctypeslib.as_array(C.cast(oo, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
    flat = np.ctypeslib.as_array(C.cast(oi, C.POINTER(C.c_uint32)), shape=(max(int(offs[-1]), 1),)).copy()
    for p in (out, oi, oo):
        l.fzo_free(p)
    return recs, [flat[int(offs[i]) : int(offs[i + 1])].tolist() for i in range(n.value)]


def _order_indices(recs, idx, sort):
    """The ordering step of `Matcher::match_list_indices` (src/matcher/mod.rs:268-273): reverse for the *Desc strategies, then a
    stable sort by descending score for the Score* ones."""
    items = [(int(r["index"]), int(r["score"]), bool(r["exact"]), ix) for r, ix in zip(recs, idx)]
    if sort in (SORT["IndexDesc"], SORT["ScoreThenIndexDesc"]):
        items.reverse()
    if sort in (SORT["ScoreThenIndexAsc"], SORT["ScoreThenIndexDesc"]):
        items.sort(key=lambda t: -t[1])  # list.sort is stable, like sort_by_key
    return items


INHERIT = "inherit"


def P(needle, negated=False, max_typos=INHERIT, casing=None, unicode=None, scoring=None, matching=None):
    """One pattern of a multi-pattern matcher (reference `Pattern` + `PatternConfig`, src/pattern.rs:9-18, 230-262; fuzzy matching
    only).  max_typos=INHERIT is PatternConfig's `None`; an int is `Some(k)`."""
    return dict(needle=needle, negated=negated, max_typos=max_typos, casing=casing, unicode=unicode, scoring=scoring, matching=matching)


class MultiMatcher:
    """Oracle `Matcher::from_patterns` (src/matcher/mod.rs:95-111, 178-204; src/matcher/multi.rs)."""

    def __init__(self, patterns, lanes=(64, 64, 32), native=False, **cfg):
        self.cfg = make_config(**cfg)
        self.lib = lib(native)
        arr = (FzoPattern * max(len(patterns), 1))()
        self._keep = []
        for i, p in enumerate(patterns):
            n = _b(p["needle"])
            self._keep.append(n)
            arr[i].needle, arr[i].needle_len, arr[i].negated = n, len(n), int(p["negated"])
            arr[i].has_max_typos = int(p["max_typos"] != INHERIT)
            arr[i].max_typos = 0 if p["max_typos"] == INHERIT else int(p["max_typos"])
            arr[i].casing = -1 if p["casing"] is None else CASING[p["casing"]]
            arr[i].unicode = -1 if p["unicode"] is None else UNICODE[p["unicode"]]
            arr[i].has_scoring = int(p["scoring"] is not None)
            arr[i].matching = -1 if p.get("matching") is None else MATCHING[p["matching"]]
            for k, v in enumerate(p["scoring"] or DEFAULT_SCORING):
                arr[i].scoring[k] = v
        self.h = self.lib.fzo_multi_create(C.byref(self.cfg), arr, len(patterns), *lanes)
        if not self.h:
            raise RuntimeError(self.lib.fzo_last_error().decode())

    def _run(self, data, ends, mode):
        out, n = C.c_void_p(), C.c_size_t()
        rc = self.lib.fzo_multi_match_list(self.h, data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), mode, C.byref(out), C.byref(n))
        if rc:
            raise RuntimeError(self.lib.fzo_last_error().decode())
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 8,))[: n.value * 8].view(MATCH_DTYPE).copy()
        self.lib.fzo_free(out)
        return arr

    def match_packed(self, data, ends):
        return self._run(data, ends, 0)

    def match_list(self, haystacks):
        return self._run(*pack(haystacks), 0)

    def match_list_indices(self, haystacks):
        """`Matcher::match_list_indices` over CompiledPatterns (match_one_indices_multi, src/matcher/multi.rs:56-82), haystack order"""
        return _indices_call(self.lib, self.lib.fzo_multi_match_list_indices, self.h, haystacks)

    def match_list_indices_ordered(self, haystacks):
        return _order_indices(*self.match_list_indices(haystacks), self.cfg.sort)

    def reference_composition(self, haystacks):
        """the reference's own test oracle for the composition (tests/api_properties.rs:316-361), index order"""
        return self._run(*pack(haystacks), 1)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.fzo_multi_free(self.h)
                self.h = None
        except Exception:
            pass


def parse_query(query):
    """Oracle `Pattern::parse_query` (src/pattern.rs:186-222) -> list of P(...) with the matching mode the syntax implies"""
    q = _b(query)
    buf = C.create_string_buffer(16 * len(q) + 256)
    n = lib().fzo_parse_query(q, len(q), buf, len(buf))
    if n < 0:
        raise RuntimeError(lib().fzo_last_error().decode())
    inv = {v: k for k, v in MATCHING.items()}
    out = []
    for line in buf.value.decode().splitlines():
        neg, matching, hexs = (line.split(" ") + [""])[:3]
        out.append(P(bytes.fromhex(hexs).decode("utf-8"), negated=neg == "1", matching=None if int(matching) < 0 else inv[int(matching)]))
    return out


def radix_sort(arr):
    a = np.ascontiguousarray(arr.copy())
    lib().fzo_radix_sort(a.ctypes.data, len(a))
    return a


def k_merge(order, runs):
    lens = np.array([len(r) for r in runs], dtype=np.uint64)
    cat = np.concatenate(runs) if runs else np.zeros(0, MATCH_DTYPE)
    cat = np.ascontiguousarray(cat)
    out = np.zeros(len(cat), MATCH_DTYPE)
    lib().fzo_k_merge(SORT[order] if isinstance(order, str) else order, cat.ctypes.data, lens.ctypes.data, len(runs), out.ctypes.data)
    return out


def respects_case_for(casing, needle):
    n = _b(needle)
    return bool(lib().fzo_respects_case_for(CASING[casing], n, len(n)))


def case_needle_unicode(needle, case_sensitive):
    n = _b(needle)
    buf = C.create_string_buffer(9 * (len(n) + 1))
    k = lib().fzo_case_needle_unicode(n, len(n), int(case_sensitive), buf, len(n) + 1)
    raw = buf.raw
    return [(raw[9 * i : 9 * i + 4], raw[9 * i + 4 : 9 * i + 8], raw[9 * i + 8]) for i in range(k)]
