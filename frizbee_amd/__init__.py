"""frizbee_amd - MI355X (gfx950) backend for saghen/frizbee's batched fuzzy-scoring path.

A thin ctypes mirror of the reference's public surface for this path (`Matcher::new`, `match_list`,
`match_list_parallel`, `Config`, `Scoring`, `Match`, `radix_sort_matches`; reference src/lib.rs:110-138,
src/matcher/mod.rs:86-222, src/matcher/parallel.rs:18-89) over the C ABI in include/frizbee_hip.h.
All scoring runs in hand-written HIP kernels (frizbee_amd/csrc); there is no CPU fallback: importing
works anywhere, but every scoring call raises unless libfrizbee_hip.so is built and a GPU is present.
"""
import ctypes as C
import weakref
import enum
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FRIZBEE_HIP_LIB", os.path.join(_HERE, "libfrizbee_hip.so"))

MATCH_DTYPE = np.dtype([("index", "<u4"), ("score", "<u2"), ("exact", "u1"), ("_pad", "u1")])
MATCH_INDICES_DTYPE = np.dtype([("index", "<u4"), ("score", "<u2"), ("exact", "u1"), ("_pad", "u1"), ("positions_begin", "<u4"), ("positions_len", "<u4")])


class FrizbeeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class PanicError(FrizbeeError):
    """The reference would `panic!` here; the message is the reference's panic text."""


class CaseMatching(enum.IntEnum):  # src/lib.rs:357-368
    Ignore = 0
    Smart = 1
    Respect = 2


class UnicodeMatching(enum.IntEnum):  # src/lib.rs:379-392
    Ignore = 0
    Smart = 1
    Always = 2


class SortStrategy(enum.IntEnum):  # src/lib.rs:311-326
    ScoreThenIndexAsc = 0
    ScoreThenIndexDesc = 1
    IndexAsc = 2
    IndexDesc = 3


class Matching(enum.IntEnum):  # src/lib.rs:414-427
    Fuzzy = 0
    Exact = 1
    Prefix = 2
    Suffix = 3
    Substring = 4


@dataclass
class Scoring:  # src/lib.rs:439-478, defaults src/const.rs:1-10
    match_score: int = 12
    mismatch_penalty: int = 6
    gap_open_penalty: int = 5
    gap_extend_penalty: int = 1
    prefix_bonus: int = 12
    capitalization_bonus: int = 4
    matching_case_bonus: int = 4
    exact_match_bonus: int = 8
    delimiter_bonus: int = 4

    def as_list(self):
        return [self.match_score, self.mismatch_penalty, self.gap_open_penalty, self.gap_extend_penalty, self.prefix_bonus,
                self.capitalization_bonus, self.matching_case_bonus, self.exact_match_bonus, self.delimiter_bonus]


@dataclass
class Config:  # src/lib.rs:236-271
    max_typos: "int | None" = 0
    casing: CaseMatching = CaseMatching.Smart
    unicode: UnicodeMatching = UnicodeMatching.Smart
    sort: SortStrategy = SortStrategy.ScoreThenIndexAsc
    scoring: Scoring = field(default_factory=Scoring)
    # which reference CPU backend to be bit-exact against; (0, 0) = what frizbee would pick on this host
    pf_lanes: int = 0
    sw_lanes: int = 0
    matching: Matching = Matching.Fuzzy


class _CScoring(C.Structure):
    _fields_ = [(n, C.c_uint16) for n in ("match_score", "mismatch_penalty", "gap_open_penalty", "gap_extend_penalty", "prefix_bonus",
                                          "capitalization_bonus", "matching_case_bonus", "exact_match_bonus", "delimiter_bonus")]


class _CConfig(C.Structure):
    _fields_ = [("max_typos", C.c_int32), ("casing", C.c_int32), ("unicode", C.c_int32), ("sort", C.c_int32), ("scoring", _CScoring),
                ("pf_lanes", C.c_uint16), ("sw_lanes", C.c_uint16), ("matching", C.c_int32)]


class _CPattern(C.Structure):
    _fields_ = [("needle_utf8", C.c_void_p), ("needle_len", C.c_size_t), ("negated", C.c_int32), ("has_max_typos", C.c_int32), ("max_typos", C.c_int32),
                ("casing", C.c_int32), ("unicode", C.c_int32), ("has_scoring", C.c_int32), ("scoring", _CScoring), ("matching", C.c_int32)]


_lib = None

SYMBOLS = [
    "fzb_last_error", "fzb_config_default", "fzb_matcher_create", "fzb_matcher_clone", "fzb_matcher_free", "fzb_matcher_info", "fzb_matcher_set_pattern", "fzb_matcher_set_config",
    "fzb_corpus_upload", "fzb_corpus_from_device", "fzb_corpus_set_max_len", "fzb_corpus_set_uniform_len", "fzb_corpus_free", "fzb_corpus_len", "fzb_match_list", "fzb_match_list_into",
    "fzb_match_list_device", "fzb_match_list_sorted_device", "fzb_match_list_parallel", "fzb_matches_free", "fzb_radix_sort_matches", "fzb_k_merge_matches",
    "fzb_matcher_reserve", "fzb_debug_unicode_dfa_accepts", "fzb_set_profiling", "fzb_last_timings", "fzb_last_stage_timings", "fzb_last_counters",
    "fzb_multi_matcher_create", "fzb_multi_matcher_free", "fzb_multi_matcher_len", "fzb_multi_match_list", "fzb_multi_match_list_device",
    "fzb_parse_query", "fzb_patterns_free", "fzb_match_list_indices", "fzb_match_indices_free", "fzb_multi_match_list_indices",
    "fzb_match_list_indices_into", "fzb_multi_match_list_into", "fzb_multi_match_list_indices_into",
    "fzb_device_count", "fzb_shard_ranges", "fzb_corpus_upload_sharded", "fzb_sharded_corpus_free", "fzb_sharded_corpus_len", "fzb_sharded_corpus_shards",
    "fzb_sharded_corpus_shard", "fzb_match_list_parallel_sharded", "fzb_debug_lcs_dfa_accepts", "fzb_debug_cdfa_state",
    "fzb_merge_shard_runs", "fzb_corpus_build_view", "fzb_debug_reload_knobs", "fzb_matcher_shard_report",
    "fzb_rccl_unique_id", "fzb_shard_comm_create", "fzb_shard_comm_free", "fzb_shard_comm_rank", "fzb_shard_comm_world", "fzb_match_list_parallel_rccl",
    "fzb_shard_comm_last_exchange",
]


def lib():
    """Loads libfrizbee_hip.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise FrizbeeError(4, f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` (make -C frizbee_amd/csrc)")
        l = C.CDLL(_LIB_PATH)
        l.fzb_last_error.restype = C.c_char_p
        l.fzb_config_default.argtypes = [C.POINTER(_CConfig)]
        l.fzb_matcher_create.argtypes = [C.POINTER(_CConfig), C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        l.fzb_matcher_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        l.fzb_matcher_free.argtypes = [C.c_void_p]
        l.fzb_matcher_set_pattern.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.fzb_matcher_set_config.argtypes = [C.c_void_p, C.POINTER(_CConfig)]
        l.fzb_matcher_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        l.fzb_corpus_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        l.fzb_corpus_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.POINTER(C.c_void_p)]
        l.fzb_corpus_set_max_len.argtypes = [C.c_void_p, C.c_uint32]
        l.fzb_corpus_set_uniform_len.argtypes = [C.c_void_p, C.c_uint32]
        l.fzb_corpus_free.argtypes = [C.c_void_p]
        l.fzb_corpus_len.argtypes = [C.c_void_p]
        l.fzb_corpus_len.restype = C.c_size_t
        l.fzb_match_list.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_match_list_into.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_match_list_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        l.fzb_match_list_sorted_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        l.fzb_match_list_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_matches_free.argtypes = [C.c_void_p]
        l.fzb_radix_sort_matches.argtypes = [C.c_void_p, C.c_size_t]
        l.fzb_k_merge_matches.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        l.fzb_matcher_reserve.argtypes = [C.c_void_p, C.c_void_p]
        l.fzb_debug_unicode_dfa_accepts.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.fzb_set_profiling.argtypes = [C.c_void_p, C.c_int]
        l.fzb_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.fzb_last_stage_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.fzb_last_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        l.fzb_multi_matcher_create.argtypes = [C.POINTER(_CConfig), C.POINTER(_CPattern), C.c_size_t, C.POINTER(C.c_void_p)]
        l.fzb_multi_matcher_free.argtypes = [C.c_void_p]
        l.fzb_parse_query.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(_CPattern)), C.POINTER(C.c_size_t)]
        l.fzb_patterns_free.argtypes = [C.POINTER(_CPattern), C.c_size_t]
        l.fzb_multi_matcher_len.argtypes = [C.c_void_p]
        l.fzb_multi_matcher_len.restype = C.c_size_t
        l.fzb_multi_match_list.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_multi_match_list_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        l.fzb_match_list_indices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        l.fzb_match_indices_free.argtypes = [C.c_void_p, C.c_void_p]
        l.fzb_multi_match_list_indices.argtypes = l.fzb_match_list_indices.argtypes
        l.fzb_match_list_indices_into.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        l.fzb_multi_match_list_indices_into.argtypes = l.fzb_match_list_indices_into.argtypes
        l.fzb_multi_match_list_into.argtypes = l.fzb_match_list_into.argtypes
        l.fzb_device_count.argtypes = [C.POINTER(C.c_int)]
        l.fzb_shard_ranges.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        l.fzb_corpus_upload_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        l.fzb_sharded_corpus_free.argtypes = [C.c_void_p]
        l.fzb_sharded_corpus_len.argtypes = [C.c_void_p]
        l.fzb_sharded_corpus_len.restype = C.c_size_t
        l.fzb_sharded_corpus_shards.argtypes = [C.c_void_p]
        l.fzb_sharded_corpus_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        l.fzb_match_list_parallel_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_matcher_shard_report.argtypes = [C.c_void_p]
        l.fzb_matcher_shard_report.restype = C.c_char_p
        l.fzb_merge_shard_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_corpus_build_view.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.fzb_rccl_unique_id.argtypes = [C.c_void_p]
        l.fzb_shard_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        l.fzb_shard_comm_free.argtypes = [C.c_void_p]
        l.fzb_shard_comm_rank.argtypes = [C.c_void_p]
        l.fzb_shard_comm_world.argtypes = [C.c_void_p]
        l.fzb_match_list_parallel_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        l.fzb_shard_comm_last_exchange.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        l.fzb_debug_lcs_dfa_accepts.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
        l.fzb_debug_cdfa_state.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
        _lib = l
    return _lib


def _check(rc):
    if rc:
        msg = lib().fzb_last_error().decode("utf-8", "replace")
        raise (PanicError if rc == 2 else FrizbeeError)(rc, msg)


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode("utf-8")


def pack(haystacks):
    """list[str|bytes] -> (uint8 array of the concatenated bytes, uint64 exclusive end offsets) - the upload format."""
    bs = [_b(h) for h in haystacks]
    ends = np.cumsum(np.fromiter((len(b) for b in bs), dtype=np.uint64, count=len(bs)), dtype=np.uint64) if bs else np.zeros(0, np.uint64)
    data = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
    return data, ends


def _take(out, n, copy=True):
    """Result list of the C ABI -> numpy.  copy=False wraps the library's (pinned, pooled) buffer without copying and
    returns it to the pool when the array is garbage collected."""
    if copy or not n.value:
        arr = np.zeros(n.value, MATCH_DTYPE)
        if n.value:
            C.memmove(arr.ctypes.data, out, n.value * 8)
        lib().fzb_matches_free(out)
        return arr
    base = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n.value * 8,))
    weakref.finalize(base, lib().fzb_matches_free, C.c_void_p(out.value))
    return base.view(MATCH_DTYPE)


class Corpus:
    """A haystack list packed and resident in HBM (what `match_list(&haystacks)` borrows, uploaded once)."""

    def __init__(self, haystacks=None, *, packed=None):
        data, ends = packed if packed is not None else pack(haystacks)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        self.h = C.c_void_p()
        self._keep = None
        _check(lib().fzb_corpus_upload(data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), C.byref(self.h)))

    @classmethod
    def from_device(cls, dev_bytes_ptr, dev_ends_ptr, n, total_bytes, ends_are_u64=False, keep=None, max_len=0, uniform_len=0):
        """Borrow device memory already in the padded-16 layout (see include/frizbee_hip.h).  max_len: optional upper bound
        on the haystack length (0 = unknown).  uniform_len: optional promise that EVERY haystack has exactly this many bytes
        (the hot kernels then do not read the end offsets; Corpus(...) / fzb_corpus_upload detect it themselves)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        self._keep = keep
        _check(lib().fzb_corpus_from_device(dev_bytes_ptr, dev_ends_ptr, int(ends_are_u64), n, total_bytes, C.byref(self.h)))
        # (both promises are verified on the device against the end offsets when they are made; the uniform length first: it implies its bound)
        if uniform_len:
            _check(lib().fzb_corpus_set_uniform_len(self.h, uniform_len))
        if max_len:  # (beside a uniform length a looser bound is accepted and ignored, a tighter one refused)
            _check(lib().fzb_corpus_set_max_len(self.h, max_len))
        return self

    def build_view(self):
        """fzb_corpus_build_view: the streaming filter's interleaved view for a borrowed ragged corpus (an uploaded one has it already).
        Returns True when the corpus has a view afterwards."""
        built = C.c_int()
        _check(lib().fzb_corpus_build_view(self.h, C.byref(built)))
        return bool(built.value)

    def __len__(self):
        return lib().fzb_corpus_len(self.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().fzb_corpus_free(self.h)
                self.h = None
        except Exception:
            pass


SHARD_BY_COUNT, SHARD_BY_BYTES, SHARD_OVERSUBSCRIBE = 0, 1, 2  # include/frizbee_hip.h FZB_SHARD_*


def device_count():
    n = C.c_int()
    _check(lib().fzb_device_count(C.byref(n)))
    return n.value


def shard_ranges(ends, nshards, by_bytes=False):
    """fzb_shard_ranges: the contiguous index ranges [(lo, hi)] the sharded upload cuts a list into (host arithmetic, no device)."""
    ends = np.ascontiguousarray(ends, dtype=np.uint64)
    out = np.zeros(nshards + 1, np.uint64)
    _check(lib().fzb_shard_ranges(ends.ctypes.data if len(ends) else None, len(ends), nshards, int(by_bytes), out.ctypes.data))
    return [(int(out[g]), int(out[g + 1])) for g in range(nshards)]


class ShardedCorpus:
    """A haystack list cut into contiguous shards, shard g resident on device g (fzb_corpus_upload_sharded): the multi-device form of
    `match_list_parallel`'s chunks (src/matcher/parallel.rs:55-63).  Everything below it is the C ABI - no torch, no process group."""

    def __init__(self, haystacks=None, ndev=1, *, packed=None, by_bytes=False, oversubscribe=False):
        data, ends = packed if packed is not None else pack(haystacks)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        ends = np.ascontiguousarray(ends, dtype=np.uint64)
        self.h = C.c_void_p()
        flags = (SHARD_BY_BYTES if by_bytes else 0) | (SHARD_OVERSUBSCRIBE if oversubscribe else 0)
        _check(lib().fzb_corpus_upload_sharded(data.ctypes.data, ends.ctypes.data if len(ends) else None, len(ends), ndev, flags, C.byref(self.h)))

    def __len__(self):
        return lib().fzb_sharded_corpus_len(self.h)

    def shards(self):
        """[(lo, hi, device)] per shard"""
        out = []
        for g in range(lib().fzb_sharded_corpus_shards(self.h)):
            lo, hi, dev = C.c_uint64(), C.c_uint64(), C.c_int()
            _check(lib().fzb_sharded_corpus_shard(self.h, g, C.byref(lo), C.byref(hi), C.byref(dev)))
            out.append((lo.value, hi.value, dev.value))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().fzb_sharded_corpus_free(self.h)
                self.h = None
        except Exception:
            pass


def _c_config(config):
    c = _CConfig()
    c.max_typos = -1 if config.max_typos is None else int(config.max_typos)
    c.casing, c.unicode, c.sort = int(config.casing), int(config.unicode), int(config.sort)
    for name, v in zip([f[0] for f in _CScoring._fields_], config.scoring.as_list()):
        setattr(c.scoring, name, v)
    c.pf_lanes, c.sw_lanes = config.pf_lanes, config.sw_lanes
    c.matching = int(config.matching)
    return c


@dataclass
class Pattern:
    """Reference `Pattern` + `PatternConfig` (src/pattern.rs:9-18, 230-262), fuzzy matching only.  `None` fields inherit the
    matcher's config; max_typos=k is `Some(k)` (there is no way to ask for unlimited typos per pattern, as in the reference)."""
    needle: "str | bytes"
    negated: bool = False
    max_typos: "int | None" = None
    casing: "CaseMatching | None" = None
    unicode: "UnicodeMatching | None" = None
    scoring: "Scoring | None" = None
    matching: "Matching | None" = None


def parse_query(query):
    """`Pattern::parse_query` (src/pattern.rs:186-222) -> list[Pattern]"""
    q = _b(query)
    arr, n = C.POINTER(_CPattern)(), C.c_size_t()
    _check(lib().fzb_parse_query(q, len(q), C.byref(arr), C.byref(n)))
    out = [Pattern(C.string_at(arr[i].needle_utf8, arr[i].needle_len).decode("utf-8"), negated=bool(arr[i].negated),
                   matching=None if arr[i].matching < 0 else Matching(arr[i].matching)) for i in range(n.value)]
    lib().fzb_patterns_free(arr, n.value)
    return out


class _IterApi:
    """The per-item side of the reference's interface (`match_iter`, `match_one`, `match_iter_indices`, `match_one_indices`,
    src/matcher/mod.rs:277-371), served by ONE batched device pass in list order - there is no per-item CPU path."""

    def match_iter(self, haystacks):
        """`Matcher::match_iter` (src/matcher/mod.rs:290-301): the matches in haystack order, whatever `config.sort` says"""
        return iter(self.match_list_into(haystacks))

    def match_one(self, haystack, index=0):
        """`Matcher::match_one(haystack, index)` (src/matcher/mod.rs:341-349) -> one record of MATCH_DTYPE or None"""
        r = self.match_list_into([haystack], index_offset=index)
        return r[0] if len(r) else None

    def match_iter_indices(self, haystacks):
        """`Matcher::match_iter_indices` (src/matcher/mod.rs:321-334)"""
        return iter(self._indices_into(haystacks, 0))

    def match_one_indices(self, haystack, index=0):
        """`Matcher::match_one_indices` (src/matcher/mod.rs:357-371)"""
        r = self._indices_into([haystack], index)
        return r[0] if r else None


def fuzzy_match(haystacks, needle, config=None):
    """`iter::FuzzyMatchExt::fuzzy_match` (src/matcher/iter.rs:35-78) = `Matcher::new(needle, config).match_iter(haystacks)`"""
    return Matcher(needle, config).match_iter(haystacks)


def fuzzy_match_indices(haystacks, needle, config=None):
    """`iter::FuzzyMatchExt::fuzzy_match_indices` (src/matcher/iter.rs:80-126)"""
    return Matcher(needle, config).match_iter_indices(haystacks)


class MultiMatcher(_IterApi):
    """`Matcher::from_patterns(&patterns, &config)` (src/matcher/mod.rs:95-111; composition src/matcher/multi.rs:84-152)."""

    def __init__(self, patterns, config=None):
        self.config = config or Config()
        pats = [p if isinstance(p, Pattern) else Pattern(p) for p in patterns]
        arr = (_CPattern * max(len(pats), 1))()
        self._keep = []
        for i, p in enumerate(pats):
            n = _b(p.needle)
            self._keep.append(n)
            arr[i].needle_utf8, arr[i].needle_len, arr[i].negated = C.cast(C.c_char_p(n), C.c_void_p), len(n), int(p.negated)
            arr[i].has_max_typos, arr[i].max_typos = int(p.max_typos is not None), int(p.max_typos or 0)
            arr[i].casing = -1 if p.casing is None else int(p.casing)
            arr[i].unicode = -1 if p.unicode is None else int(p.unicode)
            arr[i].has_scoring = int(p.scoring is not None)
            arr[i].matching = -1 if p.matching is None else int(p.matching)
            for name, v in zip([f[0] for f in _CScoring._fields_], (p.scoring or Scoring()).as_list()):
                setattr(arr[i].scoring, name, v)
        c = _c_config(self.config)
        self.h = C.c_void_p()
        _check(lib().fzb_multi_matcher_create(C.byref(c), arr, len(pats), C.byref(self.h)))

    def __len__(self):
        return lib().fzb_multi_matcher_len(self.h)

    def match_list(self, haystacks, copy=True):
        cp = haystacks if isinstance(haystacks, Corpus) else Corpus(haystacks)
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_multi_match_list(self.h, cp.h, C.byref(out), C.byref(n)))
        return _take(out, n, copy)

    def match_list_indices(self, haystacks, selection=None):
        """`Matcher::match_list_indices` over the compiled patterns (`match_one_indices_multi`, src/matcher/multi.rs:56-82); see
        `Matcher.match_list_indices` for `selection`."""
        cp = haystacks if isinstance(haystacks, Corpus) else Corpus(haystacks)
        return _match_list_indices(lib().fzb_multi_match_list_indices, self.h, cp, selection)

    def _indices_into(self, haystacks, index_offset):
        cp = haystacks if isinstance(haystacks, Corpus) else Corpus(haystacks)
        return _match_list_indices(lib().fzb_multi_match_list_indices_into, self.h, cp, None, index_offset)

    def match_list_into(self, haystacks, first=0, count=None, index_offset=0):
        """`Matcher::match_list_into` over the compiled patterns (src/matcher/mod.rs:373-392): unsorted, input order"""
        cp = haystacks if isinstance(haystacks, Corpus) else Corpus(haystacks)
        count = len(cp) - first if count is None else count
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_multi_match_list_into(self.h, cp.h, first, count, index_offset, C.byref(out), C.byref(n)))
        return _take(out, n)

    def match_list_device(self, corpus, dev_out_ptr, capacity, dev_count_ptr, stream=0, first=0, count=None, index_offset=0):
        count = len(corpus) - first if count is None else count
        _check(lib().fzb_multi_match_list_device(self.h, corpus.h, first, count, index_offset, dev_out_ptr, capacity, dev_count_ptr, stream))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().fzb_multi_matcher_free(self.h)
                self.h = None
        except Exception:
            pass


def _match_list_indices(fn, handle, cp, selection, index_offset=None):
    sel = None if selection is None else np.ascontiguousarray(selection, dtype=np.uint32)
    if sel is not None and len(sel) == 0:
        return []
    out, n, pos = C.c_void_p(), C.c_size_t(), C.c_void_p()
    args = (handle, cp.h, sel.ctypes.data if sel is not None else None, 0 if sel is None else len(sel)) + (() if index_offset is None else (index_offset,))
    _check(fn(*args, C.byref(out), C.byref(n), C.byref(pos)))
    try:
        recs = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n.value, 1) * 16,))[: n.value * 16].view(MATCH_INDICES_DTYPE).copy()
        total = int((recs["positions_begin"].astype(np.int64) + recs["positions_len"]).max()) if len(recs) else 0
        flat = np.ctypeslib.as_array(C.cast(pos, C.POINTER(C.c_uint32)), shape=(max(total, 1),)).copy()
    finally:
        lib().fzb_match_indices_free(out, pos)
    return [MatchIndices(int(r["index"]), int(r["score"]), bool(r["exact"]), flat[int(r["positions_begin"]) : int(r["positions_begin"]) + int(r["positions_len"])].tolist())
            for r in recs]


class MatchIndices:
    """`frizbee::MatchIndices` (src/lib.rs:189-199)"""

    __slots__ = ("index", "score", "exact", "indices")

    def __init__(self, index, score, exact, indices):
        self.index, self.score, self.exact, self.indices = index, score, exact, indices

    def __eq__(self, o):
        if not isinstance(o, MatchIndices):
            return NotImplemented
        return (self.index, self.score, self.exact, self.indices) == (o.index, o.score, o.exact, o.indices)

    def __repr__(self):
        return f"MatchIndices(index={self.index}, score={self.score}, exact={self.exact}, indices={self.indices})"


class Matcher(_IterApi):
    """`frizbee::Matcher` for one (non-negated, fuzzy) pattern: `Matcher::new(needle, &config)` (src/matcher/mod.rs:90-92)."""

    def __init__(self, needle, config=None):
        self.config = config or Config()
        c = _c_config(self.config)
        n = _b(needle)
        self.needle = n
        self.h = C.c_void_p()
        _check(lib().fzb_matcher_create(C.byref(c), n, len(n), C.byref(self.h)))

    def set_pattern(self, needle):
        """`Matcher::set_pattern` (src/matcher/mod.rs:154-165): same matcher, new needle; the device workspace is kept."""
        n = _b(needle)
        _check(lib().fzb_matcher_set_pattern(self.h, n, len(n)))
        self.needle = n

    def set_config(self, config):
        """`Matcher::set_config` (src/matcher/mod.rs:143-152)"""
        c = _c_config(config)
        _check(lib().fzb_matcher_set_config(self.h, C.byref(c)))
        self.config = config

    def info(self):
        out = (C.c_int32 * 6)()
        _check(lib().fzb_matcher_info(self.h, out))
        return dict(pf_lanes=out[0], sw_lanes=out[1], use_u8=bool(out[2]), case_sensitive=bool(out[3]), unicode=bool(out[4]), rows=out[5])

    def _corpus(self, haystacks):
        return haystacks if isinstance(haystacks, Corpus) else Corpus(haystacks)

    def match_list(self, haystacks, copy=True):
        """`Matcher::match_list` (src/matcher/mod.rs:212-222). `haystacks` is a list of str/bytes or a resident `Corpus`."""
        cp = self._corpus(haystacks)
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_match_list(self.h, cp.h, C.byref(out), C.byref(n)))
        return _take(out, n, copy)

    def match_list_parallel(self, haystacks, threads):
        """`Matcher::match_list_parallel` (src/matcher/parallel.rs:18-89); identical result for every thread count."""
        cp = self._corpus(haystacks)
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_match_list_parallel(self.h, cp.h, threads, C.byref(out), C.byref(n)))
        return _take(out, n)

    def match_list_parallel_sharded(self, sharded, copy=True):
        """`Matcher::match_list_parallel` with one DEVICE per worker (fzb_match_list_parallel_sharded): per-shard pipeline, the runs
        gathered device to device on the root and ordered there once (src/matcher/parallel.rs:66-87's result).  Equals `match_list`
        on the unsharded list."""
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_match_list_parallel_sharded(self.h, sharded.h, C.byref(out), C.byref(n)))
        return _take(out, n, copy)

    def shard_report(self):
        """How the runs of the last `match_list_parallel_sharded` reached the root (fzb_matcher_shard_report): gather form and, per shard,
        same device / peer access enabled / peer access refused (the runtime then stages the copy through host memory)."""
        return lib().fzb_matcher_shard_report(self.h).decode()

    @staticmethod
    def merge_args(run_ptrs, count_ptrs, run_caps):
        """Marshal the arguments of `merge_shard_runs` once (callers that merge the same buffers every step)."""
        n = len(run_ptrs)
        return ((C.c_void_p * max(n, 1))(*[int(p) for p in run_ptrs]), (C.c_void_p * max(n, 1))(*[int(p) for p in count_ptrs]),
                (C.c_size_t * max(n, 1))(*[int(c) for c in run_caps]), n)

    def merge_shard_runs(self, runs, cnts, caps, n=None, stream=0, copy=True):
        """fzb_merge_shard_runs: index-ordered per-shard runs resident on the current device (ascending shard order; each count pointer
        = the (records written, matches found) pair fzb_match_list_device wrote, in device memory) -> `match_list`'s ordered result on the
        host.  Concatenation + reverse / stable radix sort on the device.  Arguments: lists of addresses, or `merge_args(...)` unpacked."""
        if n is None:
            runs, cnts, caps, n = self.merge_args(runs, cnts, caps)
        out, ln = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_merge_shard_runs(self.h, runs, cnts, caps, n, stream, C.byref(out), C.byref(ln)))
        return _take(out, ln, copy)

    def match_list_indices(self, haystacks, selection=None):
        """`Matcher::match_list_indices` (src/matcher/mod.rs:234-275): list of `MatchIndices` (src/lib.rs:189-199), the matched byte
        positions in reverse order.  `selection` (corpus indices) plays the role of the haystack list - typically the top of a
        `match_list` result over a resident `Corpus`; `index` then numbers the selection."""
        return _match_list_indices(lib().fzb_match_list_indices, self.h, self._corpus(haystacks), selection)

    def _indices_into(self, haystacks, index_offset):
        return _match_list_indices(lib().fzb_match_list_indices_into, self.h, self._corpus(haystacks), None, index_offset)

    def match_list_into(self, haystacks, first=0, count=None, index_offset=0):
        """`Specialized::match_list(haystacks, haystack_index_offset, &mut matches)` (src/matcher/algo.rs:78-103): unsorted, input order."""
        cp = self._corpus(haystacks)
        count = len(cp) - first if count is None else count
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_match_list_into(self.h, cp.h, first, count, index_offset, C.byref(out), C.byref(n)))
        return _take(out, n)

    def match_list_device(self, corpus, dev_out_ptr, capacity, dev_count_ptr, stream=0, first=0, count=None, index_offset=0):
        """Device-resident form: records + count stay in HBM, asynchronous on `stream` (a hipStream_t handle)."""
        count = len(corpus) - first if count is None else count
        _check(lib().fzb_match_list_device(self.h, corpus.h, first, count, index_offset, dev_out_ptr, capacity, dev_count_ptr, stream))

    def match_list_sorted_device(self, corpus, dev_out_ptr, capacity, dev_count_ptr, stream=0):
        """`match_list` with the result left in HBM, already in `config.sort` order (device-side reverse + radix sort)."""
        _check(lib().fzb_match_list_sorted_device(self.h, corpus.h, dev_out_ptr, capacity, dev_count_ptr, stream))

    def reserve(self, corpus):
        """Allocate every device buffer queries over `corpus` can need now, so that no later query (also after set_pattern) allocates."""
        _check(lib().fzb_matcher_reserve(self.h, corpus.h))

    def set_profiling(self, on=True):
        _check(lib().fzb_set_profiling(self.h, int(on)))

    def last_timings_ms(self):
        out = (C.c_float * 4)()
        _check(lib().fzb_last_timings(self.h, out))
        return dict(filter=out[0], total=out[1], calls=int(out[2]))

    def last_stage_timings_ms(self):
        """HIP-event averages of the profiled calls, by stage (filter kernel / compaction + lane-exact prefilter / scorers / whole pipeline)"""
        out = (C.c_float * 6)()
        _check(lib().fzb_last_stage_timings(self.h, out))
        return dict(filter=out[0], compaction_and_window=out[1], scorers=out[2], total=out[3], calls=int(out[4]))

    def last_counters(self):
        out = (C.c_uint32 * 4)()
        _check(lib().fzb_last_counters(self.h, out))
        return dict(filter_survivors=out[0], kept_by_exact_prefilter=out[1], generic_scored=out[2], multi_chunk_scored=out[3])

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().fzb_matcher_free(self.h)
                self.h = None
        except Exception:
            pass


def radix_sort_matches(matches):
    """`radix_sort_matches(&mut [Match])` (src/sort.rs:6-40): stable, descending score. Returns a sorted copy."""
    a = np.ascontiguousarray(matches.copy())
    lib().fzb_radix_sort_matches(a.ctypes.data, len(a))
    return a


def k_merge_matches(sort, runs):
    """`k_merge_matches_by_*` (src/k_merge.rs:56-132): merge per-shard runs, each already ordered per `sort`."""
    lens = np.array([len(r) for r in runs], dtype=np.uint64)
    cat = np.ascontiguousarray(np.concatenate(runs)) if runs else np.zeros(0, MATCH_DTYPE)
    out = np.zeros(len(cat), MATCH_DTYPE)
    _check(lib().fzb_k_merge_matches(int(sort), cat.ctypes.data, lens.ctypes.data, len(runs), out.ctypes.data))
    return out
