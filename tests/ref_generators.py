"""The reference's deterministic test-input generators, restated (so that its property tests can be re-run here on the same input
distribution): `ByteCursor` of src/prefilter/mod.rs:752-868, `ByteCursor` of src/smith_waterman/backend/tests/generator.rs:21-123
(shared with tests/api_properties.rs), `ApiCase::from_bytes` (tests/api_properties.rs:22-69) and its indices contract (:116-166)."""
import numpy as np


class ByteCursor:  # src/prefilter/mod.rs:752-868
    def __init__(self, data):
        self.data, self.pos = data, 0

    def next(self):
        if not self.data:
            b = (self.pos * 37 + 11) & 255
        else:
            b = (self.data[self.pos % len(self.data)] + (self.pos // len(self.data)) * 17) & 255
        self.pos += 1
        return b

    def bool(self):
        return self.next() & 1 == 1

    def usize(self):
        v = 0
        for shift in range(0, 64, 8):
            v |= self.next() << shift
        return v

    def len(self, mx, boundaries):
        if self.next() % 4 == 0:
            return min(boundaries[self.next() % len(boundaries)], mx)
        return self.usize() % (mx + 1)

    def char(self):
        b = self.next()
        k = b % 16
        if k == 0:
            return "\0"
        if k <= 7:
            return " /.,_-:"[k - 1]
        if k <= 10:
            return chr(ord("a") + b % 26)
        if k <= 13:
            return chr(ord("A") + b % 26)
        return chr(ord("0") + b % 10)

    def byte(self):
        b = self.next()
        k = b % 18
        if k == 0:
            return 0
        if k <= 7:
            return ord(" /.,_-:"[k - 1])
        if k <= 10:
            return ord("a") + b % 26
        if k <= 13:
            return ord("A") + b % 26
        if k <= 15:
            return ord("0") + b % 10
        if k == 16:
            return 0x80 | (b & 0x3F)
        return b

    def unicode_char(self):
        b = self.next()
        k = b % 12
        if k <= 7:
            return "éن다😀न _/"[k]
        if k <= 9:
            return chr(ord("a") + b % 26)
        if k == 10:
            return chr(ord("A") + b % 26)
        return chr(ord("0") + b % 10)


class SwCursor(ByteCursor):  # src/smith_waterman/backend/tests/generator.rs:21-123 (shared with tests/api_properties.rs)
    def next(self):
        if not self.data:
            b = (self.pos * 29 + 7) & 255
        else:
            b = (self.data[self.pos % len(self.data)] + (self.pos // len(self.data)) * 19) & 255
        self.pos += 1
        return b

    def byte(self):
        b = self.next()
        k = b % 16
        if k == 0:
            return ord("a")
        if k <= 7:
            return ord(" /.,_-:"[k - 1])
        if k <= 10:
            return ord("a") + b % 26
        if k <= 13:
            return ord("A") + b % 26
        return ord("0") + b % 10

    def char(self):
        b = self.next()
        if b & 0x0F == 0:
            return "éن다😀"[(b >> 4) & 3]
        k = b % 18
        if k == 0:
            return "a"
        if k <= 7:
            return " /.,_-:"[k - 1]
        if k <= 11:
            return chr(ord("a") + b % 26)
        if k <= 15:
            return chr(ord("A") + b % 26)
        return chr(ord("0") + b % 10)

    def string(self, n):
        return "".join(self.char() for _ in range(n))


def _api_case(data):  # ApiCase::from_bytes, tests/api_properties.rs:22-69
    c = SwCursor(data)
    needle_len = c.len(32, [0, 1, 2, 7, 8, 15, 16, 31, 32])
    haystack_count = c.len(32, [0, 1, 2, 7, 8, 15, 16, 31, 32])
    haystacks = [c.string(c.len(96, [0, 1, 2, 7, 8, 15, 16, 31, 32, 63, 64, 95, 96])) for _ in range(haystack_count)]
    b = c.next() % 5
    max_typos = None if b == 0 else b - 1 if b <= 3 else c.next() % 8
    casing = ["Ignore", "Smart", "Respect"][c.next() % 3]
    matching = ["Fuzzy", "Exact", "Prefix", "Suffix", "Substring"][c.next() % 5]
    sort = "ScoreThenIndexAsc" if c.bool() else "IndexAsc"
    return c.string(needle_len), haystacks, dict(max_typos=max_typos, casing=casing, matching=matching, sort=sort)


def api_cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        yield _api_case(rng.integers(0, 256, int(rng.integers(0, 4097))).tolist())


def assert_indices_contract(needle, haystacks, cfg, matches, indices):  # tests/api_properties.rs:116-166
    match_set = {int(m["index"]): (int(m["score"]), bool(m["exact"])) for m in matches}
    for index, score, exact, ix in indices:
        assert index < len(haystacks) and match_set.get(index) == (score, exact), (needle, cfg)
        h = haystacks[index].encode()
        assert all(a > b for a, b in zip(ix[:-1], ix[1:])) and len(ix) <= len(needle.encode()) and all(i < len(h) for i in ix), (needle, cfg, ix)
    if cfg["max_typos"] is None or cfg["matching"] != "Fuzzy":
        assert {i[0]: (i[1], i[2]) for i in indices} == match_set, (needle, cfg)




def _multi_case(data):  # MultiPatternCase::from_bytes, tests/api_properties.rs:251-309
    c = SwCursor(data)
    patterns = []
    for _ in range(1 + c.next() % 3):
        matching = [None, "Fuzzy", "Exact", "Prefix", "Suffix", "Substring"][min(c.next() % 6, 5)]
        negated = c.bool()
        needle_len = c.len(8, [0, 1, 2, 3, 7, 8])
        patterns.append(dict(needle=c.string(needle_len), negated=negated, matching=matching))
    haystack_count = c.len(24, [0, 1, 2, 7, 8, 15, 16, 24])
    haystacks = [c.string(c.len(48, [0, 1, 2, 7, 8, 15, 16, 31, 32, 48])) for _ in range(haystack_count)]
    max_typos = [None, 0, 1, 2][min(c.next() % 4, 3)]
    casing = ["Ignore", "Smart", "Respect"][c.next() % 3]
    matching = ["Fuzzy", "Exact", "Prefix", "Suffix", "Substring"][c.next() % 5]
    return patterns, haystacks, dict(max_typos=max_typos, casing=casing, matching=matching)


def multi_cases(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        yield _multi_case(rng.integers(0, 256, int(rng.integers(0, 4097))).tolist())
