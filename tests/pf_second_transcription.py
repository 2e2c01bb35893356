"""A SECOND, independent transcription of the reference's ASCII prefilter family (src/prefilter/algo/ascii.rs:6-72,
ascii_typos.rs:6-398), bit masks as Python ints, lane count as a parameter.  Returns the reference's (matched, start, end)
window.  Test infrastructure: tests/test_oracle_reference_properties.py runs it against the C++ oracle on random inputs."""


def case_needle(needle, case_sensitive):  # src/prefilter/mod.rs:49-65
    out = []
    for c in needle:
        if case_sensitive:
            out.append((c, c))
        elif 97 <= c <= 122:
            out.append((c, c - 32))
        elif 65 <= c <= 90:
            out.append((c, c + 32))
        else:
            out.append((c, c))
    return out


class _K:
    def __init__(self, lanes):
        self.L = lanes

    def load_window(self, hay, start):  # load.rs:4-26: (chunk, valid-lane mask); bytes past the end never reach a result
        chunk = list(hay[start : start + self.L])
        mask = (1 << len(chunk)) - 1
        return chunk + [0] * (self.L - len(chunk)), mask

    @staticmethod
    def occ(chunk, pair):
        m = 0
        for i, b in enumerate(chunk):
            if b == pair[0] or b == pair[1]:
                m |= 1 << i
        return m

    @staticmethod
    def clear_through_lowest(mask, matches):
        low = matches & -matches
        return mask & ~((low << 1) - 1)

    @staticmethod
    def tz(m):
        return (m & -m).bit_length() - 1

    def lz(self, m):  # leading zeros of an L-bit mask
        return self.L - m.bit_length()


def find_last_char_pos(k, pair, hay):  # ascii.rs:57-72
    ln = len(hay)
    start = max(ln - k.L, 0)
    while True:
        chunk, cm = k.load_window(hay, start)
        m = k.occ(chunk, pair) & cm
        if m:
            return start + k.L - k.lz(m)
        start = max(start - k.L, 0)


def match_haystack(needle, hay, lanes, case_sensitive):  # ascii.rs:6-54
    k = _K(lanes)
    nd = case_needle(needle, case_sensitive)
    ln = len(hay)
    if ln == 0:
        return (False, 0, 0)
    can_skip, match_start = True, 0
    it = iter(nd)
    needle_char = next(it)
    start = 0
    while start < ln:
        chunk, chunk_mask = k.load_window(hay, start)
        while True:
            mask = k.occ(chunk, needle_char) & chunk_mask
            if not mask:
                break
            chunk_mask = k.clear_through_lowest(chunk_mask, mask)
            if can_skip:
                match_start = start + k.tz(mask)
                can_skip = False
            nxt = next(it, None)
            if nxt is not None:
                needle_char = nxt
            elif start + lanes >= ln:
                return (True, match_start, start + lanes - k.lz(mask))
            else:
                return (True, match_start, start + find_last_char_pos(k, nd[-1], hay[start:]))
        start += lanes
    return (False, match_start, ln)


def find_end_pos_with_typos(k, nd, hay, max_typos):  # ascii_typos.rs:374-398
    ln = len(hay)
    first = len(nd) - 1 - max_typos
    start = (ln - 1) // k.L * k.L
    while True:
        chunk, cm = k.load_window(hay, start)
        m = 0
        for pair in nd[first:]:
            m |= k.occ(chunk, pair)
        m &= cm
        if m:
            return start + k.L - k.lz(m)
        if start == 0:
            break
        start -= k.L
    return ln


def _fixed_paths(needle, hay, lanes, case_sensitive, npaths):
    """match_haystack_1_typo (npaths = 2, ascii_typos.rs:14-118) and match_haystack_2_typos (npaths = 3, :120-264): path p allows p typos"""
    k = _K(lanes)
    nd = case_needle(needle, case_sensitive)
    typos = npaths - 1
    n, ln = len(nd), len(hay)
    if n <= typos:
        return (True, 0, ln)
    if ln == 0:
        return (False, 0, 0)
    idx = list(range(npaths))
    match_start = None
    found = lambda: (True, match_start, find_end_pos_with_typos(k, nd, hay, typos))
    for start in range(0, ln, lanes):
        chunk, chunk_mask = k.load_window(hay, start)
        masks = [k.occ(chunk, nd[i]) for i in idx]
        cms = [chunk_mask] * npaths
        while True:
            advanced = False
            for p in range(1, npaths):  # the catch-up blocks, in path order
                cand = idx[p - 1] + 1
                if cand > idx[p]:
                    if cand == n:
                        return found()
                    idx[p], cms[p] = cand, cms[p - 1]
                    masks[p] = k.occ(chunk, nd[idx[p]])
                elif cand == idx[p] and cms[p - 1] > cms[p]:
                    cms[p] = cms[p - 1]
            for p in range(npaths):  # then every path tries to take its next needle byte
                hits = masks[p] & cms[p]
                if hits:
                    pos = start + k.tz(hits)
                    match_start = pos if match_start is None else min(match_start, pos)
                    idx[p] += 1
                    if p > 0 and idx[p] >= n:
                        return found()
                    cms[p] = k.clear_through_lowest(cms[p], hits)
                    masks[p] = k.occ(chunk, nd[idx[p]])
                    advanced = True
            if not advanced:
                break
    return (False, match_start or 0, ln)


def match_haystack_1_typo(needle, hay, lanes, case_sensitive):
    return _fixed_paths(needle, hay, lanes, case_sensitive, 2)


def match_haystack_2_typos(needle, hay, lanes, case_sensitive):
    return _fixed_paths(needle, hay, lanes, case_sensitive, 3)


def match_haystack_many_typos(needle, hay, lanes, case_sensitive, max_typos):  # ascii_typos.rs:266-360
    k = _K(lanes)
    nd = case_needle(needle, case_sensitive)
    n, ln = len(nd), len(hay)
    if n <= max_typos:
        return (True, 0, ln)
    if ln == 0:
        return (False, 0, 0)
    npaths = max_typos + 1
    idx = [0] * npaths
    match_start = None
    found = lambda: (True, match_start, find_end_pos_with_typos(k, nd, hay, max_typos))
    for start in range(0, ln, lanes):
        chunk, chunk_mask = k.load_window(hay, start)
        masks = [k.occ(chunk, nd[i]) for i in idx]
        while True:
            for p in range(1, npaths):
                cand = idx[p - 1] + 1
                if cand > idx[p]:
                    if cand == n:
                        return found()
                    idx[p] = cand
                    masks[p] = k.occ(chunk, nd[cand])
            mm = 0
            for m in masks:
                mm |= m
            matches = mm & chunk_mask
            if not matches:
                break
            hit_pos = k.tz(matches)
            hit = matches & ((1 << (hit_pos + 1)) - 1)
            match_start = start + hit_pos if match_start is None else min(match_start, start + hit_pos)
            for p in range(npaths):
                if not (masks[p] & hit):
                    continue
                idx[p] += 1
                if idx[p] == n:
                    return found()
                masks[p] = k.occ(chunk, nd[idx[p]])
            chunk_mask = k.clear_through_lowest(chunk_mask, hit)
    return (False, match_start or 0, ln)


def prefilter(needle, hay, max_typos, case_sensitive, lanes):  # kernel_result, src/prefilter/mod.rs:663-676
    if max_typos == 0:
        return match_haystack(needle, hay, lanes, case_sensitive)
    if max_typos == 1:
        return match_haystack_1_typo(needle, hay, lanes, case_sensitive)
    if max_typos == 2:
        return match_haystack_2_typos(needle, hay, lanes, case_sensitive)
    return match_haystack_many_typos(needle, hay, lanes, case_sensitive, max_typos)


# ---- the unicode family (src/prefilter/algo/unicode.rs:7-277, unicode_typos.rs:6-509) -------------------------------------------
def case_needle_unicode(needle, case_sensitive):  # src/prefilter/mod.rs:70-96 -> [(utf8 bytes, flipped utf8 bytes)]
    out = []
    for c in needle:
        b = c.encode()
        flipped = None
        if not case_sensitive and c.isupper():
            low = c.lower()
            if len(low) == 1 and len(low.encode()) == len(b):
                flipped = low
        elif not case_sensitive and c.islower():
            up = c.upper()
            if len(up) == 1 and len(up.encode()) == len(b):
                flipped = up
        out.append((b, (flipped or c).encode()))
    return out


def _eq_mask(chunk, byte):
    m = 0
    for i, b in enumerate(chunk):
        if b == byte:
            m |= 1 << i
    return m


def _load_maskless(k, hay, start):  # load.rs:28-46: lanes past the end hold whatever follows; they never reach a result (0 here)
    chunk = list(hay[start : start + k.L]) if start < len(hay) else []
    return chunk + [0] * (k.L - len(chunk))


def _prefix_mask(k, start, hay, variant):  # match_unicode_char_prefix, unicode.rs:9-52: all bytes but the last, lane i = scalar starting at start + i
    m = (1 << k.L) - 1
    for j in range(len(variant) - 1):
        m &= _eq_mask(_load_maskless(k, hay, start + j), variant[j])
    return m


def _variant_mask(k, chunk, chunk_mask, start, hay, variant):  # char_variant_mask, unicode.rs:56-73
    mask = _eq_mask(chunk, variant[-1]) & chunk_mask
    if mask and len(variant) > 1:
        mask &= _prefix_mask(k, start, hay, variant)
    return mask


def unicode_char_mask(k, start, hay, ch):  # unicode.rs:75-116
    clen = len(ch[0])
    if start + clen > len(hay):
        return 0
    chunk, chunk_mask = k.load_window(hay, start + clen - 1)
    return _variant_mask(k, chunk, chunk_mask, start, hay, ch[0]) | _variant_mask(k, chunk, chunk_mask, start, hay, ch[1])


def find_last_unicode_char_pos(k, ch, hay):  # unicode.rs:221-277
    ln, clen = len(hay), len(ch[0])
    start = max(ln - (k.L + clen - 1), 0)
    while True:
        chunk, chunk_mask = k.load_window(hay, start + clen - 1)
        mask = (_eq_mask(chunk, ch[0][-1]) | _eq_mask(chunk, ch[1][-1])) & chunk_mask
        if mask and clen > 1:
            mask &= _prefix_mask(k, start, hay, ch[0]) | _prefix_mask(k, start, hay, ch[1])
        if mask:
            return start + k.L - k.lz(mask) + clen - 1
        if start == 0:
            break
        start = max(start - k.L, 0)
    return ln


def match_haystack_unicode(needle, hay, lanes, case_sensitive):  # unicode.rs:118-219
    k = _K(lanes)
    chars = case_needle_unicode(needle, case_sensitive)
    ln = len(hay)
    if ln == 0:
        return (False, 0, 0)
    can_skip, match_start = True, 0
    it = iter(chars)
    ch = next(it)
    start = 0
    ALL = (1 << lanes) - 1
    while start + len(ch[0]) <= ln:
        char_len = len(ch[0])
        chunk, valid = k.load_window(hay, start + char_len - 1)
        available = ALL
        while True:
            chunk_mask = available & valid
            mask = _variant_mask(k, chunk, chunk_mask, start, hay, ch[0]) | _variant_mask(k, chunk, chunk_mask, start, hay, ch[1])
            if not mask:
                break
            available = k.clear_through_lowest(available, mask)
            if can_skip:
                match_start = start + k.tz(mask)
                can_skip = False
            nxt = next(it, None)
            if nxt is not None:
                ch = nxt
                if len(ch[0]) != char_len:  # reload the window when the char width changes to realign the lanes
                    if start + len(ch[0]) > ln:
                        break
                    char_len = len(ch[0])
                    chunk, valid = k.load_window(hay, start + char_len - 1)
            elif start + len(ch[0]) - 1 + lanes >= ln:
                return (True, match_start, start + lanes - k.lz(mask) + len(ch[0]) - 1)
            else:
                return (True, match_start, start + find_last_unicode_char_pos(k, ch, hay[start:]))
        start += lanes
    return (False, match_start, ln)


def find_end_pos_with_unicode_typos(k, chars, hay, max_typos):  # unicode_typos.rs:483-509
    ln = len(hay)
    first = len(chars) - 1 - max_typos
    start = max(ln - k.L, 0)
    while True:
        end_pos = 0
        for ch in chars[first:]:
            mask = unicode_char_mask(k, start, hay, ch)
            if mask:
                end_pos = max(end_pos, start + k.L - k.lz(mask) + len(ch[0]) - 1)
        if end_pos:
            return end_pos
        if start == 0:
            break
        start = max(start - k.L, 0)
    return ln


def _unicode_fixed_paths(needle, hay, lanes, case_sensitive, npaths):  # unicode_typos.rs:14-139 (1 typo), :141-339 (2 typos)
    k = _K(lanes)
    chars = case_needle_unicode(needle, case_sensitive)
    typos = npaths - 1
    n, ln = len(chars), len(hay)
    if n <= typos:
        return (True, 0, ln)
    if ln == 0:
        return (False, 0, 0)
    ALL = (1 << lanes) - 1
    idx = list(range(npaths))
    match_start = None
    found = lambda: (True, match_start, find_end_pos_with_unicode_typos(k, chars, hay, typos))
    for start in range(0, ln, lanes):
        masks = [unicode_char_mask(k, start, hay, chars[i]) for i in idx]
        cms = [ALL] * npaths
        while True:
            advanced = False
            for p in range(1, npaths):
                cand = idx[p - 1] + 1
                if cand > idx[p]:
                    if cand == n:
                        return found()
                    idx[p], cms[p] = cand, cms[p - 1]
                    masks[p] = unicode_char_mask(k, start, hay, chars[idx[p]])
                elif cand == idx[p] and cms[p - 1] > cms[p]:
                    cms[p] = cms[p - 1]
            for p in range(npaths):
                hits = masks[p] & cms[p]
                if hits:
                    pos = start + k.tz(hits)
                    match_start = pos if match_start is None else min(match_start, pos)
                    idx[p] += 1
                    if p > 0 and idx[p] >= n:
                        return found()
                    cms[p] = k.clear_through_lowest(cms[p], hits)
                    masks[p] = unicode_char_mask(k, start, hay, chars[idx[p]])
                    advanced = True
            if not advanced:
                break
    return (False, match_start or 0, ln)


def match_haystack_unicode_many_typos(needle, hay, lanes, case_sensitive, max_typos):  # unicode_typos.rs:341-468
    k = _K(lanes)
    chars = case_needle_unicode(needle, case_sensitive)
    n, ln = len(chars), len(hay)
    if n <= max_typos:
        return (True, 0, ln)
    if ln == 0:
        return (False, 0, 0)
    npaths = max_typos + 1
    idx = [0] * npaths
    match_start = None
    found = lambda: (True, match_start, find_end_pos_with_unicode_typos(k, chars, hay, max_typos))
    for start in range(0, ln, lanes):
        chunk_mask = (1 << lanes) - 1
        masks = [unicode_char_mask(k, start, hay, chars[i]) for i in idx]
        while True:
            for p in range(1, npaths):
                cand = idx[p - 1] + 1
                if cand > idx[p]:
                    if cand == n:
                        return found()
                    idx[p] = cand
                    masks[p] = unicode_char_mask(k, start, hay, chars[cand])
            mm = 0
            for m in masks:
                mm |= m
            matches = mm & chunk_mask
            if not matches:
                break
            hit_pos = k.tz(matches)
            hit = matches & ((1 << (hit_pos + 1)) - 1)
            match_start = start + hit_pos if match_start is None else min(match_start, start + hit_pos)
            for p in range(npaths):
                if not (masks[p] & hit):
                    continue
                idx[p] += 1
                if idx[p] == n:
                    return found()
                masks[p] = unicode_char_mask(k, start, hay, chars[idx[p]])
            chunk_mask = k.clear_through_lowest(chunk_mask, hit)
    return (False, match_start or 0, ln)


def prefilter_unicode(needle, hay, max_typos, case_sensitive, lanes):  # kernel_result_unicode, src/prefilter/mod.rs:678-691
    if max_typos == 0:
        return match_haystack_unicode(needle, hay, lanes, case_sensitive)
    if max_typos <= 2:
        return _unicode_fixed_paths(needle, hay, lanes, case_sensitive, max_typos + 1)
    return match_haystack_unicode_many_typos(needle, hay, lanes, case_sensitive, max_typos)
