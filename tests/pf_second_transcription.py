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
