"""A SECOND, independent transcription of the reference's ASCII Smith-Waterman scorer, written straight from the Rust text
(src/smith_waterman/algo/ascii.rs:10-158, ascii_gap.rs:11-105, backend/scalar.rs ScoreVec / MaskVec semantics, prefilter/mod.rs:49-65
case_needle) in plain Python lists, lane count and lane type as parameters.  Test infrastructure: tests/test_oracle_reference_properties.py
runs it against the C++ oracle on random inputs, so a transcription slip in either shows up as a difference."""


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


def score_haystack(needle, haystack, scoring, case_sensitive, include_prefix, lanes, bits, matrices=None):
    """scoring = (match, mismatch, gap_open, gap_extend, prefix, capitalization, matching_case, exact, delimiter); bits = 8 or 16.
    `matrices` (optional dict) receives score_matrix / match_masks as {(row, chunk): [lanes]}."""
    match_s, mismatch, gap_open, gap_extend, prefix_b, cap_b, case_b, _exact, delim_b = scoring
    M = (1 << bits) - 1
    L = lanes
    splat = lambda v: [v & M] * L  # `value as u8` for the u8 class
    add = lambda a, b: [(x + y) & M for x, y in zip(a, b)]  # wrapping_add
    subs = lambda a, b: [x - y if x > y else 0 for x, y in zip(a, b)]  # saturating_sub
    vmax = lambda a, b: [max(x, y) for x, y in zip(a, b)]
    vand = lambda a, b: [x & y for x, y in zip(a, b)]
    shift = lambda v, n, prev: prev[L - n :] + v[: L - n]  # shift_right_padded::<n>
    widen = lambda mask: [M if x else 0 for x in mask]
    zero = [0] * L

    needle_simd = case_needle(needle, case_sensitive)
    n = len(haystack)
    chunks = (n + L - 1) // L + 1
    gex = splat(gap_extend)
    gop = splat(max(gap_open - gap_extend, 0))  # saturating_sub on u16, then splat
    match_score = splat(min(match_s + mismatch, 0xFFFF))  # saturating_add on u16
    mismatch_v = splat(mismatch)
    case_v, cap_v, delim_v = splat(case_b), splat(cap_b), splat(delim_b)
    prefix_masked = [prefix_b & M] + [0] * (L - 1) if include_prefix else list(zero)
    prev_delim = [False] * L
    prev_lower = [False] * L
    max_scores = list(zero)
    S, MM = {}, {}
    get = lambda d, r, c: d.get((r, c), zero)  # row 0 and column 0 are never written

    for col in range(1, chunks):
        base = (col - 1) * L
        chunk = [haystack[base + i] if base + i < n else 0 for i in range(L)]  # load_partial zero-pads
        is_upper = [65 <= b <= 90 for b in chunk]
        is_lower = [97 <= b <= 122 for b in chunk]
        is_letter = [u or l for u, l in zip(is_upper, is_lower)]
        lower_shifted = [prev_lower[L - 1]] + is_lower[: L - 1]
        cap_masked = vand(widen([u and p for u, p in zip(is_upper, lower_shifted)]), cap_v)
        prev_lower = is_lower
        is_digit = [48 <= b <= 57 for b in chunk]
        is_delim = [not (le or d or b > 127) for le, d, b in zip(is_letter, is_digit, chunk)]
        delim_shifted = [prev_delim[L - 1]] + is_delim[: L - 1]
        delim_masked = vand(widen([p and not c for p, c in zip(delim_shifted, is_delim)]), delim_v)
        prev_delim = is_delim
        bonuses = add(add(add(delim_masked, cap_masked), prefix_masked), match_score)

        up_gap_mask = list(zero)
        prev_row = list(zero)
        row = list(zero)
        for r, (c1, c2) in enumerate(needle_simd, start=1):
            exact = [b == c1 for b in chunk]
            match_mask = widen([b == c1 or b == c2 for b in chunk])
            exact_w = widen(exact)
            diag = shift(prev_row, 1, get(S, r - 1, col - 1))
            diag = add(diag, vand(match_mask, bonuses))
            diag = subs(diag, mismatch_v)
            diag = add(diag, vand(exact_w, case_v))
            up = subs(subs(prev_row, gex), vand(up_gap_mask, gop))
            # propagate_horizontal_gaps (ascii_gap.rs): log-step scan; `gex` doubles at every step
            row = vmax(diag, up)
            adj, mm_adj = get(S, r, col - 1), get(MM, r, col - 1)
            g = list(gex)
            step = 1
            while step < L:
                shifted_row = shift(row, step, adj)
                shifted_mm = shift(match_mask, step, mm_adj)
                penalty = add(g, vand(gop, shifted_mm))
                row = vmax(row, subs(shifted_row, penalty))
                g = add(g, g)
                step *= 2
            S[(r, col)] = row
            MM[(r, col)] = match_mask
            prev_row = row
            up_gap_mask = match_mask
        max_scores = vmax(max_scores, row)
        prefix_masked = list(zero)
    if matrices is not None:
        matrices["S"], matrices["MM"], matrices["chunks"] = S, MM, chunks
    return max(max_scores)


def alignment_indices(needle_len, matrices, lanes, bits, score, max_typos=None, haystack_start_pos=0, unicode_haystack=None, on_match=None):
    """AlignmentPathIter (src/smith_waterman/alignment_iter.rs:35-181) over the matrices of the call above, collecting what
    score_haystack_indices (algo/mod.rs:49-94) collects: the haystack position of every Match step, in walk (= reverse) order."""
    S, MM, chunks = matrices["S"], matrices["MM"], matrices["chunks"]
    L = lanes
    zero = [0] * L
    cell = lambda d, r, c: d.get((r, c // L), zero)[c % L]
    search = score & ((1 << bits) - 1)
    col = None
    for chunk_idx in range(1, chunks):  # get_col_idx: first lane of the last row holding the score
        row = S.get((needle_len, chunk_idx), zero)
        if search in row:
            col = chunk_idx * L + row.index(search)
            break
    assert col is not None, "could not find max score in score matrix final row"
    out = []
    r, typos = needle_len, 0
    while True:
        if r == 0:
            break
        if max_typos is not None and typos > max_typos:
            break
        if col < L or score == 0:
            break
        hidx = col - L
        if unicode_haystack is not None and hidx < len(unicode_haystack) and unicode_haystack[hidx] & 0xC0 == 0x80:
            col -= 1
            score = cell(S, r, col)
            continue
        if cell(MM, r, col) != 0:
            (on_match or (lambda needle_idx, pos: out.append(pos)))(r - 1, hidx + haystack_start_pos)
            r -= 1
            col -= 1
            score = cell(S, r, col)
            continue
        diag, left, up = cell(S, r - 1, col - 1), cell(S, r, col - 1), cell(S, r - 1, col)
        if diag >= left and diag >= up:
            r, col, typos, score = r - 1, col - 1, typos + 1, diag
        elif left >= up:
            col, score = col - 1, left
        else:
            typos, r, score = typos + 1, r - 1, up
    return out


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


def score_haystack_unicode(needle, haystack, scoring, case_sensitive, include_prefix, lanes, bits, matrices=None):
    """src/smith_waterman/algo/unicode.rs:10-273 + unicode_gap.rs:110-236; `needle` is a str, `haystack` bytes."""
    match_s, mismatch, gap_open, gap_extend, prefix_b, cap_b, case_b, _exact, delim_b = scoring
    M = (1 << bits) - 1
    L = lanes
    splat = lambda v: [v & M] * L
    add = lambda a, b: [(x + y) & M for x, y in zip(a, b)]
    subs = lambda a, b: [x - y if x > y else 0 for x, y in zip(a, b)]
    vmax = lambda a, b: [max(x, y) for x, y in zip(a, b)]
    vand = lambda a, b: [x & y for x, y in zip(a, b)]
    shift = lambda v, n, prev: prev[L - n :] + v[: L - n]
    widen = lambda mask: [M if x else 0 for x in mask]
    zero = [0] * L

    chars = case_needle_unicode(needle, case_sensitive)
    if not chars:
        return 0
    n = len(haystack)
    chunks = (n + L - 1) // L + 1
    gex = splat(gap_extend)
    gop = splat(max(gap_open - gap_extend, 0))
    match_score = splat(min(match_s + mismatch, 0xFFFF))
    mismatch_v = splat(mismatch)
    case_v, cap_v, delim_v = splat(case_b), splat(cap_b), splat(delim_b)
    rows = len(chars)
    max_scores = list(zero)
    pending = {r: list(zero) for r in range(rows + 1)}
    prefix_masked = [prefix_b & M] + [0] * (L - 1) if include_prefix else list(zero)
    prev_delim = [False] * L
    prev_lower = [False] * L
    prev_cont_gex = list(zero)
    prev_scalar_start = list(zero)
    S, MM = {}, {}
    get = lambda d, r, c: d.get((r, c), zero)
    load = lambda start: [haystack[start + i] if start + i < n else 0 for i in range(L)]  # load_partial(ptr, start, len)

    def char_match_mask(byte_chunks, scalar_start, b):  # unicode_char_match_mask, unicode.rs:219-241
        clen = len(b)
        mask = [x == b[clen - 1] and s for x, s in zip(byte_chunks[4 - clen], scalar_start)]
        if clen > 1 and any(mask):
            for byte_idx in range(clen - 1):
                mask = [m and x == b[byte_idx] for m, x in zip(mask, byte_chunks[3 - byte_idx])]
        return mask

    for col in range(1, chunks):
        start = (col - 1) * L
        byte_chunks = [load(start + 3), load(start + 2), load(start + 1), load(start)]
        chunk = byte_chunks[3]
        valid = [i < min(max(n - start, 0), L) for i in range(L)]
        cont = [0x7F < b < 0xC0 and v for b, v in zip(chunk, valid)]
        scalar_start = [(not c) and v for c, v in zip(cont, valid)]
        scalar_start_w = widen(scalar_start)
        cont_gex = vand(widen(cont), gex)

        is_upper = [65 <= b <= 90 for b in chunk]
        is_lower = [97 <= b <= 122 for b in chunk]
        is_letter = [u or l for u, l in zip(is_upper, is_lower)]
        lower_shifted = [prev_lower[L - 1]] + is_lower[: L - 1]
        cap_masked = vand(widen([u and p for u, p in zip(is_upper, lower_shifted)]), cap_v)
        prev_lower = is_lower
        is_digit = [48 <= b <= 57 for b in chunk]
        is_delim = [not (le or d or b > 127) for le, d, b in zip(is_letter, is_digit, chunk)]
        delim_shifted = [prev_delim[L - 1]] + is_delim[: L - 1]
        delim_masked = vand(widen([p and not c for p, c in zip(delim_shifted, is_delim)]), delim_v)
        prev_delim = is_delim
        bonuses = add(add(add(delim_masked, cap_masked), prefix_masked), match_score)
        prefix_masked = list(zero)

        up_gap_mask = list(zero)
        prev_row = list(zero)
        row = list(zero)
        for r, (exact_b, flipped_b) in enumerate(chars, start=1):
            exact = char_match_mask(byte_chunks, scalar_start, exact_b)
            flipped = char_match_mask(byte_chunks, scalar_start, flipped_b)
            match_mask = widen([a or b for a, b in zip(exact, flipped)])
            exact_w = widen(exact)
            diag = shift(prev_row, 1, get(S, r - 1, col - 1))
            diag = add(diag, vand(match_mask, bonuses))
            diag = subs(diag, mismatch_v)
            diag = add(diag, vand(exact_w, case_v))
            diag = vand(diag, scalar_start_w)
            up = vand(subs(subs(prev_row, gex), vand(up_gap_mask, gop)), scalar_start_w)

            # propagate_horizontal_unicode_gaps (unicode_gap.rs:172-236): [gap step, prepare] for 1, 2, ..., L/4 then a final gap step at L/2
            row = vmax(diag, up)
            pend = list(match_mask)
            adj_row, adj_pend = get(S, r, col - 1), pending[r]
            c_gex, adj_c_gex = list(cont_gex), list(prev_cont_gex)
            end_mask, adj_end_mask = list(scalar_start_w), list(prev_scalar_start)
            total = list(gex)

            def gap_step(s):
                nonlocal row, pend
                shifted_row = shift(row, s, adj_row)
                shifted_pend = shift(pend, s, adj_pend)
                scalar_gex = subs(total, c_gex)
                crossed = vand(shifted_pend, end_mask)
                penalty = add(scalar_gex, vand(gop, crossed))
                row = vmax(row, subs(shifted_row, penalty))
                pend = vmax(pend, subs(shifted_pend, end_mask))

            s = 1
            while s < L // 2:
                gap_step(s)
                # prepare_next_unicode_gap_step
                c_gex = add(c_gex, shift(c_gex, s, adj_c_gex))
                adj_c_gex = add(adj_c_gex, shift(adj_c_gex, s, zero))
                end_mask = vmax(end_mask, shift(end_mask, s, adj_end_mask))
                adj_end_mask = vmax(adj_end_mask, shift(adj_end_mask, s, zero))
                total = add(total, total)
                s *= 2
            gap_step(s)

            S[(r, col)] = row
            MM[(r, col)] = match_mask
            pending[r] = pend
            prev_row = row
            up_gap_mask = match_mask
        max_scores = vmax(max_scores, row)
        prev_cont_gex = cont_gex
        prev_scalar_start = scalar_start_w
    if matrices is not None:
        matrices["S"], matrices["MM"], matrices["chunks"] = S, MM, chunks
    return max(max_scores)


def unicode_indices(needle, haystack, matrices, lanes, bits, score, max_typos=None, haystack_start_pos=0, case_sensitive=False):
    """score_haystack_unicode_indices (algo/mod.rs:97-152): every needle scalar's whole UTF-8 run, high byte first, once per position"""
    chars = case_needle_unicode(needle, case_sensitive)
    out, prev = [], [None]

    def on_match(needle_idx, pos):
        if prev[0] != pos:
            ln = len(chars[needle_idx][0])
            out.extend(pos + off for off in range(ln - 1, -1, -1))
            prev[0] = pos

    alignment_indices(len(chars), matrices, lanes, bits, score, max_typos, haystack_start_pos, haystack, on_match)
    return out


def match_greedy(needle, haystack, scoring, case_sensitive, include_prefix):
    """src/smith_waterman/greedy.rs:7-91 -> (score, forward positions) or None"""
    match_s, _mismatch, gap_open, gap_extend, prefix_b, cap_b, case_b, _exact, delim_b = scoring
    sat_add = lambda a, b: min(a + b, 0xFFFF)
    sat_sub = lambda a, b: max(a - b, 0)
    nd = case_needle(needle, case_sensitive)
    if len(nd) > len(haystack):
        return None
    score, indices, hi = 0, [], 0
    bonus_enabled = prev_lower = prev_delim = False
    for ni, (c, f) in enumerate(nd):
        start = hi
        found = False
        while hi <= len(haystack) - len(nd) + ni:
            h = haystack[hi]
            digit, upper, lower = 48 <= h <= 57, 65 <= h <= 90, 97 <= h <= 122
            delim = h < 128 and not (lower or upper or digit)
            if not delim:
                bonus_enabled = True
            if c != h and f != h:
                prev_delim, prev_lower = bonus_enabled and delim, lower
                hi += 1
                continue
            score = sat_add(score, match_s)
            if hi != start and ni != 0:
                gap_len = min(max(hi - start - 1, 0), 0xFFFF)
                score = sat_sub(score, sat_add(gap_open, min(gap_extend * gap_len, 0xFFFF)))
            if c == h:
                score = sat_add(score, case_b)
            if upper and prev_lower:
                score = sat_add(score, cap_b)
            if include_prefix and hi == 0:
                score = sat_add(score, prefix_b)
            if prev_delim and not delim:
                score = sat_add(score, delim_b)
            prev_delim, prev_lower = bonus_enabled and delim, lower
            indices.append(hi)
            hi += 1
            found = True
            break
        if not found:
            return None
    return score, indices
