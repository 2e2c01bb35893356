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
