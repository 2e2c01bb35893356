#!/usr/bin/env python3
"""Audit trail for the one place where the oracle (and the HIP path) follow the reference's CODE against the reference's own
PROPERTY TEST: prints, statement by statement, what `match_haystack_1_typo` (/root/reference/src/prefilter/algo/ascii_typos.rs:15-110)
does on an input at a given lane count, so that "the reference itself rejects here at 32 lanes" can be checked by reading the Rust
next to this trace.  Usage: trace_one_typo.py [needle] [haystack]   (defaults: the first pinned input of
tests/test_oracle_reference_properties.py LCS_DEVIATIONS_1_TYPO)"""
import sys

NEEDLE = "aa_CB-bA A"
HAY = "b/-cbcCCc0_C_ cAAc- Ac/_0a0b_b _CBaB0c a_cAbC0A B-CA0c-//aBcbCa0"


def occ(chunk, c):  # B::occ: lanes equal to the needle byte in either case (case-insensitive trace)
    lo, up = c.lower(), c.upper()
    return sum(1 << i for i, h in enumerate(chunk) if h == lo or h == up)


def bits(m, width):
    return "".join("1" if (m >> i) & 1 else "." for i in range(width))


def trace(needle, hay, lanes, out):
    n, ln = len(needle), len(hay)
    p1, p2, start_pos = 0, 1, None  # :25-27 first_path_needle_idx, second_path_needle_idx, match_start_pos
    P = lambda s: out.append(s)
    P(f"--- LANES = {lanes}: needle {needle!r} ({n} bytes), haystack {ln} bytes, {-(-ln // lanes)} chunk(s)")
    for start in range(0, ln, lanes):  # :29
        chunk = hay[start:start + lanes]
        full = (1 << len(chunk)) - 1   # :31 load_window: chunk_mask = the valid lanes
        m1, m2 = occ(chunk, needle[p1]), occ(chunk, needle[p2])  # :32-35
        c1 = c2 = full                 # :37-38
        P(f"chunk @{start}: {chunk!r}")
        it = 0
        while True:                    # :40
            it += 1
            adv = False
            cand = p1 + 1              # :43
            if cand > p2:              # :44
                if cand == n:          # :47
                    P(f"  it{it}: path1 holds all but the last needle byte -> FOUND (:47-49)")
                    return True
                p2, c2 = cand, c1      # :53-54: path 2 := path 1 with needle[{p1}] skipped
                m2 = occ(chunk, needle[p2])
                P(f"  it{it}: path1 (idx {p1}) passed path2 -> path2 := idx {p2} from path1's position (:51-56)")
            elif cand == p2 and c1 > c2:   # :57-61
                c2 = c1
                P(f"  it{it}: path1 (idx {p1}) is right behind path2 (idx {p2}) and earlier in the chunk -> path2 restarts from path1's position (:57-61)")
            h1 = m1 & c1               # :64
            if h1:                     # :65
                pos = start + (h1 & -h1).bit_length() - 1
                start_pos = pos if start_pos is None else min(start_pos, pos)
                P(f"  it{it}: path1 takes needle[{p1}]={needle[p1]!r} at {pos} (:64-76)")
                p1 += 1
                c1 &= ~((h1 & -h1) * 2 - 1)    # clear_through_lowest
                m1 = occ(chunk, needle[p1]) if p1 < n else 0
                adv = True
            else:
                P(f"  it{it}: path1 finds no needle[{p1}]={needle[p1]!r} in the rest of this chunk  [{bits(m1 & full, len(chunk))} & {bits(c1, len(chunk))}]")
            h2 = m2 & c2               # :80
            if h2:                     # :81
                pos = start + (h2 & -h2).bit_length() - 1
                start_pos = pos if start_pos is None else min(start_pos, pos)
                P(f"  it{it}: path2 takes needle[{p2}]={needle[p2]!r} at {pos} (:80-95)")
                p2 += 1
                if p2 >= n:            # :86
                    P(f"  it{it}: path2 has consumed the needle with one byte skipped -> FOUND (:86-88)")
                    return True
                c2 &= ~((h2 & -h2) * 2 - 1)
                m2 = occ(chunk, needle[p2])
                adv = True
            else:
                P(f"  it{it}: path2 finds no needle[{p2}]={needle[p2]!r} in the rest of this chunk")
            if not adv:                # :98-100
                P(f"  -> nothing advanced: next chunk with path1 at idx {p1}, path2 at idx {p2}")
                break
    P(f"end of haystack: path1 idx {p1}, path2 idx {p2} of {n} -> NOT FOUND (:103-109)")
    return False


def lcs(a, b):
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b):
            cur.append(prev[j] + 1 if x.lower() == y.lower() else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[-1]


if __name__ == "__main__":
    needle = sys.argv[1] if len(sys.argv) > 1 else NEEDLE
    hay = sys.argv[2] if len(sys.argv) > 2 else HAY
    out = []
    out.append(f"LCS(needle, haystack) = {lcs(needle, hay)} of {len(needle)}: with max_typos = 1 the reference's test oracle (src/prefilter/mod.rs:1013-1084) expects ACCEPT iff LCS + 1 >= {len(needle)}")
    for lanes in (64, 32, 16):
        r = trace(needle, hay, lanes, out)
        out.append(f"=> {lanes} lanes: {'accept' if r else 'REJECT'}")
    print("\n".join(out))
