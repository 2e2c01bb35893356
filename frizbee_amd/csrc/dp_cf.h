// Single-chunk Smith-Waterman, thread per haystack, second form: the whole DP stays in the BIASED domain and the zero-padding
// lanes of the chunk are not computed at all - their contribution to the final maximum has a closed form.
//
// Restates, bit-exactly for a single chunk, the same reference code as dp_body.h's dp_single_chunk:
//   score_haystack              src/smith_waterman/algo/ascii.rs:10-158
//   propagate_horizontal_gaps   src/smith_waterman/algo/ascii_gap.rs:11-105  (log-step scan, steps 1, 2, ..., SWL/2)
// Preconditions (checked on the host, LaunchCfg::cf_ok; otherwise dp_single_chunk runs):
//   * the needle has no NUL byte           -> a zero-padding lane never matches a needle row,
//   * gap_extend <= mismatch_penalty        -> inside the padding a step down ("up" move) is never worse than a diagonal,
//   * the biased values fit 16 bits (bias_ok).
//
// Notation: e = gap_extend, x = mismatch_penalty, o = gap_open - gap_extend, S(i, L) = the reference's row value after
// propagate_horizontal_gaps, P = 2 * REAL = number of lanes that are computed ("real" lanes: the window's bytes and, up to
// P, NUL lanes treated like any other lane).
//
// 1. Biased domain.  B(i, L) = S(i, L) + L*e.  With `a (-) b` = saturating subtract:
//      diag:  S(i-1, L-1) + match*bonus (-) x       ->  (B(i-1, L-1) + match*bonus) (-) (x - e)      [lane 0: (-) x]
//      up:    S(i-1, L) (-) e (-) o*match(i-1, L)    ->   B(i-1, L) (-) (e + o*match(i-1, L))
//      the reference's floor at 0 (its saturating subtracts) becomes one max with the lane's bias L*e,
//      gap step s:  S(L) = max(S(L), S(L-s) (-) (s*e + o*match(L-s)))  ->  B(L) = max(B(L), B(L-s) (-) o*match(L-s)).
//    (dp_body.h biases only the scan; here nothing is converted back until the final maximum.)
//
// 2. The last needle row is not propagated: every value the scan produces is an earlier lane's value minus a
//    non-negative cost, so the maximum over the lanes - the only thing read from the last row - is the maximum of the row
//    before the scan (ascii.rs:152-156 takes the horizontal max of the last row only).
//
// 3. Padding lanes (L >= P; NUL bytes that match nothing).  Their cells are reached from real cells only by
//      (a) the diagonal out of the last real lane:      S(i-1, P-1) (-) x                    into (i, P),
//      (b) a gap step s from real lane k, k + s >= P:   (B_s(i, k) (-) o*match(i, k)) - (k+s)*e   into (i, k+s),
//          B_s = the row's state just before step s,
//    and inside the padding every move only subtracts: down e (no gap-open charge: nothing matches there), diagonal x >= e,
//    right e per lane.  Padding cells never feed real cells (every move goes right or down).  So the padding's share of the
//    final maximum is  max over entries of  entry (-) e * (rows below the entry)  - one running maximum `acc` that loses e per row.
//    Of the (k, s) pairs in (b) only one per real lane matters: for k >= P - s/2 the step-s entry is dominated either by the
//    step-s/2 entry from the same lane (if step s/2 left B(k) unchanged: same value, s/2 lanes less to pay) or by the step-s
//    entry from lane k - s/2 (if step s/2 raised B(k) from there: one gap-open charge and s/2 lanes less).  What is left:
//    step s takes lanes [P-s, P-s/2) (step 1: lane P-1), each lane exactly once.
//    For the last row (b) is dominated by the lane's own value, (a) is kept.
//    And an entry (b) of row i is dominated by walking DOWN lane k itself whenever s*e >= (rows-2-i) * o: the padding route pays
//    o*match(i,k) + s*e + e per remaining row, the column pays e + o*match per remaining row and meets at most rows-2-i matching
//    cells after (i,k) (the last row's own match is never charged).  Both are wave-uniform tests, so with the default scoring
//    (e = 1, o = 4) a 6-row needle visits 5, 5, 3, 2, 0 dwords in its five propagated rows.
//    Targets k + s stay below SWL as long as P <= 3/4 SWL (static_assert).
//
// tests/test_kernel_math_host.py compiles this header for the host and fuzzes it against the oracle (all scorings, widths).
#pragma once
#include "dp_body.h"

__device__ __forceinline__ u32 p_adds(u32 a, u32 b) { return as_u32(__builtin_elementwise_add_sat(as_us2(a), as_us2(b))); }

// is real lane k the padding-entry source of gap step s?  (see 3. above)
constexpr bool cf_entry_lane(int k, int s, int P) {
    if (k < 0 || k >= P) return false;
    if (s == 1) return k == P - 1;
    const int lo = P - s > 0 ? P - s : 0, hi = P - s / 2 > 0 ? P - s / 2 : 0;
    return k >= lo && k < hi;
}

template <int REAL, bool UPPER, bool LAST>
__device__ __forceinline__ void cf_row_cells(const u32 (&hw)[REAL], const u32 (&bonus)[REAL], const u32 (&B)[REAL], const u32 (&ge)[REAL], const u32 (&biasv)[REAL],
                                             u32 orv, u32 cmpv, u32 cv, bool ci, u32 xp0, u32 xpv, u32 casev, u32 gopmv, u32 (&b)[REAL], u32 (&g)[REAL]) {
    const u32 ONE = 0x00010001u;
#pragma unroll
    for (int d = 0; d < REAL; d++) {
        const u32 sh = __builtin_amdgcn_alignbit(B[d], d ? B[d - 1] : 0u, 16);  // B(i-1, L-1)
        const u32 t = (hw[d] | orv) ^ cmpv;  // one v_bitop3_b32
        const u32 mm = p_subs(ONE, t);  // match mask as 0/1 per lane
        u32 mb = p_mul(mm, bonus[d]);
        if (UPPER) {  // literal form: separate exact-case compare (bonus[] holds no case term here)
            const u32 ex = ci ? p_subs(ONE, hw[d] ^ cv) : mm;
            mb = p_add(mb, p_mul(ex, casev));
        }
        const u32 D = p_subs(p_add(sh, mb), d == 0 ? xp0 : xpv);
        const u32 U = p_subs(B[d], ge[d]);
        if (LAST) {
            b[d] = p_subs(p_max(D, U), biasv[d]);  // unbiased, floored at 0
        } else {
            b[d] = p_max(p_max(D, U), biasv[d]);
            g[d] = p_mul(mm, gopmv);
        }
    }
}

// Scores the trimmed window (1 <= m <= 2*REAL bytes, zero padded in hb) as ONE chunk of SWL lanes of which the first
// 2*REAL are computed.  REAL == SWL/2: no padding lanes exist (1. and 2. only).
template <int SWL, bool UPPER, int REAL>
__device__ __forceinline__ u32 dp_single_chunk_cf(const NeedleDev& nd, bool include_prefix, const u8* cls, const u32 (&hb)[SWL / 4]) {
    constexpr int NW = SWL / 2;
    constexpr int P = 2 * REAL;
    constexpr bool PAD = REAL < NW;
    static_assert(REAL >= 1 && REAL <= NW, "REAL");
    static_assert(!PAD || 4 * P <= 3 * SWL, "padding entries must land inside the chunk");
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 e = nd.gex, x = nd.mismatch;
    const u32 Mv = splat16(nd.match_plus_mismatch), ev = splat16(e), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const u32 xpv = splat16(x - e), xp0 = x | ((x - e) << 16);
    u32 hw[REAL], bonus[REAL], biasv[REAL];
    {
        u32 clsw_prev = 0;
#pragma unroll
        for (int d = 0; d < REAL; d++) {
            const u32 w = hb[d / 2];
            const u32 b0 = (d & 1) ? (w >> 16) & 0xFF : w & 0xFF;
            const u32 b1 = (d & 1) ? w >> 24 : (w >> 8) & 0xFF;
            hw[d] = b0 | (b1 << 16);
            const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
            const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);  // class of lane-1 (lane -1 of chunk 0: none)
            const u32 cap01 = (clsw >> 1) & sh & ONE;                        // upper(j) & lower(j-1)
            const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;                 // delim(j-1) & !delim(j)
            u32 bn = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
            if (d == 0 && include_prefix) bn = p_add(bn, (u32)nd.prefix);  // first_lane(prefix_bonus)
            // matching-case bonus (ascii.rs:121-131) folded in: without an uppercase needle byte, a matching lane has the
            // needle byte's exact case iff it is not an uppercase letter (lowercase needle letter: h == c; non-letter: h == c)
            bonus[d] = UPPER ? bn : p_add(bn, p_mul(~(clsw >> 1) & ONE, casev));
            clsw_prev = clsw;
            biasv[d] = e * (u32)(2 * d + ((2 * d + 1) << 16));
        }
    }
    u32 B[REAL], ge[REAL];
#pragma unroll
    for (int d = 0; d < REAL; d++) B[d] = biasv[d], ge[d] = ev;
    u32 acc0 = 0, acc1 = 0;                                                    // the padding's running maximum (unbiased), two chains
    const u32 edc = 0xFFFFu | (((u32)(P - 1) * e + x) << 16);                  // (a): S(i-1, P-1) (-) x from the biased high lane
    for (u32 r = 0; r + 1 < rows; r++) {
        // needle bytes through aligned dword reads of the by-value argument: wave-uniform, so they are scalar loads
        const u32 c = (((const u32*)nd.c)[r >> 2] >> (8 * (r & 3))) & 0xFF, f = (((const u32*)nd.f)[r >> 2] >> (8 * (r & 3))) & 0xFF;
        const bool ci = c != f;  // case-folded ASCII letter: (h | 0x20) == (c | 0x20) <=> h in {c, flip(c)}
        const u32 cmpv = splat16(ci ? (c | 0x20) : c), cv = splat16(c);
        if (PAD) {
            acc0 = p_max(p_subs(acc0, ev), p_subs(B[REAL - 1], edc));
            acc1 = p_subs(acc1, ev);
        }
        const u32 lim = (rows - 2 - r) * (u32)nd.gopm;  // entries of step s matter only while s*e < lim
        u32 b[REAL], g[REAL];
        cf_row_cells<REAL, UPPER, false>(hw, bonus, B, ge, biasv, ci ? 0x00200020u : 0u, cmpv, cv, ci, xp0, xpv, casev, gopmv, b, g);
        // ---- propagate_horizontal_gaps over the real lanes; the padding entries (3b) are read off on the way ---------
        {  // step 1
            u32 cc[REAL], nb[REAL];
#pragma unroll
            for (int d = 0; d < REAL; d++) cc[d] = p_subs(b[d], g[d]);
            if (PAD && e < lim) acc1 = p_max(acc1, p_subs(cc[REAL - 1], 0xFFFFu | (((u32)P * e) << 16)));  // lane P-1 -> lane P
#pragma unroll
            for (int d = 0; d < REAL; d++) nb[d] = p_max(b[d], __builtin_amdgcn_alignbit(cc[d], d ? cc[d - 1] : 0u, 16));
#pragma unroll
            for (int d = 0; d < REAL; d++) b[d] = nb[d];
        }
#pragma unroll
        for (int s = 2; s <= SWL / 2; s *= 2) {
            const int off = s / 2;  // in dwords
            if (PAD && (u32)s * e < lim) {
#pragma unroll
                for (int d = 0; d < REAL; d++) {
                    const bool in0 = cf_entry_lane(2 * d, s, P), in1 = cf_entry_lane(2 * d + 1, s, P);
                    if (in0 || in1) {
                        const u32 k0 = in0 ? (u32)(2 * d + s) * e : 0xFFFFu, k1 = in1 ? (u32)(2 * d + 1 + s) * e : 0xFFFFu;  // target lane * e
                        const u32 v = p_subs(p_subs(b[d], g[d]), k0 | (k1 << 16));
                        if (d & 1) acc1 = p_max(acc1, v);
                        else acc0 = p_max(acc0, v);
                    }
                }
            }
            if (off < REAL) {
                u32 nb[REAL];
#pragma unroll
                for (int d = 0; d < REAL; d++) nb[d] = d >= off ? p_max(b[d], p_subs(b[d - off], g[d - off])) : b[d];
#pragma unroll
                for (int d = 0; d < REAL; d++) b[d] = nb[d];
            }
        }
#pragma unroll
        for (int d = 0; d < REAL; d++) B[d] = b[d], ge[d] = p_add(g[d], ev);
    }
    // ---- last row: cells only (2.), unbiased; its maximum joins the padding's ---------------------------------------
    u32 mx;
    {
        const u32 r = rows - 1;
        const u32 c = (((const u32*)nd.c)[r >> 2] >> (8 * (r & 3))) & 0xFF, f = (((const u32*)nd.f)[r >> 2] >> (8 * (r & 3))) & 0xFF;
        const bool ci = c != f;
        const u32 cmpv = splat16(ci ? (c | 0x20) : c), cv = splat16(c);
        if (PAD) {
            acc0 = p_max(p_subs(acc0, ev), p_subs(B[REAL - 1], edc));
            acc1 = p_subs(acc1, ev);
        }
        u32 b[REAL], g[REAL];
        cf_row_cells<REAL, UPPER, true>(hw, bonus, B, ge, biasv, ci ? 0x00200020u : 0u, cmpv, cv, ci, xp0, xpv, casev, gopmv, b, g);
        mx = p_max(acc0, acc1);
#pragma unroll
        for (int d = 0; d < REAL; d++) mx = p_max(mx, b[d]);
    }
    return max(mx & 0xFFFF, mx >> 16);
}
