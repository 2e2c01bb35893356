// Single-chunk Smith-Waterman, thread per haystack, second form: the whole DP stays in a BIASED domain and the zero-padding
// lanes of the chunk are not computed at all - their contribution to the final maximum has a closed form.
//
// Restates, bit-exactly for a single chunk, the same reference code as dp_body.h's dp_single_chunk:
//   score_haystack              src/smith_waterman/algo/ascii.rs:10-158
//   propagate_horizontal_gaps   src/smith_waterman/algo/ascii_gap.rs:11-105  (log-step scan, steps 1, 2, ..., SWL/2)
// Preconditions (checked on the host, LaunchCfg::cf_ok; otherwise dp_single_chunk runs):
//   * the needle has no NUL byte              -> a zero-padding lane never matches a needle row,
//   * 2 * gap_extend <= mismatch_penalty       -> inside the padding a step down ("up" move) is never worse than a diagonal, and the
//                                                 diagonal's constant x - 2e below is not negative,
//   * the biased values stay below 0x7C00 (cf_ok: the cell's three-way maximum is v_pk_maximum3_f16, exact on such values - p_max3_s).
//
// Notation: e = gap_extend, x = mismatch_penalty, o = gap_open - gap_extend, S(i, L) = the reference's row value after
// propagate_horizontal_gaps (row -1 = the zero row), P = 2 * REAL = number of lanes that are computed ("real" lanes: the window's
// bytes and, up to P, NUL lanes treated like any other lane), `a (-) b` = saturating subtract.
//
// 1. Biased domain.  T(i, L) = S(i, L) + (L + i + 1) * e : one e per lane AND one per row.  Then
//      diag:  S(i-1, L-1) + match*bonus (-) x      ->  (T(i-1, L-1) + match*bonus) (-) (x - 2e)     [virtual lane -1: T = (i-1)*e]
//      up:    S(i-1, L) (-) e (-) o*match(i-1, L)   ->   T(i-1, L) (-) o*match(i-1, L)               (the e is in the row bias)
//      the reference's floor at 0 (its saturating subtracts) becomes one max with the cell's bias (L + i + 1) * e,
//      gap step s:  S(L) = max(S(L), S(L-s) (-) (s*e + o*match(L-s)))  ->  T(L) = max(T(L), T(L-s) (-) o*match(L-s)).
//    Row 0 is peeled (its predecessor is the zero row: no up move, diag = match*bonus (-) x).
//    (dp_body.h biases only the scan; here nothing is converted back until the final maximum.)
//
// 2. The last needle row is not propagated: every value the scan produces is an earlier lane's value minus a
//    non-negative cost, so the maximum over the lanes - the only thing read from the last row - is the maximum of the row
//    before the scan (ascii.rs:152-156 takes the horizontal max of the last row only).
//
// 3. Padding lanes (L >= P; NUL bytes that match nothing).  Their cells are reached from real cells only by
//      (a) the diagonal out of the last real lane:      S(i-1, P-1) (-) x                          into (i, P),
//      (b) a gap step s from real lane k, k + s >= P:   S_s(i, k) (-) o*match(i, k) (-) s*e         into (i, k+s),
//          S_s = the row's state just before step s,
//    and inside the padding every move only subtracts: down e (no gap-open charge: nothing matches there), diagonal x >= e,
//    right e per lane.  Padding cells never feed real cells (every move goes right or down).  So the padding's share of the
//    final maximum is  max over entries of  entry (-) e * (rows below the entry): one running maximum, kept as
//    A = value + (i + 1) * e so that it needs no per-row decay ((b) in that domain is T_s(k) (-) o*match (-) (k+s)*e).
//    Of the (k, s) pairs in (b) only one per real lane matters: for k >= P - s/2 the step-s entry is dominated either by the
//    step-s/2 entry from the same lane (if step s/2 left lane k unchanged: same value, s/2 lanes less to pay) or by the step-s
//    entry from lane k - s/2 (if step s/2 raised lane k from there: one gap-open charge and s/2 lanes less).  What is left:
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

// is real lane k the padding-entry source of gap step s?  (see 3. above)
constexpr bool cf_entry_lane(int k, int s, int P) {
    if (k < 0 || k >= P) return false;
    if (s == 1) return k == P - 1;
    const int lo = P - s > 0 ? P - s : 0, hi = P - s / 2 > 0 ? P - s / 2 : 0;
    return k >= lo && k < hi;
}

struct CfRow {  // wave-uniform per-row constants (scalar registers)
    u32 orv, cmpv, cv;
    bool ci;
};
// (a long needle's rows: NeedleLongRows below - k2d_dp_long stages them in LDS, two bytes per row)
struct NeedleLongRows : NeedleLongDev {
    const u16* cf;  // [rows] byte as compared | case flip << 8, in the workgroup's LDS
};
template <typename ND>
__device__ __forceinline__ CfRow cf_row_consts(const ND& nd, u32 r) {
    u32 c, f;
    if constexpr (ND::kLong) {
        const u32 cf = nd.cf[r];
        c = cf & 0xFF, f = cf >> 8;
    } else {
        // needle bytes through aligned dword reads of the by-value argument: wave-uniform, so they are scalar loads
        c = (((const u32*)nd.c)[r >> 2] >> (8 * (r & 3))) & 0xFF, f = (((const u32*)nd.f)[r >> 2] >> (8 * (r & 3))) & 0xFF;
    }
    CfRow k;
    k.ci = c != f;  // case-folded ASCII letter: (h | 0x20) == (c | 0x20) <=> h in {c, flip(c)}
    k.orv = k.ci ? 0x00200020u : 0u;
    k.cmpv = splat16(k.ci ? (c | 0x20) : c);
    k.cv = splat16(c);
    return k;
}

// match mask (0/1 per lane) of a row and the bonus it earns on the diagonal
template <bool UPPER>
__device__ __forceinline__ void cf_match(const CfRow& k, u32 hw, u32 bonus, u32 casev, u32& mm, u32& mb) {
    const u32 ONE = 0x00010001u;
    mm = p_subs(ONE, (hw | k.orv) ^ k.cmpv);  // v_bitop3 + v_pk_sub clamp
    mb = p_mul(mm, bonus);
    if (UPPER) {  // literal form: separate exact-case compare (bonus[] holds no case term here)
        const u32 ex = k.ci ? p_subs(ONE, hw ^ k.cv) : mm;
        mb = p_add(mb, p_mul(ex, casev));
    }
}

// propagate_horizontal_gaps over the real lanes (biased: a step is a shift, a subtract of the source's gap-open charge and a max);
// the padding entries (3b) are read off on the way into acc0 / acc1
template <int SWL, int REAL>
__device__ __forceinline__ void cf_scan(u32 (&b)[REAL], const u32 (&g)[REAL], u32 e, u32 lim, u32& acc0, u32& acc1) {
    constexpr int NW = SWL / 2, P = 2 * REAL;
    constexpr bool PAD = REAL < NW;
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
}

// Called once, half way through the rows (wave-uniform): k2b_dp_short steps the wave's issue priority down there
// (kernels_common.h, FzbProgressPrio); the default does nothing.
struct CfNoRowHook {
    __device__ __forceinline__ void operator()() const {}
};

// The rows.  hw[d] = haystack bytes of lanes 2d, 2d+1 (one per 16-bit half), bonus[d] = what a match earns on those lanes
// (match + mismatch, delimiter / capitalisation / prefix bonuses, and - unless UPPER - the matching-case bonus on the lanes that
// are not uppercase letters).  Returns max over all SWL lanes of the last row.  REAL == SWL/2: no padding lanes exist.
template <int SWL, bool UPPER, int REAL, typename RowHook = CfNoRowHook>
__device__ __forceinline__ u32 cf_rows(const NeedleDev& nd, const u32 (&hw)[REAL], const u32 (&bonus)[REAL], const RowHook& hook = RowHook()) {
    constexpr int NW = SWL / 2;
    constexpr int P = 2 * REAL;
    constexpr bool PAD = REAL < NW;
    static_assert(REAL >= 1 && REAL <= NW, "REAL");
    static_assert(!PAD || 4 * P <= 3 * SWL, "padding entries must land inside the chunk");
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex, x = nd.mismatch, o = nd.gopm;
    const u32 ev = splat16(e), gopmv = splat16(o), casev = splat16(nd.matching_case);
    const u32 xv = splat16(x), xqv = splat16(x - 2 * e);
    if (rows == 1) {  // the only row is the last row: max over the lanes of match*bonus (-) x
        const CfRow k = cf_row_consts(nd, 0);
        u32 mx = 0;
#pragma unroll
        for (int d = 0; d < REAL; d++) {
            u32 mm, mb;
            cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
            mx = p_max(mx, p_subs(mb, xv));
        }
        return max(mx & 0xFFFF, mx >> 16);
    }
    u32 T[REAL], g[REAL];
    u32 acc0 = 0, acc1 = 0;  // the padding's running maximum, A domain, two chains
    {  // ---- row 0 (peeled): the row above is the zero row - no up move, diag = match*bonus (-) x ------------------------
        const CfRow k = cf_row_consts(nd, 0);
        u32 bias = e + (e << 17);  // (L + 0 + 1) * e for lanes 0, 1
#pragma unroll
        for (int d = 0; d < REAL; d++, bias = fzb_sadd(bias, 2 * ev)) {
            u32 mm, mb;
            cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
            T[d] = p_max(p_subs(p_add(mb, bias), xv), bias);
            g[d] = p_mul(mm, gopmv);
        }
        cf_scan<SWL, REAL>(T, g, e, (rows - 2) * o, acc0, acc1);
    }
    const u32 edc = 0xFFFFu | (((u32)(P - 2) * e + x) << 16);  // (a) in the A domain: T(i-1, P-1) (-) ((P-2)*e + x), high lane only
    for (u32 r = 1; r + 1 < rows; r++) {
        if (r == rows / 2) hook();
        const CfRow k = cf_row_consts(nd, r);
        const u32 rb = (r + 1) * ev;        // the row's share of the bias
        const u32 z = ((r - 1) * e) << 16;  // T(r-1, lane -1): the zero column
        if (PAD) acc0 = p_max(acc0, p_subs(T[REAL - 1], edc));
        u32 b[REAL], gn[REAL];
        u32 bias = (e << 16) + rb;  // lanes 0, 1 of this row; + 2e per lane pair (a chain of scalar adds instead of a multiply per dword)
#pragma unroll
        for (int d = 0; d < REAL; d++, bias = fzb_sadd(bias, 2 * ev)) {
            const u32 sh = __builtin_amdgcn_alignbit(T[d], d ? T[d - 1] : z, 16);  // T(r-1, L-1)
            u32 mm, mb;
            cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
            const u32 D = p_subs(p_add(sh, mb), xqv);
            const u32 U = p_subs(T[d], g[d]);
            b[d] = p_max3_s(D, U, bias);  // (all three below 0x7C00: cf_ok)
            gn[d] = p_mul(mm, gopmv);
        }
        cf_scan<SWL, REAL>(b, gn, e, (rows - 2 - r) * o, acc0, acc1);
#pragma unroll
        for (int d = 0; d < REAL; d++) T[d] = b[d], g[d] = gn[d];
    }
    // ---- last row: cells only (2.), converted back; its maximum joins the padding's -----------------------------------------
    u32 mx;
    {
        const u32 r = rows - 1;
        const CfRow k = cf_row_consts(nd, r);
        const u32 rb = (r + 1) * ev;
        const u32 z = ((r - 1) * e) << 16;
        if (PAD) acc0 = p_max(acc0, p_subs(T[REAL - 1], edc));
        mx = p_subs(p_max(acc0, acc1), rb);  // A domain -> value
        u32 bias = (e << 16) + rb;
#pragma unroll
        for (int d = 0; d < REAL; d++, bias = fzb_sadd(bias, 2 * ev)) {
            const u32 sh = __builtin_amdgcn_alignbit(T[d], d ? T[d - 1] : z, 16);
            u32 mm, mb;
            cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
            const u32 D = p_subs(p_add(sh, mb), xqv);
            const u32 U = p_subs(T[d], g[d]);
            mx = p_max(mx, p_subs(p_max(D, U), bias));
        }
    }
    return max(mx & 0xFFFF, mx >> 16);
}

// ---- haystack-side vectors (ascii.rs:59-101), two ways ------------------------------------------------------------------
// (1) arithmetic on the byte-class bits of dp_body.h's table (general kernel)
template <int SWL, bool UPPER, int REAL>
__device__ __forceinline__ void cf_setup_bits(const NeedleDev& nd, bool include_prefix, const u8* cls, const u32 (&hb)[SWL / 4], u32 (&hw)[REAL], u32 (&bonus)[REAL]) {
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
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
    }
}

// (2) two small LDS tables (short-haystack kernel): cls2[byte] = 2 * class (0 other, 1 lower, 2 upper, 3 delimiter) and
// bon[4 * class(prev) + class(cur)] = the complete per-lane bonus as u16, so a lane costs two LDS reads and ~3 VALU ops
struct CfTables {
    u8 cls2[256];
    u16 bon[16];
};
template <bool UPPER, typename ND = NeedleDev>
__device__ __forceinline__ void cf_build_tables(const ND& nd, CfTables& t) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        const bool lower = b >= 'a' && b <= 'z', upper = b >= 'A' && b <= 'Z', digit = b >= '0' && b <= '9';
        const bool delim = !(lower || upper || digit || b > 127);
        t.cls2[b] = (u8)(2 * (lower ? 1 : upper ? 2 : delim ? 3 : 0));
    }
    if (threadIdx.x < 16) {
        const u32 cp = threadIdx.x >> 2, cc = threadIdx.x & 3;
        u32 v = nd.match_plus_mismatch;
        if (cp == 3 && cc != 3) v += nd.delimiter;       // delim(j-1) & !delim(j)
        if (cc == 2 && cp == 1) v += nd.capitalization;  // upper(j) & lower(j-1)
        if (!UPPER && cc != 2) v += nd.matching_case;    // see cf_setup_bits
        t.bon[threadIdx.x] = (u16)v;
    }
}
template <int SWL, int REAL>
__device__ __forceinline__ void cf_setup_table(const NeedleDev& nd, bool include_prefix, const CfTables& t, const u32 (&hb)[SWL / 4], u32 (&hw)[REAL], u32 (&bonus)[REAL]) {
    u32 cprev = 0;  // lane -1: no delimiter, no lowercase letter (ascii.rs:55-56)
#pragma unroll
    for (int d = 0; d < REAL; d++) {
        hw[d] = __builtin_amdgcn_perm(0u, hb[d / 2], (d & 1) ? 0x0c030c02u : 0x0c010c00u);
        const u32 c0 = t.cls2[hw[d] & 0xFF], c1 = t.cls2[hw[d] >> 16];
        const u32 i0 = (cprev << 2) | c0, i1 = (c0 << 2) | c1;  // byte offsets into bon[]
        bonus[d] = (u32) * (const u16*)((const u8*)t.bon + i0) | ((u32) * (const u16*)((const u8*)t.bon + i1) << 16);
        cprev = c1;
    }
    if (include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);  // first_lane(prefix_bonus)
}

// Scores the trimmed window (1 <= m <= 2*REAL bytes, zero padded in hb) as ONE chunk of SWL lanes of which the first
// 2*REAL are computed.
template <int SWL, bool UPPER, int REAL>
__device__ __forceinline__ u32 dp_single_chunk_cf(const NeedleDev& nd, bool include_prefix, const u8* cls, const u32 (&hb)[SWL / 4]) {
    u32 hw[REAL], bonus[REAL];
    cf_setup_bits<SWL, UPPER, REAL>(nd, include_prefix, cls, hb, hw, bonus);
    return cf_rows<SWL, UPPER, REAL>(nd, hw, bonus);
}
template <int SWL, bool UPPER, int REAL, typename RowHook = CfNoRowHook>
__device__ __forceinline__ u32 dp_single_chunk_cf_tab(const NeedleDev& nd, bool include_prefix, const CfTables& t, const u32 (&hb)[SWL / 4], const RowHook& hook = RowHook()) {
    u32 hw[REAL], bonus[REAL];
    cf_setup_table<SWL, REAL>(nd, include_prefix, t, hb, hw, bonus);
    return cf_rows<SWL, UPPER, REAL, RowHook>(nd, hw, bonus, hook);
}

// ---- 0-typo ASCII window of a haystack of at most 32 bytes held in two vectors (short-haystack kernel) ------------------------
// first occurrence of needle[0], 1 + last occurrence of needle[rows-1], either case (src/prefilter/algo/ascii.rs:6-72).  Bytes
// past the haystack's end are zero in the padded-16 layout and the needle has no NUL byte (cf_ok), so they need no masking.
// Per byte: 0x80 where it equals the needle byte; the eight dwords' flags are merged into one word with bit 8j + k = byte j of
// dword k (position 4k + j), and the smallest position falls out of four 8-bit find-first-sets.
__device__ __forceinline__ u32 cf_first_pos(u32 y) {
    u32 best = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 f = (y >> (8 * j)) & 0xFF;
        const u32 k = f ? (u32)__builtin_ctz(f) : 0x3FFFFFFFu;
        best = min(best, 4 * k + j);
    }
    return best;
}
__device__ __forceinline__ void cf_window_first_last_regs(const NeedleDev& nd, const uint4& q0, const uint4& q1, u32& ws, u32& we) {
    const u32 rows = (u32)nd.rows;
    const u32 a = nd.c[0], af = nd.f[0], z = nd.c[rows - 1], zf = nd.f[rows - 1];
    const u32 aor = a != af ? 0x20202020u : 0u, zor = z != zf ? 0x20202020u : 0u;
    const u32 apat = (a != af ? (a | 0x20) : a) * 0x01010101u, zpat = (z != zf ? (z | 0x20) : z) * 0x01010101u;
    const u32 w8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    u32 ya = 0, yz = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 ta = (w8[k] | aor) ^ apat, tz = (w8[k] | zor) ^ zpat;
        ya |= (~(((ta & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | ta) & 0x80808080u) >> (7 - k);
        yz |= (~(((tz & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | tz) & 0x80808080u) >> (7 - k);
    }
    ws = ya ? cf_first_pos(ya) : 0u;  // (no occurrence cannot happen for a survivor of the exact filter)
    // last occurrence: reversing the word maps bit 8j + k to 8(3-j) + (7-k), so "first" of the reversed word is 31 - last position
    we = yz ? 32u - cf_first_pos(__builtin_bitreverse32(yz)) : 0u;
}

// ---- the typo prefilter's window of an ACCEPTED haystack of at most 32 bytes, lane-free form ------------------------------------
// (src/prefilter/algo/ascii_typos.rs: every path records its first hit in match_start_pos, :64-88; find_end_pos_with_typos, :374-398;
//  checked against the oracle at all three widths by tests/test_oracle_reference_properties.py::test_ascii_typo_windows_have_a_lane_free_form)
//   start = the earliest first occurrence of any of needle[0..=k], end = one past the last occurrence of any of needle[n-1-k..]
//   (either case; len when none occurs).
// fl[byte]: bit 0 = the byte is one of needle[0..=k] (either case), bit 1 = one of needle[n-1-k..].  Bytes past the haystack's end
// are zero in the padded-16 layout and the needle has no NUL (cf_ok), so fl[0] = 0 masks them.
__device__ __forceinline__ void cf_build_typo_table(const NeedleDev& nd, u8* fl) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        const int n = nd.rows, k = nd.max_typos;
        u32 v = 0;
        for (int j = 0; j < n; j++) {
            if (b != nd.c[j] && b != nd.f[j]) continue;
            if (j <= k) v |= 1;
            if (j + k + 1 >= n) v |= 2;
        }
        fl[b] = (u8)v;
    }
}
__device__ __forceinline__ void cf_window_typos_regs(const u8* fl, const uint4& q0, const uint4& q1, u32 L, u32& ws, u32& we) {
    const u32 w8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    u32 mf = 0, ml = 0;
#pragma unroll
    for (int p = 0; p < 32; p++) {
        const u32 v = fl[(w8[p >> 2] >> (8 * (p & 3))) & 0xFF];
        mf |= (v & 1u) << p;
        ml |= (v >> 1) << p;
    }
    ws = mf ? (u32)__builtin_ctz(mf) : 0u;  // (an accepted haystack has one of needle[0..=k]: at most k needle bytes go unmatched)
    we = ml ? 32u - (u32)__builtin_clz(ml) : L;
}
