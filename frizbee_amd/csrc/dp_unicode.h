// Thread-per-haystack single-chunk form of the reference's UNICODE Smith-Waterman (score_haystack_unicode,
// src/smith_waterman/algo/unicode.rs:10-273, propagate_horizontal_unicode_gaps, src/smith_waterman/algo/unicode_gap.rs:110-236).
// Rows are needle SCALARS, lanes are haystack BYTES; only the first byte lane of a haystack scalar can match, and
// UTF-8 continuation bytes are zero-cost "transport" lanes for horizontal gaps.
//
// The reference carries, through its log-step gap scan, three per-step vectors that do not depend on the needle row:
//   cont_s[L] = gex * #continuation lanes in (L-s, L]   (so  tot_s - cont_s = gex * #NON-continuation lanes crossed),
//   sm_s[L]   = "a scalar start lies in (L-s, L]",       and the pending-gap-open mask that follows the match lanes.
// Both windowed quantities are differences of prefix counts, so with
//   Q[L] = #scalar-start lanes in [0, L]   and   P[L] = gex * #non-continuation lanes in [0, L]  (padding lanes count)
// the step  cand = shift_s(row) (-) ((tot_s (-) cont_s) + (gop' & shift_s(pending) & sm_s))  becomes, in the biased domain
// r'[L] = r[L] + P[L]:   cand'[L] = r'[L-s] (-) (gop' & pending[L-s] & (Q[L] != Q[L-s])).
// Exactness of the bias is the same argument as in dp_body.h (row values are >= 0, so r'[L] >= P[L] >= P[L-s]).
// Single chunk only: the adjacent chunk is the zero column.  Wider windows go to the generic wave-per-haystack kernel.
#pragma once
#include "dp_body.h"

typedef short ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 p_neg_mask(u32 neg) {  // per 16-bit lane: 0xFFFF if the lane (as i16) is negative else 0
    return __builtin_bit_cast(u32, __builtin_bit_cast(ss2, neg) >> 15);
}
// 0x80 in every byte of x that is zero (exact)
__device__ __forceinline__ u32 zflag4(u32 x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }

// REAL = packed dwords (2 lanes each) that may hold haystack bytes: the caller guarantees m <= 2 * REAL.  Lanes at or past m are
// not valid, so they are never scalar starts: diag / up are masked to 0 there, their match / pending masks are 0.  Dwords that can
// hold no haystack byte at all are not computed (see NR below).
// 0-typo unicode window of an ACCEPTED haystack (src/prefilter/algo/unicode.rs:118-219, lane-free: see host.hip, unicode DFA):
// start = the first byte position at which the FIRST needle scalar (either case variant) occurs, end = one past the last byte of
// the LAST occurrence of the last needle scalar.  Haystacks of any length: blocks of 32 bytes (two vectors, four blocks requested together)
// through the SWAR position search of unicode_window_regs below - per needle scalar one word of match positions per block, validity
// (position + scalar length <= L) as a mask.  (Round 2 walked the haystack a byte at a time with a dependent dword load per four bytes.)
template <int CL>
__device__ __forceinline__ u32 unicode_scalar_positions(const u32 (&w)[9], u32 cw);
__device__ __forceinline__ u32 unicode_scalar_positions_cl(const u32 (&w)[9], u32 cw, u32 cl);
__device__ __forceinline__ u32 unicode_valid_positions(u32 L, u32 len);
__device__ __forceinline__ u32 unicode_first_pos(u32 y);
__device__ __forceinline__ void unicode_window_first_last(const NeedleDev& nd, const u8* __restrict__ hay, u32 L, u32& ws, u32& we) {
    const u32 n = (u32)nd.rows;
    const u32 la = nd.ulen[0], lz = nd.ulen[n - 1];
    const u32 a0 = ((const u32*)nd.uc)[0], a1 = ((const u32*)nd.uf)[0], z0 = ((const u32*)nd.uc)[n - 1], z1 = ((const u32*)nd.uf)[n - 1];
    u32 ya_keep = 0, yz_keep = 0, ba = 0, bz = 0;  // block and position word of the first / last occurrence; extracted after the scan
    const uint4* vp = (const uint4*)hay;  // every haystack starts on a 16-byte boundary; >= 80 readable bytes follow the corpus
    const u32 nblk = (L + 31) >> 5;
    for (u32 b0 = 0; b0 < nblk; b0 += 4) {
        uint4 qs[9];  // four blocks + the vector behind them (its first dword completes the last block's shifted views)
#pragma unroll
        for (int k = 0; k < 9; k++) qs[k] = 32 * b0 + 16 * k < L + 4 ? vp[2 * b0 + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32 b = b0 + t;
            if (b >= nblk) break;
            const u32 w[9] = {qs[2 * t].x, qs[2 * t].y, qs[2 * t].z, qs[2 * t].w, qs[2 * t + 1].x, qs[2 * t + 1].y, qs[2 * t + 1].z, qs[2 * t + 1].w, qs[2 * t + 2].x};
            const u32 Lb = L - 32 * b;  // bytes of the haystack from this block on (>= 1)
            u32 ya = unicode_scalar_positions_cl(w, a0, la);
            if (a1 != a0) ya |= unicode_scalar_positions_cl(w, a1, la);
            u32 yz = unicode_scalar_positions_cl(w, z0, lz);
            if (z1 != z0) yz |= unicode_scalar_positions_cl(w, z1, lz);
            ya &= unicode_valid_positions(Lb, la);
            yz &= unicode_valid_positions(Lb, lz);
            if (ya_keep == 0 && ya) { ya_keep = ya; ba = b; }
            if (yz) { yz_keep = yz; bz = b; }
        }
    }
    ws = ya_keep ? 32 * ba + unicode_first_pos(ya_keep) : 0u;  // (no occurrence cannot happen for a survivor of the exact filter)
    we = yz_keep ? 32 * bz + 31u - unicode_first_pos(__builtin_bitreverse32(yz_keep)) + lz : 0u;
}

// The window of a haystack a unicode TYPO query accepted, in its lane-free form (unicode_typos.rs:15-141 `match_start_pos`, :469-509
// find_end_pos_with_unicode_typos; tests/test_oracle_reference_properties.py::test_unicode_typo_windows_have_the_same_lane_free_form): start = the
// earliest occurrence of any of the needle scalars [0 ..= k], end = the end of the latest occurrence of any of the scalars [n-1-k ..] (occurrences
// never overlap, so the latest start has the latest end), the haystack's length if there is none.  One block of 32 bytes: `head` = OR of the
// position words of the head rows, `tail_end` = max over the tail rows of (last position + scalar length), 0 if none occurs.
__device__ __forceinline__ void unicode_typo_block(const NeedleDev& nd, const u32 (&w)[9], u32 Lb, u32 k, u32& head, u32& tail_end) {
    const u32 n = (u32)nd.rows;
    head = 0;
    tail_end = 0;
    for (u32 j = 0; j <= k && j < n; j++) {  // wave-uniform trip counts
        const u32 c0 = ((const u32*)nd.uc)[j], c1 = ((const u32*)nd.uf)[j], cl = nd.ulen[j];
        u32 y = unicode_scalar_positions_cl(w, c0, cl);
        if (c1 != c0) y |= unicode_scalar_positions_cl(w, c1, cl);
        head |= y & unicode_valid_positions(Lb, cl);
    }
    for (u32 j = n > k + 1 ? n - 1 - k : 0; j < n; j++) {
        const u32 c0 = ((const u32*)nd.uc)[j], c1 = ((const u32*)nd.uf)[j], cl = nd.ulen[j];
        u32 y = unicode_scalar_positions_cl(w, c0, cl);
        if (c1 != c0) y |= unicode_scalar_positions_cl(w, c1, cl);
        y &= unicode_valid_positions(Lb, cl);
        if (y) tail_end = max(tail_end, 31u - unicode_first_pos(__builtin_bitreverse32(y)) + cl);
    }
}
__device__ __forceinline__ void unicode_window_typos(const NeedleDev& nd, const u8* __restrict__ hay, u32 L, u32 k, u32& ws, u32& we) {
    u32 head_keep = 0, bh = 0, end = 0;
    const uint4* vp = (const uint4*)hay;  // every haystack starts on a 16-byte boundary; >= 80 readable bytes follow the corpus
    const u32 nblk = (L + 31) >> 5;
    for (u32 b0 = 0; b0 < nblk; b0 += 4) {
        uint4 qs[9];  // four blocks + the vector behind them (its first dword completes the last block's shifted views)
#pragma unroll
        for (int t = 0; t < 9; t++) qs[t] = 32 * b0 + 16 * t < L + 4 ? vp[2 * b0 + t] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32 b = b0 + t;
            if (b >= nblk) break;
            const u32 w[9] = {qs[2 * t].x, qs[2 * t].y, qs[2 * t].z, qs[2 * t].w, qs[2 * t + 1].x, qs[2 * t + 1].y, qs[2 * t + 1].z, qs[2 * t + 1].w, qs[2 * t + 2].x};
            u32 head, tail_end;
            unicode_typo_block(nd, w, L - 32 * b, k, head, tail_end);
            if (head_keep == 0 && head) { head_keep = head; bh = b; }
            if (tail_end) end = 32 * b + tail_end;
        }
    }
    ws = head_keep ? 32 * bh + unicode_first_pos(head_keep) : 0u;
    we = end ? end : L;
}
__device__ __forceinline__ void unicode_window_typos_regs(const NeedleDev& nd, const uint4& q0, const uint4& q1, u32 L, u32 k, u32& ws, u32& we);

template <int SWL, int REAL = SWL / 2>
__device__ __forceinline__ u32 dp_unicode_single_chunk(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const u8* cls) {
    constexpr int NW = SWL / 2;
    constexpr int NB = SWL / 4;
    constexpr int RB = (REAL + 1) / 2;  // byte dwords that may hold haystack bytes
    // Score dwords that are computed: the 2 * RB that can hold a haystack byte.  The dwords above them are pure padding, and nothing they
    // hold is ever read: values only travel rightwards (gap steps, diagonal), a padding lane is never a scalar start (its diag / up are
    // masked to 0), and the score is the maximum of the LAST row before its propagation (below), where padding lanes are 0.  (Round 1 took
    // the maximum after the propagation and had to carry all SWL / 2 dwords.)
    constexpr int NR = 2 * RB;
    static_assert(REAL >= 1 && REAL <= NW, "REAL");
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    // ---- bytes (+1 guard dword of zeros for the shifted views) ----
    u32 hb[RB + 1];
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const u32 p = 4 * k;
        u32 v = 0;
        if (p < m) {
            v = load_u32_unaligned(th, p);
            const u32 rem = m - p;
            if (rem < 4) v &= (1u << (8 * rem)) - 1;
        }
        hb[k] = v;
    }
    hb[RB] = 0;
    // ---- per-byte scalar-start flags (0x80 where valid && !continuation), prefix counts Q, bonus ----
    u32 Q[2 * RB], bonus[2 * RB];
    u32 qtotv;  // the final count in both halves: Q of every padding dword
    {
        u32 clsw_prev = 0;
        u32 qrun = 0;  // running count, replicated in both halves
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const u32 w = hb[k];
            // continuation byte: 0x80..0xBF  <=>  (b & 0xC0) == 0x80
            const u32 contf = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u);
            // valid lanes: position < m
            const u32 p = 4 * k;
            const u32 nv = m > p ? min(m - p, 4u) : 0u;
            const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
            const u32 sflagk = validf & ~contf;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int d = 2 * k + h;
                const u32 b0 = h ? (w >> 16) & 0xFF : w & 0xFF;
                const u32 b1 = h ? w >> 24 : (w >> 8) & 0xFF;
                const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
                const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);
                const u32 cap01 = (clsw >> 1) & sh & ONE;
                const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;
                bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
                clsw_prev = clsw;
                // scalar-start 0/1 for the two lanes of this dword
                const u32 t = sflagk >> 7;
                const u32 s0 = h ? (t >> 16) & 1 : t & 1;
                const u32 s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
                const u32 q0 = qrun + s0, q1 = q0 + s1;
                Q[d] = q0 | (q1 << 16);
                qrun = q1;
            }
        }
        if (include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
        qtotv = qrun | (qrun << 16);
    }
    auto Qof = [&](int d) { return d < 2 * RB ? Q[d < 2 * RB ? d : 0] : qtotv; };
    const u32 mv = splat16(m);
    // P[d] = gex * (Q[d] + #padding lanes up to and including the lane) ; padding lanes are non-continuation lanes too
    auto Pof = [&](int d) {
        const u32 lanepos1 = (u32)(2 * d + 1) | ((u32)(2 * d + 2) << 16);
        return p_mul(p_add(Qof(d), p_subs(lanepos1, mv)), gexv);
    };
    u32 prev[NR], upm[2 * RB];
#pragma unroll
    for (int d = 0; d < NR; d++) prev[d] = 0;
#pragma unroll
    for (int d = 0; d < 2 * RB; d++) upm[d] = 0;
#pragma unroll 1
    for (u32 r = 0; r < rows; r++) {
        const u32 cl = nd.ulen[r];
        const u8* uc = nd.uc[r];
        const u8* uf = nd.uf[r];
        const bool two = (uc[0] != uf[0]) || (uc[1] != uf[1]) || (uc[2] != uf[2]) || (uc[3] != uf[3]);
        // Keep what is derived from the prefix counts and the byte views (scalar-start masks, biases, shifted views) out of registers
        // across rows: the compiler otherwise hoists ~100 loop-invariant values out of this loop (k2u_dp_unicode_half<64>: 256 VGPRs + 92
        // bytes of scratch at 2 waves per SIMD; with these barriers 168 VGPRs at 3 waves, 5 % fewer VALU instructions: C5 179 -> 162 us).
#pragma unroll
        for (int d = 0; d < 2 * RB; d++) FZB_OPAQUE_V(Q[d]);
#pragma unroll
        for (int k = 0; k <= RB; k++) FZB_OPAQUE_V(hb[k]);
        // ---- byte-level match flags: scalar start && bytes [L, L+cl) equal the needle scalar (unicode.rs:221-241) ----
        u32 row[NR], pend[2 * RB];
#pragma unroll
        for (int k = 0; k < RB; k++) {
            const u32 w0 = hb[k];
            const u32 w1 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 1);
            const u32 w2 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 2);
            const u32 w3 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 3);
            const u32 wl = cl == 1 ? w0 : cl == 2 ? w1 : cl == 3 ? w2 : w3;  // view holding each lane's LAST scalar byte
            u32 fe = zflag4(wl ^ (uc[cl - 1] * 0x01010101u));
            if (cl > 1) fe &= zflag4(w0 ^ (uc[0] * 0x01010101u));
            if (cl > 2) fe &= zflag4(w1 ^ (uc[1] * 0x01010101u));
            if (cl > 3) fe &= zflag4(w2 ^ (uc[2] * 0x01010101u));
            u32 fm = fe;
            if (two) {
                u32 ff = zflag4(wl ^ (uf[cl - 1] * 0x01010101u));
                if (cl > 1) ff &= zflag4(w0 ^ (uf[0] * 0x01010101u));
                if (cl > 2) ff &= zflag4(w1 ^ (uf[1] * 0x01010101u));
                if (cl > 3) ff &= zflag4(w2 ^ (uf[2] * 0x01010101u));
                fm |= ff;
            }
            // expand to per-lane 16-bit masks (two score dwords per byte dword)
            const u32 te = fe >> 7, tm = fm >> 7;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int d = 2 * k + h;
                const u32 sel = h ? 0x0c030c02u : 0x0c010c00u;
                const u32 e01 = __builtin_amdgcn_perm(0u, te, sel) & ONE;
                const u32 m01 = __builtin_amdgcn_perm(0u, tm, sel) & ONE;
                // scalar-start mask of the two lanes, from the prefix counts; only scalar-start lanes can match
                const u32 qs = __builtin_amdgcn_alignbit(Q[d], d ? Q[d - 1] : 0u, 16);
                const u32 sst = p_neg_mask(p_sub(qs, Q[d]));
                const u32 exm = p_sub(0u, e01) & sst, mmk = p_sub(0u, m01) & sst;  // 0xFFFF where set
                // diagonal / up (unicode.rs:165-182), both masked to scalar-start lanes
                const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : 0u, 16);
                u32 t = p_add(sh, mmk & bonus[d]);
                t = p_subs(t, Xv);
                const u32 diag = p_add(t, exm & casev);
                const u32 up = p_subs(p_subs(prev[d], gexv), upm[d] & gopmv);  // upm[d] still holds the PREVIOUS row's match mask
                row[d] = p_max(diag, up) & sst;
                pend[d] = mmk;
                upm[d] = mmk;  // ... and from here on this row's (read again only by the next row)
            }
        }
        // the LAST needle row is not propagated: every value the gap scan produces is an earlier lane's value minus a non-negative
        // cost (continuation bytes are free transport lanes, never a gain), and only the row's maximum is read (unicode.rs: the
        // horizontal max of the last row) - so that maximum is the maximum before the scan
        if (r + 1 == rows) {
            u32 mxl = row[0];
#pragma unroll
            for (int d = 1; d < 2 * RB; d++) mxl = p_max(mxl, row[d]);
            return max(mxl & 0xFFFF, mxl >> 16);
        }
        // ---- propagate_horizontal_unicode_gaps in the biased domain ----
#pragma unroll
        for (int d = 0; d < NR; d++) row[d] = p_add(row[d], Pof(d));
        // every step updates in place from the highest dword down: entry d only reads entries <= d, which are still the
        // values from before the step (keeps the live register set to one copy of row / pending)
#pragma unroll
        for (int d = NR - 1; d >= 0; d--) {
            const u32 bs = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : 0u, 16);
            if (d <= 2 * RB) {  // a source lane (2d-1 or 2d) may be a real lane
                const u32 ps = __builtin_amdgcn_alignbit(d < 2 * RB ? pend[d < 2 * RB ? d : 0] : 0u, (d && d - 1 < 2 * RB) ? pend[d ? d - 1 : 0] : 0u, 16);
                const u32 qs = __builtin_amdgcn_alignbit(Qof(d), d ? Qof(d - 1) : 0u, 16);
                const u32 fl = p_neg_mask(p_sub(qs, Qof(d)));  // a scalar start lies in (L-1, L]
                row[d] = p_max(row[d], p_subs(bs, ps & fl & gopmv));
                if (d < 2 * RB) pend[d] = pend[d] | (ps & ~fl);
            } else {
                row[d] = p_max(row[d], bs);  // padding -> padding: no scalar start crossed, nothing pending can be charged
            }
        }
#pragma unroll
        for (int off = 1; off < NR; off *= 2) {
#pragma unroll
            for (int d = NR - 1; d >= off; d--) {
                if (d - off < 2 * RB) {
                    const u32 fl = p_neg_mask(p_sub(Q[d - off < 2 * RB ? d - off : 0], Qof(d)));
                    row[d] = p_max(row[d], p_subs(row[d - off], pend[d - off < 2 * RB ? d - off : 0] & fl & gopmv));
                    if (d < 2 * RB) pend[d] = pend[d] | (pend[d - off < 2 * RB ? d - off : 0] & ~fl);
                } else {
                    row[d] = p_max(row[d], row[d - off]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < NR; d++) prev[d] = p_sub(row[d], Pof(d));
    }
    u32 mx = prev[0];  // (rows == 0 only: the loop returns from its last row)
#pragma unroll
    for (int d = 1; d < NR; d++) mx = p_max(mx, prev[d]);
    return max(mx & 0xFFFF, mx >> 16);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Windows WIDER than one chunk (SWL < m <= 1024 bytes), still one thread per haystack (round 4; until then every such window took a whole
// wavefront in kernels_generic.hip - 45 k of them under "All Scores" on the Arabic-shaped list: 209 of the step's 305 us).
// The reference walks the window chunk by chunk (score_haystack_unicode, unicode.rs:10-217): per needle row it keeps the previous chunk's
// propagated row (score_matrix) and pending mask (unicode_pending), and its gap scan shifts those in from the left together with the
// previous chunk's continuation-byte costs and scalar-start mask (propagate_horizontal_unicode_gaps, unicode_gap.rs:110-236); the largest
// shift is half a chunk, so only the TOP HALF of the previous chunk is ever read.  Here, per chunk:
//   * the prefix counts of the first form run over [adjacent half | this chunk]: Q = #scalar starts, P = gex * #non-continuation lanes,
//     both counted from the adjacent half's first lane (QA / Q, PA / P) - the windowed quantities of a step are their differences whether
//     or not the window crosses the chunk boundary; the adjacent half's counts are recomputed from its bytes (no state to carry);
//   * per needle row the previous chunk's top half comes back from the scratch slab ([row][dword][thread], as dp_body.h's dp_multi_chunk
//     parks them: values unbiased - bytes in the u8 class - and the pending mask as bits), is biased with PA and serves the low source
//     lanes of every step; its last lane is the next row's diagonal into lane 0;
//   * the last needle row is neither propagated nor parked in ANY chunk: what its scan would produce - here or, through the adjacent
//     half, in the next chunk's last row - is some unpropagated last-row value minus costs, and only the maximum over the last rows is read.
// First-form arithmetic (the bias is added before a row's scan and removed after it); preconditions as dp_unicode_single_chunk
// (LaunchCfg::bias_ok).  tests/test_kernel_math_host.py fuzzes it against the oracle at every lane width.
// ------------------------------------------------------------------------------------------------------------------------------
template <int SWL>
__device__ __forceinline__ u32 dp_unicode_multi_chunk(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const u8* cls,
                                                      u32* __restrict__ scratch, u32 sstride, u32 sidx) {
    constexpr int NW = SWL / 2;        // score dwords of a chunk
    constexpr int NB = SWL / 4;        // byte dwords of a chunk
    constexpr int HT = NW / 2;         // score dwords of the adjacent half
    constexpr int NCHG = (HT + 7) / 8; // parked words of pending bits (two lanes of a dword 16 bits apart, as dp_cfm.h's gap-open flags)
    static_assert(HT >= 1 && HT + NCHG <= NW, "a parked row must fit its NW dwords of the slab");
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const bool u8class = nd.lane_mask == 0xFF;
    const u32 nchunks = (m + SWL - 1) / SWL;
    const u32 mv = splat16(m);
    u32 mx = 0;
    u32 clsw_prev = 0;  // byte classes of the lane before the current one: carried across chunks (unicode.rs: prev_chunk_*_mask)
#pragma unroll 1
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 cbase = ch * SWL;
        // ---- bytes of the chunk (+ the next chunk's first dword for the shifted views: the reference loads its four byte views at
        // chunk_start + 0..3 up to the haystack's end) ----
        u32 hb[NB + 1];
#pragma unroll
        for (int k = 0; k <= NB; k++) {
            const u32 p = cbase + 4 * k;
            u32 v = 0;
            if (p < m) {
                v = load_u32_unaligned(th, p);
                const u32 rem = m - p;
                if (rem < 4) v &= (1u << (8 * rem)) - 1;
            }
            hb[k] = v;
        }
        // ---- scalar-start prefix counts: the adjacent half (every lane valid), then this chunk; bonuses ----
        u32 QA[HT], Q[NW], bonus[NW];
        u32 qrun = 0;
        if (ch) {
#pragma unroll
            for (int k = 0; k < NB / 2; k++) {
                const u32 w = load_u32_unaligned(th, cbase - SWL / 2 + 4 * k);
                const u32 t = (0x80808080u & ~zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u)) >> 7;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 s0 = h ? (t >> 16) & 1 : t & 1, s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
                    const u32 q0 = qrun + s0, q1 = q0 + s1;
                    QA[2 * k + h] = q0 | (q1 << 16);
                    qrun = q1;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < HT; t++) QA[t] = 0;
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const u32 w = hb[k];
            const u32 contf = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u);
            const u32 p = cbase + 4 * k;
            const u32 nv = m > p ? min(m - p, 4u) : 0u;
            const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
            const u32 t = (validf & ~contf) >> 7;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int d = 2 * k + h;
                const u32 b0 = h ? (w >> 16) & 0xFF : w & 0xFF;
                const u32 b1 = h ? w >> 24 : (w >> 8) & 0xFF;
                const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
                const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);
                const u32 cap01 = (clsw >> 1) & sh & ONE;
                const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;
                bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
                clsw_prev = clsw;
                const u32 s0 = h ? (t >> 16) & 1 : t & 1, s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
                const u32 q0 = qrun + s0, q1 = q0 + s1;
                Q[d] = q0 | (q1 << 16);
                qrun = q1;
            }
        }
        if (ch == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
        // P = gex * (Q + #padding lanes up to and including the lane): lanes at or past m are non-continuation lanes that start no scalar
        auto Pof = [&](int d) {
            const u32 lanepos1 = (cbase + (u32)(2 * d + 1)) | ((cbase + (u32)(2 * d + 2)) << 16);
            return p_mul(p_add(Q[d], p_subs(lanepos1, mv)), gexv);
        };
        u32 prev[NW], upm[NW];
#pragma unroll
        for (int d = 0; d < NW; d++) prev[d] = 0, upm[d] = 0;
        u32 carry = 0;  // the previous chunk's last lane of the row above (the diagonal into lane 0); row -1 is the zero row
#pragma unroll 1
        for (u32 r = 0; r < rows; r++) {
            const u32 cl = nd.ulen[r];
            const u8* uc = nd.uc[r];
            const u8* uf = nd.uf[r];
            const bool two = (uc[0] != uf[0]) || (uc[1] != uf[1]) || (uc[2] != uf[2]) || (uc[3] != uf[3]);
#pragma unroll
            for (int d = 0; d < NW; d++) FZB_OPAQUE_V(Q[d]);  // (see dp_unicode_single_chunk: keeps what derives from Q / hb out of the loop-invariant set)
#pragma unroll
            for (int k = 0; k <= NB; k++) FZB_OPAQUE_V(hb[k]);
            u32 row[NW], pend[NW];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const u32 w0 = hb[k];
                const u32 w1 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 1);
                const u32 w2 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 2);
                const u32 w3 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 3);
                const u32 wl = cl == 1 ? w0 : cl == 2 ? w1 : cl == 3 ? w2 : w3;
                u32 fe = zflag4(wl ^ (uc[cl - 1] * 0x01010101u));
                if (cl > 1) fe &= zflag4(w0 ^ (uc[0] * 0x01010101u));
                if (cl > 2) fe &= zflag4(w1 ^ (uc[1] * 0x01010101u));
                if (cl > 3) fe &= zflag4(w2 ^ (uc[2] * 0x01010101u));
                u32 fm = fe;
                if (two) {
                    u32 ff = zflag4(wl ^ (uf[cl - 1] * 0x01010101u));
                    if (cl > 1) ff &= zflag4(w0 ^ (uf[0] * 0x01010101u));
                    if (cl > 2) ff &= zflag4(w1 ^ (uf[1] * 0x01010101u));
                    if (cl > 3) ff &= zflag4(w2 ^ (uf[2] * 0x01010101u));
                    fm |= ff;
                }
                const u32 te = fe >> 7, tm = fm >> 7;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int d = 2 * k + h;
                    const u32 sel = h ? 0x0c030c02u : 0x0c010c00u;
                    const u32 e01 = __builtin_amdgcn_perm(0u, te, sel) & ONE;
                    const u32 m01 = __builtin_amdgcn_perm(0u, tm, sel) & ONE;
                    const u32 qs = __builtin_amdgcn_alignbit(Q[d], d ? Q[d - 1] : QA[HT - 1], 16);
                    const u32 sst = p_neg_mask(p_sub(qs, Q[d]));  // scalar-start lanes (the count grows there)
                    const u32 exm = p_sub(0u, e01) & sst, mmk = p_sub(0u, m01) & sst;
                    const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : (carry << 16), 16);
                    u32 t = p_add(sh, mmk & bonus[d]);
                    t = p_subs(t, Xv);
                    const u32 diag = p_add(t, exm & casev);
                    const u32 up = p_subs(p_subs(prev[d], gexv), upm[d] & gopmv);
                    row[d] = p_max(diag, up) & sst;
                    pend[d] = mmk;
                    upm[d] = mmk;
                }
            }
            if (r + 1 == rows) {  // the last row: only its maximum is read, in every chunk
#pragma unroll
                for (int d = 0; d < NW; d++) mx = p_max(mx, row[d]);
                break;
            }
            // ---- the previous chunk's top half of this row: biased values, pending masks; its last lane feeds the next row's diagonal ----
            u32 ca[HT], apend[HT];
            u32* srow = scratch + (size_t)(r * NW) * sstride + sidx;
            u32 carry_next = 0;
            if (ch) {
                u32 arow[HT];
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) {
                        const u32 pk = srow[(size_t)t * sstride];
                        arow[2 * t] = __builtin_amdgcn_perm(0u, pk, 0x0c010c00u);
                        arow[2 * t + 1] = __builtin_amdgcn_perm(0u, pk, 0x0c030c02u);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) arow[t] = srow[(size_t)t * sstride];
                }
                u32 pch[NCHG];
#pragma unroll
                for (int a = 0; a < NCHG; a++) pch[a] = srow[(size_t)(HT + a) * sstride];
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    ca[t] = p_add(arow[t], p_mul(QA[t], gexv));
                    apend[t] = p_sub(0u, (pch[t / 8] >> (2 * (t % 8))) & ONE);
                }
                carry_next = arow[HT - 1] >> 16;
            } else {
#pragma unroll
                for (int t = 0; t < HT; t++) ca[t] = 0u, apend[t] = 0u;
            }
            // ---- propagate_horizontal_unicode_gaps in the biased domain over [adjacent half | chunk] ----
#pragma unroll
            for (int d = 0; d < NW; d++) row[d] = p_add(row[d], Pof(d));
#pragma unroll
            for (int d = NW - 1; d >= 0; d--) {  // shift by one lane; in place from the top: entry d reads entries <= d, still unchanged
                const u32 bs = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : ca[HT - 1], 16);
                const u32 ps = __builtin_amdgcn_alignbit(pend[d], d ? pend[d - 1] : apend[HT - 1], 16);
                const u32 qs = __builtin_amdgcn_alignbit(Q[d], d ? Q[d - 1] : QA[HT - 1], 16);
                const u32 fl = p_neg_mask(p_sub(qs, Q[d]));  // a scalar start lies in (L-1, L]
                row[d] = p_max(row[d], p_subs(bs, ps & fl & gopmv));
                pend[d] = pend[d] | (ps & ~fl);
            }
#pragma unroll
            for (int off = 1; off < NW; off *= 2) {  // shifts of 2, 4, ..., SWL/2 lanes
#pragma unroll
                for (int d = NW - 1; d >= 0; d--) {
                    const bool adj = d < off;
                    const int si = adj ? HT + d - off : d - off;
                    const u32 rs = adj ? ca[si] : row[si], psrc = adj ? apend[si] : pend[si], qsrc = adj ? QA[si] : Q[si];
                    const u32 fl = p_neg_mask(p_sub(qsrc, Q[d]));
                    row[d] = p_max(row[d], p_subs(rs, psrc & fl & gopmv));
                    pend[d] = pend[d] | (psrc & ~fl);
                }
            }
#pragma unroll
            for (int d = 0; d < NW; d++) prev[d] = p_sub(row[d], Pof(d));
            // ---- park the top half for the next chunk ----
            if (ch + 1 < nchunks) {
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) srow[(size_t)t * sstride] = __builtin_amdgcn_perm(prev[HT + 2 * t + 1], prev[HT + 2 * t], 0x06040200u);
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) srow[(size_t)t * sstride] = prev[HT + t];
                }
                u32 chg[NCHG] = {};
#pragma unroll
                for (int t = 0; t < HT; t++) chg[t / 8] |= (pend[HT + t] & ONE) << (2 * (t % 8));
#pragma unroll
                for (int a = 0; a < NCHG; a++) srow[(size_t)(HT + a) * sstride] = chg[a];
            }
            carry = carry_next;
        }
    }
    return max(mx & 0xFFFF, mx >> 16);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same scorer in dp_cf.h's manner (round 3): BIASED THROUGHOUT.  T(i, L) = S(i, L) + P[L] + (i + 1) * e, with P as above (e per
// non-continuation lane up to and including L) and one gap-extend per row, is what the registers hold from the first row to the last:
//   * up:    S(i-1, L) (-) e (-) g   becomes   T(i-1, L) (-) g            (the row bias pays the e; g = gop' where the cell above matched)
//   * diag:  only scalar-start lanes take it, and for those P[L] - P[L-1] = e, so the bias grows by 2e along the diagonal:
//            ((S(i-1, L-1) + match*bonus) (-) x) + case   becomes   ((T(i-1, L-1) + match*bonus) (-) (x - 2e)) + case
//   * the reference's floor at 0 is one max with the cell's bias B(i, L) = P[L] + (i + 1) * e (a match cell is above it anyway:
//     bonus >= match + mismatch >= x); lanes that are not scalar starts hold exactly B (score 0)
//   * the gap scan is the biased step of the first form with nothing added before and nothing removed after it
//   * B(i, L) is ONE v_pk_mad_u16 from Qp[L] = #scalar starts + #padding lanes in [0, L] (the only per-lane state besides the
//     scalar-start flags; its differences are also the "a scalar start was crossed" test - padding lanes count as crossings, which
//     only changes values IN padding lanes, and those are never read: see NR above)
// and the needle row's bytes come from three scalar loads of the by-value argument (dwords of uc / uf / ulen) instead of byte loads
// with a dynamic index, which the compiler turned into eight dependent global loads per row; the byte compares are specialised per
// scalar length (a wave-uniform switch) and OR-ed before ONE zero-byte test per case variant.
// Preconditions (host, LaunchCfg::cfu_ok): 2 * gap_extend <= mismatch_penalty and the biased values fit 16 bits (bias_ok).
// k2u_dp_unicode_half<64>: 2143 -> see DESIGN.md; tests/test_kernel_math_host.py fuzzes this form against the oracle and the first form.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 p_mad(u32 a, u32 b, u32 c) { return as_u32(as_us2(a) * as_us2(b) + as_us2(c)); }

// true if bytes [0, m) of `th` hold four UTF-8 continuation bytes (0x80..0xBF) in a row - i.e. the window is not UTF-8 as far as the gap
// scan's shortcut is concerned (dp_unicode_single_chunk_t, UTF8)
template <int NBYTES>
__device__ __forceinline__ bool unicode_has_cont_run4(const u8* __restrict__ th, u32 m) {
    u32 run4 = 0, cont_prev = 0;
#pragma unroll
    for (int k = 0; k < NBYTES / 4; k++) {
        const u32 p = 4 * k;
        u32 w = 0;
        if (p < m) w = load_u32_unaligned(th, p);
        const u32 nv = m > p ? min(m - p, 4u) : 0u;
        const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
        const u32 cv = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u) & validf;
        // byte j of the shifted views = the flag of the byte 3 / 2 / 1 positions before byte j
        run4 |= cv & __builtin_amdgcn_alignbyte(cv, cont_prev, 1) & __builtin_amdgcn_alignbyte(cv, cont_prev, 2) & __builtin_amdgcn_alignbyte(cv, cont_prev, 3);
        cont_prev = cv;
    }
    return run4 != 0;
}

// The needle row's byte-level match flags of dp_unicode_single_chunk_t, specialised per scalar length CL and per "the row has a second case
// variant" (both wave-uniform: ONE switch per row instead of branches inside the unrolled loop over the byte dwords): fe01 / fm01 = 0x01 in
// every byte that is a scalar start and whose CL bytes equal the needle scalar (exact case / either case).  unicode.rs:221-241.
template <int RB, int CL, bool TWO>
__device__ __forceinline__ void unicode_row_flags(const u32 (&hb)[RB + 1], const u32 (&sflag)[RB], u32 ucw, u32 ufw, u32 (&fe01)[RB], u32 (&fm01)[RB]) {
    const u32 c0 = (ucw & 0xFF) * 0x01010101u, c1 = ((ucw >> 8) & 0xFF) * 0x01010101u, c2 = ((ucw >> 16) & 0xFF) * 0x01010101u, c3 = (ucw >> 24) * 0x01010101u;
    const u32 f0 = (ufw & 0xFF) * 0x01010101u, f1 = ((ufw >> 8) & 0xFF) * 0x01010101u, f2 = ((ufw >> 16) & 0xFF) * 0x01010101u, f3 = (ufw >> 24) * 0x01010101u;
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const u32 v0 = hb[k];
        u32 ze = v0 ^ c0, zf = v0 ^ f0;
        if (CL > 1) {
            const u32 v1 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 1);
            ze |= v1 ^ c1;
            if (TWO) zf |= v1 ^ f1;
        }
        if (CL > 2) {
            const u32 v2 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 2);
            ze |= v2 ^ c2;
            if (TWO) zf |= v2 ^ f2;
        }
        if (CL > 3) {
            const u32 v3 = __builtin_amdgcn_alignbyte(hb[k + 1], hb[k], 3);
            ze |= v3 ^ c3;
            if (TWO) zf |= v3 ^ f3;
        }
        const u32 fe = zflag4(ze) & sflag[k];
        fe01[k] = fe >> 7;
        fm01[k] = TWO ? (fe | (zflag4(zf) & sflag[k])) >> 7 : fe01[k];
    }
}

// set-up of the biased-throughout form from the window's bytes hb[0 .. RB) (zero padded, hb[RB] = 0): scalar-start flags per byte (0x80),
// Qp = #scalar starts + #padding lanes up to and including the lane, the bonus vector.  Returns whether the window holds four UTF-8
// continuation bytes in a row (unicode_has_cont_run4's answer, from the continuation flags this needs anyway).
template <int SWL, int REAL>
__device__ __forceinline__ bool unicode_setup_t(const NeedleDev& nd, const u32 (&hb)[(REAL + 1) / 2 + 1], u32 m, bool include_prefix, const u8* cls,
                                                u32 (&sflag)[(REAL + 1) / 2], u32 (&Qp)[2 * ((REAL + 1) / 2)], u32 (&bonus)[2 * ((REAL + 1) / 2)]) {
    constexpr int RB = (REAL + 1) / 2;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const u32 mv = splat16(m);
    u32 clsw_prev = 0, qrun = 0, run4 = 0, cont_prev = 0;
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const u32 w = hb[k];
        const u32 contf = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u);  // continuation byte: (b & 0xC0) == 0x80
        const u32 p = 4 * k;
        const u32 nv = m > p ? min(m - p, 4u) : 0u;
        const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
        sflag[k] = validf & ~contf;
        const u32 cv = contf & validf;
        // byte j of the shifted views = the flag of the byte 3 / 2 / 1 positions before byte j
        run4 |= cv & __builtin_amdgcn_alignbyte(cv, cont_prev, 1) & __builtin_amdgcn_alignbyte(cv, cont_prev, 2) & __builtin_amdgcn_alignbyte(cv, cont_prev, 3);
        cont_prev = cv;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int d = 2 * k + h;
            const u32 b0 = h ? (w >> 16) & 0xFF : w & 0xFF;
            const u32 b1 = h ? w >> 24 : (w >> 8) & 0xFF;
            const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
            const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);
            const u32 cap01 = (clsw >> 1) & sh & ONE;
            const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;
            bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
            clsw_prev = clsw;
            const u32 t = sflag[k] >> 7;
            const u32 s0 = h ? (t >> 16) & 1 : t & 1;
            const u32 s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
            const u32 q0 = qrun + s0, q1 = q0 + s1;
            const u32 lanepos1 = (u32)(2 * d + 1) | ((u32)(2 * d + 2) << 16);
            Qp[d] = p_add(q0 | (q1 << 16), p_subs(lanepos1, mv));
            qrun = q1;
        }
    }
    if (include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
    return run4 != 0;
}

// UTF8 = true: the caller has checked (wave-uniform) that no window of the wave holds four continuation bytes in a row.  In UTF-8 a scalar
// has at most three, so every window of >= 4 lanes holds a scalar start (or padding, which Qp counts the same way): the "crossed a scalar
// start" test of the 4-, 8-, 16-lane steps is then always true - the pending charge is always taken and the pending mask never moves - and
// those steps are a subtract and a max.  Any other byte string takes UTF8 = false (every test kept).
template <int SWL, int REAL, bool UTF8>
__device__ __forceinline__ u32 unicode_rows_t(const NeedleDev& nd, u32 (&hb)[(REAL + 1) / 2 + 1], u32 (&sflag)[(REAL + 1) / 2], u32 (&Qp)[2 * ((REAL + 1) / 2)],
                                              const u32 (&bonus)[2 * ((REAL + 1) / 2)]) {
    constexpr int NW = SWL / 2;
    constexpr int RB = (REAL + 1) / 2;  // byte dwords that may hold haystack bytes
    constexpr int NR = 2 * RB;          // score dwords computed (two lanes each)
    static_assert(REAL >= 1 && REAL <= NW, "REAL");
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex;
    const u32 xqv = splat16(nd.mismatch - 2 * e), gexv = splat16(e), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case);
    u32 prev[NR], upg[NR];  // upg / pendg: the match masks already AND-ed with gop' (every use of them is)
#pragma unroll
    for (int d = 0; d < NR; d++) prev[d] = p_mul(Qp[d], gexv), upg[d] = 0;  // T(-1, L) = B(-1, L) = P[L]
#pragma unroll 1
    for (u32 r = 0; r < rows; r++) {
        // the needle row: three scalar loads of the by-value argument
        const u32 ucw = ((const u32*)nd.uc)[r], ufw = ((const u32*)nd.uf)[r];
        const u32 cl = (((const u32*)nd.ulen)[r >> 2] >> (8 * (r & 3))) & 0xFF;
        const bool two = ucw != ufw;
        const u32 rbv = splat16((r + 1) * e);
        // keep the per-lane state out of loop-invariant hoisting (see the first form)
#pragma unroll
        for (int d = 0; d < NR; d++) FZB_OPAQUE_V(Qp[d]);
#pragma unroll
        for (int k = 0; k <= RB; k++) FZB_OPAQUE_V(hb[k]);
#pragma unroll
        for (int k = 0; k < RB; k++) FZB_OPAQUE_V(sflag[k]);  // (its lane masks are row-invariant: sixteen more registers if hoisted)
        // ---- byte-level match flags: scalar start && the cl bytes from the lane on equal the needle scalar (unicode.rs:221-241) ----
        u32 fe01[RB], fm01[RB];
        switch (two ? cl + 4 : cl) {
            case 1: unicode_row_flags<RB, 1, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 2: unicode_row_flags<RB, 2, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 3: unicode_row_flags<RB, 3, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 5: unicode_row_flags<RB, 1, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 6: unicode_row_flags<RB, 2, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 7: unicode_row_flags<RB, 3, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
            case 8: unicode_row_flags<RB, 4, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
            default: unicode_row_flags<RB, 4, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
        }
        u32 row[NR], pendg[NR];
#pragma unroll
        for (int k = 0; k < RB; k++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int d = 2 * k + h;
                const u32 sel01 = h ? 0x0c030c02u : 0x0c010c00u;  // flag byte 2h -> lane 0, byte 2h+1 -> lane 1, as 0 / 1 per 16-bit lane
                const u32 ex01 = __builtin_amdgcn_perm(0u, fe01[k], sel01);
                const u32 mm01 = __builtin_amdgcn_perm(0u, fm01[k], sel01);
                const u32 sst = p_neg_mask(__builtin_amdgcn_perm(0u, sflag[k], h ? 0x03030202u : 0x01010000u));  // 0xFFFF where the lane is a scalar start
                const u32 z = (r * e) << 16;  // T(r-1, lane -1) = B(r-1, -1) = r * e: the zero column
                const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : z, 16);
                const u32 t = p_subs(p_mad(mm01, bonus[d], sh), xqv);
                const u32 diag = p_mad(ex01, casev, t);
                const u32 up = p_subs(prev[d], upg[d]);  // upg[d] still holds the PREVIOUS row's match mask (x gop')
                const u32 Bd = p_mad(Qp[d], gexv, rbv);
                const u32 v = p_max(p_max(diag, up), Bd);
                row[d] = (v & sst) | (Bd & ~sst);
                pendg[d] = upg[d] = p_mul(mm01, gopmv);
            }
            if (k & 1) FZB_SCHED_FENCE();  // four lane pairs between fences: enough independent chains to fill the packed-op forwarding slots
        }
        if (r + 1 == rows) {  // the last row is not propagated: its maximum, unbiased (non-start lanes hold exactly B: 0)
            u32 mxl = 0;
#pragma unroll
            for (int d = 0; d < NR; d++) mxl = p_max(mxl, p_sub(row[d], p_mad(Qp[d], gexv, rbv)));
            return max(mxl & 0xFFFF, mxl >> 16);
        }
        // ---- propagate_horizontal_unicode_gaps, in place from the highest dword down (entry d reads entries <= d only) ---------------
#pragma unroll
        for (int d = NR - 1; d >= 0; d--) {  // shift by one lane: "a scalar start lies in (L-1, L]" <=> lane L is a scalar start
            const u32 bs = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : 0u, 16);
            const u32 ps = __builtin_amdgcn_alignbit(pendg[d], d ? pendg[d - 1] : 0u, 16);
            const u32 fl = p_neg_mask(__builtin_amdgcn_perm(0u, sflag[d >> 1], (d & 1) ? 0x03030202u : 0x01010000u));
            row[d] = p_max(row[d], p_subs(bs, ps & fl));
            pendg[d] = pendg[d] | (ps & ~fl);
            if ((d & 1) == 0) FZB_SCHED_FENCE();
        }
#pragma unroll
        for (int off = 1; off < NR; off *= 2) {
            if (UTF8 && off >= 2) {
#pragma unroll
                for (int d = NR - 1; d >= off; d--) row[d] = p_max(row[d], p_subs(row[d - off], pendg[d - off]));
                FZB_SCHED_FENCE();
                continue;
            }
#pragma unroll
            for (int d = NR - 1; d >= off; d--) {
                const u32 fl = p_neg_mask(p_sub(Qp[d - off], Qp[d]));  // Qp differs <=> a scalar start (or padding) lies in (L - 2 off, L]
                row[d] = p_max(row[d], p_subs(row[d - off], pendg[d - off] & fl));
                pendg[d] = pendg[d] | (pendg[d - off] & ~fl);
                if ((d & 3) == 0) FZB_SCHED_FENCE();
            }
        }
#pragma unroll
        for (int d = 0; d < NR; d++) prev[d] = row[d];
    }
    return 0;  // rows == 0
}

// window bytes th[0 .. m) from memory -> hb (zero padded, + the guard dword the shifted views read)
template <int RB>
__device__ __forceinline__ void unicode_load_bytes(const u8* __restrict__ th, u32 m, u32 (&hb)[RB + 1]) {
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const u32 p = 4 * k;
        u32 v = 0;
        if (p < m) {
            v = load_u32_unaligned(th, p);
            const u32 rem = m - p;
            if (rem < 4) v &= (1u << (8 * rem)) - 1;
        }
        hb[k] = v;
    }
    hb[RB] = 0;
}

// the whole scorer with the UTF-8 decision made by the caller (host harness; general kernel)
template <int SWL, int REAL = SWL / 2, bool UTF8 = false>
__device__ __forceinline__ u32 dp_unicode_single_chunk_t(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const u8* cls) {
    constexpr int RB = (REAL + 1) / 2, NR = 2 * RB;
    u32 hb[RB + 1], sflag[RB], Qp[NR], bonus[NR];
    unicode_load_bytes<RB>(th, m, hb);
    unicode_setup_t<SWL, REAL>(nd, hb, m, include_prefix, cls, sflag, Qp, bonus);
    return unicode_rows_t<SWL, REAL, UTF8>(nd, hb, sflag, Qp, bonus);
}

// ... and with the window's bytes already in registers and the UTF-8 decision taken here: `all_of(flag)` is the wave's vote (the kernel
// passes __all; the host harness the identity)
template <int SWL, int REAL, typename Vote>
__device__ __forceinline__ u32 dp_unicode_single_chunk_tr(const NeedleDev& nd, u32 (&hb)[(REAL + 1) / 2 + 1], u32 m, bool include_prefix, const u8* cls, const Vote& all_of) {
    constexpr int RB = (REAL + 1) / 2, NR = 2 * RB;
    u32 sflag[RB], Qp[NR], bonus[NR];
    const bool run4 = unicode_setup_t<SWL, REAL>(nd, hb, m, include_prefix, cls, sflag, Qp, bonus);
    return all_of(!run4) ? unicode_rows_t<SWL, REAL, true>(nd, hb, sflag, Qp, bonus) : unicode_rows_t<SWL, REAL, false>(nd, hb, sflag, Qp, bonus);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Windows wider than one chunk in the BIASED-THROUGHOUT form (round 4, second step): dp_unicode_multi_chunk's chunk walk with
// unicode_rows_t's arithmetic.  T(i, L) = S(i, L) + e * Qp[L] + (i + 1) * e, where Qp counts scalar starts and padding lanes from the
// ADJACENT HALF's first lane on (QpA over lanes -SWL/2 .. -1, Qp over the chunk): P is continuous across the chunk boundary, so the diagonal
// into lane 0 ("P grows by e on a scalar-start lane"), the crossing tests (Qp differs) and the biased steps hold unchanged when a source lane
// lies in the adjacent half.  Per row the parked top half of the previous chunk (unbiased values; pending as bits) is biased with
// e * QpA + (r + 1) * e and its pending bits become gop' masks; T(r - 1, lane -1) = carry + e * QpA[last] + r * e.  Parked again: the top half
// minus its bias, pending as "pendg != 0" (with gop' = 0 a pending charge is 0 anyway).  The last needle row is neither propagated nor parked
// in any chunk (see dp_unicode_multi_chunk).  UTF8 (the caller's wave-uniform vote over the WHOLE window, unicode_window_has_cont_run4): no four
// continuation bytes in a row anywhere, so every window of >= 4 lanes - across the boundary too - holds a scalar start or padding: the
// 4-lane and wider steps are a subtract and a max.  Preconditions: LaunchCfg::cfu_ok.  Fuzzed against the oracle and the first form
// (tests/test_kernel_math_host.py).
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool unicode_window_has_cont_run4(const u8* __restrict__ th, u32 m) {
    u32 run4 = 0, cont_prev = 0;
    for (u32 p = 0; p < m; p += 4) {
        const u32 w = load_u32_unaligned(th, p);
        const u32 nv = min(m - p, 4u);
        const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
        const u32 cv = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u) & validf;
        run4 |= cv & __builtin_amdgcn_alignbyte(cv, cont_prev, 1) & __builtin_amdgcn_alignbyte(cv, cont_prev, 2) & __builtin_amdgcn_alignbyte(cv, cont_prev, 3);
        cont_prev = cv;
    }
    return run4 != 0;
}

template <int SWL, bool UTF8>
__device__ __forceinline__ u32 dp_unicode_multi_chunk_t(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const u8* cls,
                                                        u32* __restrict__ scratch, u32 sstride, u32 sidx) {
    constexpr int NW = SWL / 2, NB = SWL / 4, HT = NW / 2;
    constexpr int NCHG = (HT + 7) / 8;
    static_assert(HT >= 1 && HT + NCHG <= NW, "a parked row must fit its NW dwords of the slab");
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 e = nd.gex;
    const u32 Mv = splat16(nd.match_plus_mismatch), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const u32 xqv = splat16(nd.mismatch - 2 * e), gexv = splat16(e), gopmv = splat16(nd.gopm), casev = splat16(nd.matching_case);
    const bool u8class = nd.lane_mask == 0xFF;
    const u32 nchunks = (m + SWL - 1) / SWL;
    const u32 mv = splat16(m);
    u32 mx = 0;
    u32 clsw_prev = 0;
#pragma unroll 1
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 cbase = ch * SWL;
        u32 hb[NB + 1];
#pragma unroll
        for (int k = 0; k <= NB; k++) {
            const u32 p = cbase + 4 * k;
            u32 v = 0;
            if (p < m) {
                v = load_u32_unaligned(th, p);
                const u32 rem = m - p;
                if (rem < 4) v &= (1u << (8 * rem)) - 1;
            }
            hb[k] = v;
        }
        // ---- prefix counts over [adjacent half | chunk], scalar-start flags, bonuses ----
        u32 QpA[HT], Qp[NW], bonus[NW], sflag[NB];
        u32 qrun = 0;
        if (ch) {
#pragma unroll
            for (int k = 0; k < NB / 2; k++) {
                const u32 w = load_u32_unaligned(th, cbase - SWL / 2 + 4 * k);
                const u32 t = (0x80808080u & ~zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u)) >> 7;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const u32 s0 = h ? (t >> 16) & 1 : t & 1, s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
                    const u32 q0 = qrun + s0, q1 = q0 + s1;
                    QpA[2 * k + h] = q0 | (q1 << 16);
                    qrun = q1;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < HT; t++) QpA[t] = 0;
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const u32 w = hb[k];
            const u32 contf = zflag4((w & 0xC0C0C0C0u) ^ 0x80808080u);
            const u32 p = cbase + 4 * k;
            const u32 nv = m > p ? min(m - p, 4u) : 0u;
            const u32 validf = nv >= 4 ? 0x80808080u : (0x80808080u & ((1u << (8 * nv)) - 1));
            sflag[k] = validf & ~contf;
            const u32 t = sflag[k] >> 7;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int d = 2 * k + h;
                const u32 b0 = h ? (w >> 16) & 0xFF : w & 0xFF;
                const u32 b1 = h ? w >> 24 : (w >> 8) & 0xFF;
                const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
                const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);
                const u32 cap01 = (clsw >> 1) & sh & ONE;
                const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;
                bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
                clsw_prev = clsw;
                const u32 s0 = h ? (t >> 16) & 1 : t & 1, s1 = h ? (t >> 24) & 1 : (t >> 8) & 1;
                const u32 q0 = qrun + s0, q1 = q0 + s1;
                const u32 lanepos1 = (cbase + (u32)(2 * d + 1)) | ((cbase + (u32)(2 * d + 2)) << 16);
                Qp[d] = p_add(q0 | (q1 << 16), p_subs(lanepos1, mv));  // + padding lanes up to and including the lane
                qrun = q1;
            }
        }
        if (ch == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
        u32 prev[NW], upg[NW];
#pragma unroll
        for (int d = 0; d < NW; d++) prev[d] = p_mul(Qp[d], gexv), upg[d] = 0;  // T(-1, L) = P[L]: the zero row
        u32 carry = 0;
#pragma unroll 1
        for (u32 r = 0; r < rows; r++) {
            const u32 ucw = ((const u32*)nd.uc)[r], ufw = ((const u32*)nd.uf)[r];
            const u32 cl = (((const u32*)nd.ulen)[r >> 2] >> (8 * (r & 3))) & 0xFF;
            const bool two = ucw != ufw;
            const u32 rbv = splat16((r + 1) * e);
#pragma unroll
            for (int d = 0; d < NW; d++) FZB_OPAQUE_V(Qp[d]);
#pragma unroll
            for (int k = 0; k <= NB; k++) FZB_OPAQUE_V(hb[k]);
#pragma unroll
            for (int k = 0; k < NB; k++) FZB_OPAQUE_V(sflag[k]);
            u32 fe01[NB], fm01[NB];
            switch (two ? cl + 4 : cl) {
                case 1: unicode_row_flags<NB, 1, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 2: unicode_row_flags<NB, 2, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 3: unicode_row_flags<NB, 3, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 5: unicode_row_flags<NB, 1, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 6: unicode_row_flags<NB, 2, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 7: unicode_row_flags<NB, 3, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
                case 8: unicode_row_flags<NB, 4, true>(hb, sflag, ucw, ufw, fe01, fm01); break;
                default: unicode_row_flags<NB, 4, false>(hb, sflag, ucw, ufw, fe01, fm01); break;
            }
            // T(r-1, lane -1): the previous chunk's last lane of the row above, biased (chunk 0: the zero column, r * e)
            const u32 z = (carry + e * (QpA[HT - 1] >> 16) + r * e) << 16;
            u32 row[NW], pendg[NW];
#pragma unroll
            for (int k = 0; k < NB; k++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int d = 2 * k + h;
                    const u32 sel01 = h ? 0x0c030c02u : 0x0c010c00u;
                    const u32 ex01 = __builtin_amdgcn_perm(0u, fe01[k], sel01);
                    const u32 mm01 = __builtin_amdgcn_perm(0u, fm01[k], sel01);
                    const u32 sst = p_neg_mask(__builtin_amdgcn_perm(0u, sflag[k], h ? 0x03030202u : 0x01010000u));
                    const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : z, 16);
                    const u32 t = p_subs(p_mad(mm01, bonus[d], sh), xqv);
                    const u32 diag = p_mad(ex01, casev, t);
                    const u32 up = p_subs(prev[d], upg[d]);
                    const u32 Bd = p_mad(Qp[d], gexv, rbv);
                    const u32 v = p_max(p_max(diag, up), Bd);
                    row[d] = (v & sst) | (Bd & ~sst);
                    pendg[d] = upg[d] = p_mul(mm01, gopmv);
                }
                if (k & 1) FZB_SCHED_FENCE();
            }
            if (r + 1 == rows) {  // the last row: its maximum, unbiased, in every chunk
#pragma unroll
                for (int d = 0; d < NW; d++) mx = p_max(mx, p_sub(row[d], p_mad(Qp[d], gexv, rbv)));
                break;
            }
            // ---- the previous chunk's top half of this row, biased; its pending bits as gop' masks ----
            u32 ca[HT], apendg[HT];
            u32* srow = scratch + (size_t)(r * NW) * sstride + sidx;
            u32 carry_next = 0;
            if (ch) {
                u32 arow[HT];
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) {
                        const u32 pk = srow[(size_t)t * sstride];
                        arow[2 * t] = __builtin_amdgcn_perm(0u, pk, 0x0c010c00u);
                        arow[2 * t + 1] = __builtin_amdgcn_perm(0u, pk, 0x0c030c02u);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) arow[t] = srow[(size_t)t * sstride];
                }
                u32 pch[NCHG];
#pragma unroll
                for (int a = 0; a < NCHG; a++) pch[a] = srow[(size_t)(HT + a) * sstride];
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    ca[t] = p_add(arow[t], p_mad(QpA[t], gexv, rbv));
                    apendg[t] = p_mul((pch[t / 8] >> (2 * (t % 8))) & ONE, gopmv);
                }
                carry_next = arow[HT - 1] >> 16;
            } else {
#pragma unroll
                for (int t = 0; t < HT; t++) ca[t] = 0u, apendg[t] = 0u;
            }
            // ---- the gap scan, in place from the highest dword down ----
#pragma unroll
            for (int d = NW - 1; d >= 0; d--) {
                const u32 bs = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : ca[HT - 1], 16);
                const u32 ps = __builtin_amdgcn_alignbit(pendg[d], d ? pendg[d - 1] : apendg[HT - 1], 16);
                const u32 fl = p_neg_mask(__builtin_amdgcn_perm(0u, sflag[d >> 1], (d & 1) ? 0x03030202u : 0x01010000u));
                row[d] = p_max(row[d], p_subs(bs, ps & fl));
                pendg[d] = pendg[d] | (ps & ~fl);
                if ((d & 1) == 0) FZB_SCHED_FENCE();
            }
#pragma unroll
            for (int off = 1; off < NW; off *= 2) {
                if (UTF8 && off >= 2) {
#pragma unroll
                    for (int d = NW - 1; d >= 0; d--) {
                        const bool adj = d < off;
                        const int si = adj ? HT + d - off : d - off;
                        row[d] = p_max(row[d], p_subs(adj ? ca[si] : row[si], adj ? apendg[si] : pendg[si]));
                    }
                    FZB_SCHED_FENCE();
                    continue;
                }
#pragma unroll
                for (int d = NW - 1; d >= 0; d--) {
                    const bool adj = d < off;
                    const int si = adj ? HT + d - off : d - off;
                    const u32 rs = adj ? ca[si] : row[si], psrc = adj ? apendg[si] : pendg[si], qsrc = adj ? QpA[si] : Qp[si];
                    const u32 fl = p_neg_mask(p_sub(qsrc, Qp[d]));
                    row[d] = p_max(row[d], p_subs(rs, psrc & fl));
                    pendg[d] = pendg[d] | (psrc & ~fl);
                    if ((d & 3) == 0) FZB_SCHED_FENCE();
                }
            }
#pragma unroll
            for (int d = 0; d < NW; d++) prev[d] = row[d];
            // ---- park the top half (unbiased) and its pending bits for the next chunk ----
            if (ch + 1 < nchunks) {
                u32 top[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) top[t] = p_sub(row[HT + t], p_mad(Qp[HT + t], gexv, rbv));
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) srow[(size_t)t * sstride] = __builtin_amdgcn_perm(top[2 * t + 1], top[2 * t], 0x06040200u);
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) srow[(size_t)t * sstride] = top[t];
                }
                u32 chg[NCHG] = {};
#pragma unroll
                for (int t = 0; t < HT; t++) chg[t / 8] |= p_min(pendg[HT + t], ONE) << (2 * (t % 8));
#pragma unroll
                for (int a = 0; a < NCHG; a++) srow[(size_t)(HT + a) * sstride] = chg[a];
            }
            carry = carry_next;
        }
    }
    return max(mx & 0xFFFF, mx >> 16);
}

// ---- 0-typo unicode window of a haystack of at most 32 bytes held in two vectors (the kernel for short corpora) --------------------------
// The same window as unicode_window_first_last (src/prefilter/algo/unicode.rs:118-219), found with SWAR compares: 0x80 in every byte at
// which the CL bytes of the scalar `cw` start, all eight dwords merged into one word with bit 8j + k = byte j of dword k (position 4k + j).
template <int CL>
__device__ __forceinline__ u32 unicode_scalar_positions(const u32 (&w)[9], u32 cw) {
    const u32 c0 = (cw & 0xFF) * 0x01010101u, c1 = ((cw >> 8) & 0xFF) * 0x01010101u, c2 = ((cw >> 16) & 0xFF) * 0x01010101u, c3 = (cw >> 24) * 0x01010101u;
    u32 y = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u32 z = w[k] ^ c0;
        if (CL > 1) z |= __builtin_amdgcn_alignbyte(w[k + 1], w[k], 1) ^ c1;
        if (CL > 2) z |= __builtin_amdgcn_alignbyte(w[k + 1], w[k], 2) ^ c2;
        if (CL > 3) z |= __builtin_amdgcn_alignbyte(w[k + 1], w[k], 3) ^ c3;
        y |= zflag4(z) >> (7 - k);
    }
    return y;
}
__device__ __forceinline__ u32 unicode_scalar_positions_cl(const u32 (&w)[9], u32 cw, u32 cl) {
    switch (cl) {  // wave-uniform
        case 1: return unicode_scalar_positions<1>(w, cw);
        case 2: return unicode_scalar_positions<2>(w, cw);
        case 3: return unicode_scalar_positions<3>(w, cw);
        default: return unicode_scalar_positions<4>(w, cw);
    }
}
__device__ __forceinline__ u32 unicode_valid_positions(u32 L, u32 len) { return pos32_valid(L, len); }
__device__ __forceinline__ u32 unicode_first_pos(u32 y) { return pos32_first(y); }
__device__ __forceinline__ void unicode_window_regs(const NeedleDev& nd, const uint4& q0, const uint4& q1, u32 L, u32& ws, u32& we) {
    const u32 n = (u32)nd.rows;
    const u32 la = nd.ulen[0], lz = nd.ulen[n - 1];
    const u32 a0 = ((const u32*)nd.uc)[0], a1 = ((const u32*)nd.uf)[0], z0 = ((const u32*)nd.uc)[n - 1], z1 = ((const u32*)nd.uf)[n - 1];
    const u32 w[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, 0u};
    u32 ya = unicode_scalar_positions_cl(w, a0, la);
    if (a1 != a0) ya |= unicode_scalar_positions_cl(w, a1, la);
    u32 yz = unicode_scalar_positions_cl(w, z0, lz);
    if (z1 != z0) yz |= unicode_scalar_positions_cl(w, z1, lz);
    ya &= unicode_valid_positions(L, la);
    yz &= unicode_valid_positions(L, lz);
    ws = ya ? unicode_first_pos(ya) : 0u;  // (no occurrence cannot happen for a survivor of the exact filter)
    // last occurrence: reversing the word maps bit 8j + k to 8(3-j) + (7-k), so "first" of the reversed word is 31 - last position
    we = yz ? 31u - unicode_first_pos(__builtin_bitreverse32(yz)) + lz : 0u;
}
// ... and a typo query's window (unicode_window_typos) of a haystack held in two vectors
__device__ __forceinline__ void unicode_window_typos_regs(const NeedleDev& nd, const uint4& q0, const uint4& q1, u32 L, u32 k, u32& ws, u32& we) {
    const u32 w[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, 0u};
    u32 head, tail_end;
    unicode_typo_block(nd, w, L, k, head, tail_end);
    ws = head ? unicode_first_pos(head) : 0u;
    we = tail_end ? tail_end : L;
}

