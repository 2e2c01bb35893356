// Multi-chunk Smith-Waterman, thread per haystack, in dp_cf.h's biased domain: windows of SWL < m <= 1024 bytes, chunk by chunk.
//
// Same reference code as dp_body.h's dp_multi_chunk (score_haystack, ascii.rs:10-158; propagate_horizontal_gaps with the adjacent
// chunk's row, ascii_gap.rs:11-105; the per-row columns of score_matrix / match_masks, matrix.rs), same parked-rows scheme (per needle
// row the top half of the previous chunk's final row, unbiased, and its gap-open charges as bits, in a global slab laid out
// [row][dword][thread]).  What changes is the arithmetic between two parks:
//   * T(i, L) = S(i, L) + (L + SWL + i + 1) * e inside a chunk (L = lane of the chunk; the offset SWL keeps the adjacent half-chunk's
//     lanes -SWL/2 .. -1 non-negative): diagonal (T(i-1, L-1) + match*bonus) (-) (x - 2e), up T(i-1, L) (-) o*match(i-1, L), one max
//     with the cell's bias for the floor at 0, gap steps as shift / subtract the source's charge / max - no bias added before and removed
//     after every row's scan as in the first form;
//   * row 0 needs no special case (T(-1, L) = (L + SWL) * e is the zero row);
//   * the LAST row of the LAST chunk is not propagated (only its maximum is read; dp_cf.h, 2.).
// Preconditions (host, LaunchCfg::cfm_ok): 2 * gap_extend <= mismatch_penalty, biased values (up to 192 lanes + the needle's rows of e) stay below 0x7C00 (p_max3_s).
// ND = NeedleLongRows: the same rows for a needle of any length (what dp_quad.h runs for k2d_dp_long_quad; here for tests/kernel_host).
// tests/test_kernel_math_host.py fuzzes it against the oracle and against the first form.
#pragma once
#include "dp_cf.h"

// is lane k (a lane of this chunk, or - negative - of the adjacent half-chunk) the padding-entry source of gap step s?  (dp_cf.h, 3.: step s
// takes lanes [P-s, P-s/2); the adjacent lanes never change during the scan, so for them the same interval is simply "reaches the padding with
// step s but not with step s/2")
constexpr bool cfm_entry_lane(int k, int s, int P) { return s == 1 ? k == P - 1 : (k >= P - s && k < P - s / 2); }

// One chunk of the window: R = dwords (lane pairs) that are computed.  R == SWL/2: every lane (any chunk; `last_chunk` says whether rows are
// parked for a next one).  R < SWL/2: the LAST chunk of a window whose tail fills at most 2R lanes - the NUL lanes behind them are not computed:
// dp_cf.h's closed form (3.) with two changes: the bias carries the chunk offset SWL (A = T (-) (target lane + SWL) * e), and for 2R < SWL/2
// the adjacent half-chunk's lanes [2R - SWL/2, 0) enter the padding directly with the widest step (they are entries like any other: the
// column they would walk down instead ends in the previous chunk's last row, which the maximum covers).  Needs a needle without NUL (pad_ok).
template <int SWL, bool UPPER, int R, typename ND = NeedleDev>
__device__ __forceinline__ void cfm_chunk(const ND& nd, const u8* __restrict__ th, u32 m, u32 ch, bool last_chunk_rt, bool include_prefix, const CfTables& tab,
                                          u32* __restrict__ scratch, u32 sstride, u32 sidx, u32 rpitch, u32& mx, u32& cprev) {
    constexpr int NW = SWL / 2;
    constexpr int HT = NW / 2;  // parked dwords per vector (top half)
    constexpr int P = 2 * R;
    constexpr bool PAD = R < NW;
    constexpr int NCHG = 1;  // parked word of gap-open flags: dword HT+t's two lanes at bits t and 16 + t (HT <= 16)
    static_assert(R >= 1 && R <= NW, "R");
    static_assert(HT <= 16 && HT + NCHG <= NW, "a parked row must fit its NW dwords of the slab");
    static_assert(!PAD || 4 * P <= 3 * SWL, "padding entries must land inside the chunk");
    const bool last_chunk = PAD ? true : last_chunk_rt;
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex, x = nd.mismatch, o = nd.gopm;
    const u32 ev = splat16(e), gopmv = splat16(o), casev = splat16(nd.matching_case), xqv = splat16(x - 2 * e);
    const bool u8class = nd.lane_mask == 0xFF;  // score values of the u8 class fit a byte (score_fits_in_u8)
    const u32 flg = u8class ? HT / 2 : HT;  // where a parked row keeps its flag word: behind the dwords it uses (a row is flg + 1 dwords)
    const u32 cbase = ch * SWL;
    u32 hw[R], bonus[R];
#pragma unroll
    for (int k = 0; k < (R + 1) / 2; k++) {
        const u32 p = cbase + 4 * k;
        u32 w = 0;
        if (p < m) {
            w = load_u32_unaligned(th, p);
            const u32 rem = m - p;
            if (rem < 4) w &= (1u << (8 * rem)) - 1;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int d = 2 * k + h;
            if (d < R) {
                hw[d] = __builtin_amdgcn_perm(0u, w, h ? 0x0c030c02u : 0x0c010c00u);
                const u32 c0 = tab.cls2[hw[d] & 0xFF], c1 = tab.cls2[hw[d] >> 16];
                const u32 i0 = (cprev << 2) | c0, i1 = (c0 << 2) | c1;
                bonus[d] = (u32) * (const u16*)((const u8*)tab.bon + i0) | ((u32) * (const u16*)((const u8*)tab.bon + i1) << 16);
                cprev = c1;
            }
        }
    }
    if (ch == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
    u32 T[R], g[R];
    {
        u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16);  // T(-1, L) = (L + SWL) * e
#pragma unroll
        for (int d = 0; d < R; d++, bias = fzb_sadd(bias, 2 * ev)) T[d] = bias, g[d] = 0;
    }
    u32 carry = 0;           // S(r-1, previous chunk's last lane)
    u32 acc0 = 0, acc1 = 0;  // the padding's running maximum (A domain: value + (row + 1) * e), two chains
    const u32 edc = 0xFFFFu | (((u32)(P - 2 + SWL) * e + x) << 16);  // the diagonal out of lane P-1: T(i-1, P-1) (-) ((P-2+SWL)*e + x), high lane only
#pragma unroll 1
    for (u32 r = 0; r < rows; r++) {
        const CfRow k = cf_row_consts(nd, r);
        const u32 rb = (r + 1) * ev;
        const u32 z = (carry + ((u32)SWL - 1 + r) * e) << 16;  // T(r-1, lane -1)
        if (PAD) acc0 = p_max(acc0, p_subs(T[R - 1], edc));   // (row 0: the zero row (-) more than its bias = 0)
        u32 b[R], gn[R];
        u32 chg[NCHG] = {};
        {
            u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;  // lanes 0, 1 of this row
#pragma unroll
            for (int d = 0; d < R; d++, bias = fzb_sadd(bias, 2 * ev)) {
                const u32 sh = __builtin_amdgcn_alignbit(T[d], d ? T[d - 1] : z, 16);
                u32 mm, mb;
                cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
                const u32 D = p_subs(p_add(sh, mb), xqv);
                const u32 U = p_subs(T[d], g[d]);
                b[d] = p_max3_s(D, U, bias);  // (all three below 0x7C00: cfm_ok)
                gn[d] = p_mul(mm, gopmv);
                if constexpr (!PAD)  // the top half's match flags for the next chunk (they are its gap-open charges): dword HT+t -> bits t, 16+t
                    if (d >= HT) chg[0] |= mm << (d - HT);
            }
        }
        if (last_chunk && r + 1 == rows) {  // only the row's maximum is read: no propagation
            u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;
#pragma unroll
            for (int d = 0; d < R; d++, bias = fzb_sadd(bias, 2 * ev)) mx = p_max(mx, p_subs(b[d], bias));
            if (PAD) mx = p_max(mx, p_subs(p_max(acc0, acc1), rb));  // A domain -> value in the last row
            break;
        }
        // the previous chunk's parked vectors for this row, biased as lanes -SWL/2 .. -1 of this chunk and already charged (what a gap step
        // reads); zero for the first chunk: nothing can flow in (0 < every bias)
        u32 ca[HT];
        u32* srow = scratch + (size_t)r * rpitch + sidx;
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
            pch[0] = srow[(size_t)flg * sstride];
            u32 bias = (u32)(SWL / 2) * e + (((u32)(SWL / 2) + 1) * e << 16) + rb;
#pragma unroll
            for (int t = 0; t < HT; t++, bias = fzb_sadd(bias, 2 * ev)) {
                const u32 ag = p_mul((pch[0] >> t) & 0x00010001u, gopmv);
                ca[t] = p_subs(p_add(arow[t], bias), ag);
            }
            carry_next = arow[HT - 1] >> 16;
        } else {
#pragma unroll
            for (int t = 0; t < HT; t++) ca[t] = 0u;
        }
        // ---- propagate_horizontal_gaps over [parked top half of the previous chunk | this chunk's computed lanes] -------------------
        const u32 lim = PAD ? (rows - 2 - r) * o : 0u;  // an entry of step s is dominated by its own column when s * e >= lim (dp_cf.h, 3.)
        {
            u32 cc[R], nb[R];
#pragma unroll
            for (int d = 0; d < R; d++) cc[d] = p_subs(b[d], gn[d]);
            if (PAD && e < lim) acc1 = p_max(acc1, p_subs(cc[R - 1], 0xFFFFu | (((u32)(P + SWL) * e) << 16)));  // lane P-1 -> lane P
#pragma unroll
            for (int d = 0; d < R; d++) nb[d] = p_max(b[d], __builtin_amdgcn_alignbit(cc[d], d ? cc[d - 1] : ca[HT - 1], 16));
#pragma unroll
            for (int d = 0; d < R; d++) b[d] = nb[d];
        }
#pragma unroll
        for (int off = 1; off < NW; off *= 2) {
            const int s = 2 * off;  // lanes
            if (PAD && (u32)s * e < lim) {
#pragma unroll
                for (int d = 0; d < R; d++) {
                    const bool in0 = cfm_entry_lane(2 * d, s, P), in1 = cfm_entry_lane(2 * d + 1, s, P);
                    if (in0 || in1) {
                        const u32 k0 = in0 ? (u32)(2 * d + s + SWL) * e : 0xFFFFu, k1 = in1 ? (u32)(2 * d + 1 + s + SWL) * e : 0xFFFFu;  // (target lane + SWL) * e
                        const u32 v = p_subs(p_subs(b[d], gn[d]), k0 | (k1 << 16));
                        if (d & 1) acc1 = p_max(acc1, v);
                        else acc0 = p_max(acc0, v);
                    }
                }
#pragma unroll
                for (int t = 0; t < HT; t++) {  // adjacent lanes -SWL/2 + 2t, +1
                    const int l0 = 2 * t - SWL / 2, l1 = l0 + 1;
                    const bool in0 = cfm_entry_lane(l0, s, P), in1 = cfm_entry_lane(l1, s, P);
                    if (in0 || in1) {
                        const u32 k0 = in0 ? (u32)(l0 + s + SWL) * e : 0xFFFFu, k1 = in1 ? (u32)(l1 + s + SWL) * e : 0xFFFFu;
                        const u32 v = p_subs(ca[t], k0 | (k1 << 16));
                        if (t & 1) acc1 = p_max(acc1, v);
                        else acc0 = p_max(acc0, v);
                    }
                }
            }
            u32 nb[R];
#pragma unroll
            for (int d = 0; d < R; d++) {
                const u32 src = d >= off ? p_subs(b[d - off], gn[d - off]) : ca[HT + d - off];
                nb[d] = p_max(b[d], src);
            }
#pragma unroll
            for (int d = 0; d < R; d++) b[d] = nb[d];
        }
        // park this chunk's top half (unbiased) for the next chunk; the last row also feeds the running maximum
        if constexpr (!PAD) if (!last_chunk) {
            u32 bias = (u32)(SWL + SWL / 2) * e + (((u32)(SWL + SWL / 2) + 1) * e << 16) + rb;
            u32 top[HT];
#pragma unroll
            for (int t = 0; t < HT; t++, bias = fzb_sadd(bias, 2 * ev)) top[t] = p_sub(b[HT + t], bias);
            if (u8class) {
#pragma unroll
                for (int t = 0; t < HT / 2; t++) srow[(size_t)t * sstride] = __builtin_amdgcn_perm(top[2 * t + 1], top[2 * t], 0x06040200u);
            } else {
#pragma unroll
                for (int t = 0; t < HT; t++) srow[(size_t)t * sstride] = top[t];
            }
            srow[(size_t)flg * sstride] = chg[0];
        }
        if (r + 1 == rows) {  // last row of a chunk that is not the last: its maximum, unbiased
            u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;
#pragma unroll
            for (int d = 0; d < R; d++, bias = fzb_sadd(bias, 2 * ev)) mx = p_max(mx, p_sub(b[d], bias));
        }
#pragma unroll
        for (int d = 0; d < R; d++) T[d] = b[d], g[d] = gn[d];
        carry = carry_next;
    }
}

// The same with the class of the last chunk's tail chosen at run time (wave-uniform; k2d_dp_multi_tc): 3 = all of it, c < 3 = (c + 1) * SWL/4
// computed lanes.  One copy of the full chunk's code serves every class.
template <int SWL, bool UPPER>
__device__ __forceinline__ u32 dp_multi_chunk_tc(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const CfTables& tab, u32* __restrict__ scratch,
                                                 u32 sstride, u32 sidx, u32 rpitch, u32 wcls) {
    constexpr int NW = SWL / 2, Q = NW / 4;
    const u32 nchunks = (m + SWL - 1) / SWL;
    const u32 nfull = wcls >= 3 ? nchunks : nchunks - 1;
    u32 mx = 0;
    u32 cprev = 0;
#pragma unroll 1
    for (u32 ch = 0; ch < nfull; ch++) cfm_chunk<SWL, UPPER, NW>(nd, th, m, ch, ch + 1 == nchunks, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    if (wcls == 2) cfm_chunk<SWL, UPPER, 3 * Q>(nd, th, m, nchunks - 1, true, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    else if (wcls == 1) cfm_chunk<SWL, UPPER, 2 * Q>(nd, th, m, nchunks - 1, true, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    else if (wcls == 0) cfm_chunk<SWL, UPPER, Q>(nd, th, m, nchunks - 1, true, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    return max(mx & 0xFFFF, mx >> 16);
}

// RL = computed dwords of the LAST chunk (SWL/2: all of it; less: the caller guarantees m - (nchunks - 1) * SWL <= 2 * RL and a needle without NUL)
template <int SWL, bool UPPER, int RL = SWL / 2, typename ND = NeedleDev>
__device__ __forceinline__ u32 dp_multi_chunk_t(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const CfTables& tab,
                                                u32* __restrict__ scratch, u32 sstride, u32 sidx, u32 rpitch) {
    constexpr int NW = SWL / 2;
    const u32 nchunks = (m + SWL - 1) / SWL;
    const u32 nfull = RL < NW ? nchunks - 1 : nchunks;
    u32 mx = 0;
    u32 cprev = 0;  // class (x 2) of the previous chunk's last lane; lane -1 of chunk 0: no delimiter, no lowercase letter
#pragma unroll 1
    for (u32 ch = 0; ch < nfull; ch++) cfm_chunk<SWL, UPPER, NW, ND>(nd, th, m, ch, ch + 1 == nchunks, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    if (RL < NW) cfm_chunk<SWL, UPPER, RL, ND>(nd, th, m, nchunks - 1, true, include_prefix, tab, scratch, sstride, sidx, rpitch, mx, cprev);
    return max(mx & 0xFFFF, mx >> 16);
}
