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
// Preconditions (host, LaunchCfg::cfm_ok): 2 * gap_extend <= mismatch_penalty, biased values (up to 192 lanes + 63 rows of e) fit 16 bits.
// tests/test_kernel_math_host.py fuzzes it against the oracle and against the first form.
#pragma once
#include "dp_cf.h"

template <int SWL, bool UPPER>
__device__ __forceinline__ u32 dp_multi_chunk_t(const NeedleDev& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const CfTables& tab,
                                                u32* __restrict__ scratch, u32 sstride, u32 sidx) {
    constexpr int NW = SWL / 2;
    constexpr int NB = SWL / 4;
    constexpr int HT = NW / 2;  // parked dwords per vector (top half)
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex, x = nd.mismatch;
    const u32 ev = splat16(e), gopmv = splat16(nd.gopm), casev = splat16(nd.matching_case), xqv = splat16(x - 2 * e);
    const u32 nchunks = (m + SWL - 1) / SWL;
    const bool u8class = nd.lane_mask == 0xFF;  // score values of the u8 class fit a byte (score_fits_in_u8)
    u32 mx = 0;
    u32 cprev = 0;  // class (x 2) of the previous chunk's last lane; lane -1 of chunk 0: no delimiter, no lowercase letter
#pragma unroll 1
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 cbase = ch * SWL;
        const bool last_chunk = ch + 1 == nchunks;
        u32 hw[NW], bonus[NW];
#pragma unroll
        for (int k = 0; k < NB; k++) {
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
                hw[d] = __builtin_amdgcn_perm(0u, w, h ? 0x0c030c02u : 0x0c010c00u);
                const u32 c0 = tab.cls2[hw[d] & 0xFF], c1 = tab.cls2[hw[d] >> 16];
                const u32 i0 = (cprev << 2) | c0, i1 = (c0 << 2) | c1;
                bonus[d] = (u32) * (const u16*)((const u8*)tab.bon + i0) | ((u32) * (const u16*)((const u8*)tab.bon + i1) << 16);
                cprev = c1;
            }
        }
        if (ch == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
        u32 T[NW], g[NW];
        {
            u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16);  // T(-1, L) = (L + SWL) * e
#pragma unroll
            for (int d = 0; d < NW; d++, bias = fzb_sadd(bias, 2 * ev)) T[d] = bias, g[d] = 0;
        }
        u32 carry = 0;  // S(r-1, previous chunk's last lane)
#pragma unroll 1
        for (u32 r = 0; r < rows; r++) {
            const CfRow k = cf_row_consts(nd, r);
            const u32 rb = (r + 1) * ev;
            const u32 z = (carry + ((u32)SWL - 1 + r) * e) << 16;  // T(r-1, lane -1)
            u32 b[NW], gn[NW];
            {
                u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;  // lanes 0, 1 of this row
#pragma unroll
                for (int d = 0; d < NW; d++, bias = fzb_sadd(bias, 2 * ev)) {
                    const u32 sh = __builtin_amdgcn_alignbit(T[d], d ? T[d - 1] : z, 16);
                    u32 mm, mb;
                    cf_match<UPPER>(k, hw[d], bonus[d], casev, mm, mb);
                    const u32 D = p_subs(p_add(sh, mb), xqv);
                    const u32 U = p_subs(T[d], g[d]);
                    b[d] = p_max(p_max(D, U), bias);
                    gn[d] = p_mul(mm, gopmv);
                }
            }
            if (last_chunk && r + 1 == rows) {  // only the row's maximum is read: no propagation
                u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;
#pragma unroll
                for (int d = 0; d < NW; d++, bias = fzb_sadd(bias, 2 * ev)) mx = p_max(mx, p_subs(b[d], bias));
                break;
            }
            // the previous chunk's parked vectors for this row (zero for the first chunk), biased as lanes -SWL/2 .. -1 of this chunk
            u32 ab[HT], ag[HT];
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
                const u32 bits = srow[(size_t)HT * sstride];
                u32 bias = (u32)(SWL / 2) * e + (((u32)(SWL / 2) + 1) * e << 16) + rb;
#pragma unroll
                for (int t = 0; t < HT; t++, bias = fzb_sadd(bias, 2 * ev)) {
                    ab[t] = p_add(arow[t], bias);
                    ag[t] = p_mul(((bits >> (2 * t)) & 1u) | (((bits >> (2 * t + 1)) & 1u) << 16), gopmv);
                }
                carry_next = arow[HT - 1] >> 16;
            } else {
#pragma unroll
                for (int t = 0; t < HT; t++) ab[t] = 0u, ag[t] = 0u;  // the zero column: nothing can flow in (0 (-) anything = 0 < every bias)
            }
            // ---- propagate_horizontal_gaps over [parked top half of the previous chunk | this chunk] --------------------------------
            {
                u32 cc[NW], nb[NW];
#pragma unroll
                for (int d = 0; d < NW; d++) cc[d] = p_subs(b[d], gn[d]);
                const u32 cadj = p_subs(ab[HT - 1], ag[HT - 1]);
#pragma unroll
                for (int d = 0; d < NW; d++) nb[d] = p_max(b[d], __builtin_amdgcn_alignbit(cc[d], d ? cc[d - 1] : cadj, 16));
#pragma unroll
                for (int d = 0; d < NW; d++) b[d] = nb[d];
            }
#pragma unroll
            for (int off = 1; off < NW; off *= 2) {
                u32 nb[NW];
#pragma unroll
                for (int d = 0; d < NW; d++) {
                    const u32 src = d >= off ? p_subs(b[d - off], gn[d - off]) : p_subs(ab[HT + d - off], ag[HT + d - off]);
                    nb[d] = p_max(b[d], src);
                }
#pragma unroll
                for (int d = 0; d < NW; d++) b[d] = nb[d];
            }
            // park this chunk's top half (unbiased) for the next chunk; the last row also feeds the running maximum
            if (!last_chunk) {
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
                u32 bits = 0;
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    const u32 m01 = p_min(gn[HT + t], 0x00010001u);  // 1 where the lane is charged
                    bits |= ((m01 & 1u) | ((m01 >> 15) & 2u)) << (2 * t);
                }
                srow[(size_t)HT * sstride] = bits;
            }
            if (r + 1 == rows) {  // last row of a chunk that is not the last: its maximum, unbiased
                u32 bias = (u32)SWL * e + (((u32)SWL + 1) * e << 16) + rb;
#pragma unroll
                for (int d = 0; d < NW; d++, bias = fzb_sadd(bias, 2 * ev)) mx = p_max(mx, p_sub(b[d], bias));
            }
#pragma unroll
            for (int d = 0; d < NW; d++) T[d] = b[d], g[d] = gn[d];
            carry = carry_next;
        }
    }
    return max(mx & 0xFFFF, mx >> 16);
}
