// Multi-chunk Smith-Waterman with FOUR lanes per window (round 6): dp_cfm.h's arithmetic - the biased domain T(i, L) = S(i, L) + (L + SWL + i + 1) * e
// in which `shift_right_padded::<k>` is a move and a gap step is subtract-the-source's-charge and max - spread over four lanes, sixteen windows
// per wavefront.
//
// Why: a thread-per-window wave (dp_cfm.h) walks ~ 600 (32-lane chunks) to 1 000 (64-lane) instructions per needle row and chunk alone.  That is
// the right trade for hundreds of thousands of windows; a queue of fewer than ~ 64 k windows leaves the chip with at most one wavefront per SIMD,
// which issues one instruction per four cycles whatever its kind (profiles/r06_quad.txt: 0.67 of the wave's cycles are issue, the rest
// waits nothing else fills), and the kernel's duration is ONE wave's dependent chain: 44 us for the 27 k windows of the paths-shaped list, 1 ms for
// the 50 k windows of an 80-row needle.  (Sixteen lanes per window - this round's first attempt, profiles/r06_coop.txt - cut the chain sixteen-fold
// but spent 38 instructions per dword on the reference's statements as they stand, more in total than the thread form, and only paid below 16 k
// windows.)  Here the total is about the thread form's (~ 150-240 instruction slots per row, chunk and sixteen windows) at four times the wavefronts.
//
// Lane layout: inside each 16-lane DPP row, window w (0..3) owns row-lanes w, w + 4, w + 8, w + 12 - its quad lane L = (lane >> 2) & 3 holds
// dwords NDW * L .. NDW * L + NDW - 1 of the chunk's NW = SWL / 2 packed-u16 dwords (NDW = SWL / 8).  "One quad lane to the left" is then
// row_shr:4 and "two" row_shr:8, and the row-lanes a shift pulls in from outside the row KEEP THE DESTINATION'S OLD VALUE (bound_ctrl off) - which
// is where the previous chunk's parked row goes: no select, no second move.
//
// What a chunk leaves for the next, per needle row: the top half of its final row ALREADY in the next chunk's frame and charged
// (ca[t] of dp_cfm.h: T (-) SWL * e (-) gap_open' * match) - 2 * NDW words written by quad lanes 2 and 3, read back by lanes 0 (all of them) and 1
// (the upper NDW) as the `old` operands of the shifts, no arithmetic on the way - plus one word, the uncharged top dword, whose high half is the
// diagonal's source for lane 0, column 0 of the next row.  Words of row r at park[(r * WORDS + word) * stride]: stride = 32 in the workgroup's LDS
// (k2_classes_all: constants, every access an immediate offset), = the grid's window slots in the global slab (k2d_dp_long_quad: requested one
// needle row ahead).  (A block per window with 16-byte vectors was slower there - 0.93 against 0.83 ms: sixteen windows, sixteen lines per request.)
//
// Every chunk is computed in full (no closed-form padding); the last row of the last chunk is not propagated (dp_cf.h, 2.).
// Reference code this computes (as dp_cfm.h): score_haystack, src/smith_waterman/algo/ascii.rs:10-158 (per-chunk loop 91-158, diagonal / up /
// max 118-133, the row maximum 152-156); propagate_horizontal_gaps with the adjacent chunk's row, ascii_gap.rs:11-105; the per-row columns of
// score_matrix / match_masks that carry a chunk's last row to the next, src/smith_waterman/matrix.rs.
// Preconditions: LaunchCfg::cfm_ok.  Parity: tests/test_kernel_math_host.py (four host threads in lockstep against the oracle, no GPU),
// tests/test_gpu_quad.py, tests/test_gpu_long_needles.py, tests/test_gpu_knobs.py.
#pragma once
#include "dp_cf.h"

template <int K>
__device__ __forceinline__ u32 quad_shr(u32 old, u32 src) {  // row-lane i <- src of row-lane i - K; row-lanes i < K keep `old`
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x110 + K, 0xF, 0xF, false);
}
template <int K>
__device__ __forceinline__ u32 quad_ror(u32 src) {  // row-lane i <- src of row-lane (i - K) mod 16
    return (u32)__builtin_amdgcn_update_dpp(0, (int)src, 0x120 + K, 0xF, 0xF, false);
}
// words a window parks per needle row; the LDS layout's stride
template <int SWL>
struct QuadPark {
    static constexpr int NDW = SWL / 8;
    static constexpr int WORDS = 2 * NDW + 1;
    static constexpr u32 WPB = 32;  // windows per 128-thread workgroup
};

// The window th[0 .. m) (1 <= m <= 1024) scored by the four lanes of the caller's quad (see the layout above); every lane returns the score.
// park: the window's column of the parked rows (SLAB: global slab of `nslots` columns, else the workgroup's LDS area).
template <int SWL, bool UPPER, bool SLAB, typename ND>
__device__ __forceinline__ u32 dp_quad_window(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const CfTables& tab, u32* __restrict__ park, u32 nslots) {
    constexpr bool AHEAD = SLAB;
    const u32 pstride = SLAB ? nslots : QuadPark<SWL>::WPB, rpitch = (u32)QuadPark<SWL>::WORDS * pstride;  // (LDS: constants, every access an immediate offset)
    static_assert(SWL == 64 || SWL == 32, "four lanes x 8 or 4 dwords");
    constexpr int NDW = SWL / 8;  // dwords per lane
    constexpr int CPL = SWL / 4;  // columns (bytes) per lane
    constexpr int NP = 2 * NDW;   // parked value words per row
    const u32 L = (threadIdx.x >> 2) & 3u;
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex, x = nd.mismatch, o = nd.gopm;
    const u32 ev = splat16(e), gopmv = splat16(o), casev = splat16(nd.matching_case), xqv = splat16(x - 2 * e), swlev = splat16((u32)SWL * e);
    const u32 nchunks = (m + SWL - 1) / SWL;
    u32 bias0[NDW];  // T(-1, .) of this lane's dwords: (SWL + column) * e
#pragma unroll
    for (int j = 0; j < NDW; j++) {
        const u32 col = 2 * ((u32)NDW * L + j);
        bias0[j] = ((u32)SWL + col) * e + ((((u32)SWL + col + 1) * e) << 16);
    }
    // where this lane reads the previous chunk's words from: the 1-lane shifts' `old` (lane 0: the upper NDW words) and the 2-lane shift's
    // (lane 0: the lower NDW, lane 1: the upper NDW); lanes that never keep an `old` read what lane 0 reads
    const u32 offa = (u32)NDW, offb = L == 1 ? (u32)NDW : 0u;
    u32 mx = 0;
#pragma unroll 1
    for (u32 ch = 0; ch < nchunks; ch++) {
        const bool last_chunk = ch + 1 == nchunks;
        const u32 p0 = ch * SWL + L * CPL;
        u32 hw[NDW], bonus[NDW];
        {
            u32 cprev = 0;  // class (x 2) of the column to the left of this lane's first; column -1 of the window: no delimiter, no lowercase letter
            if (p0) cprev = tab.cls2[p0 - 1 < m ? th[p0 - 1] : 0];
#pragma unroll
            for (int k = 0; k < CPL / 4; k++) {
                const u32 p = p0 + 4 * k;
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
            if (p0 == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);  // first_lane(prefix_bonus): column 0 of the window
        }
        u32 T[NDW], g[NDW];
#pragma unroll
        for (int j = 0; j < NDW; j++) T[j] = bias0[j], g[j] = 0;
        u32 pw[NP + 1];  // the previous chunk's words of the row at hand, as this lane reads them: [0, NDW) `old` of the 2-lane shift, [NDW, 2 NDW) of the 1-lane shifts, [NP] the uncharged top dword
#pragma unroll
        for (int t = 0; t <= NP; t++) pw[t] = 0;
        const bool reads = ch != 0 && L < 2;
        auto park_load = [&](u32 r) {
            const u32* srow = park + (size_t)r * rpitch;
#pragma unroll
            for (int j = 0; j < NDW; j++) pw[j] = srow[(size_t)(offb + j) * pstride];
#pragma unroll
            for (int j = 0; j < NDW; j++) pw[NDW + j] = srow[(size_t)(offa + j) * pstride];
            pw[NP] = srow[(size_t)NP * pstride];
        };
        if (AHEAD && reads) park_load(0);
        u32 zc = (((u32)SWL - 1) * e) << 16;  // the uncharged top dword of the previous chunk's row r - 1 (its high half: T(r - 1, column -1) in this chunk's frame; row -1: the zero row)
#pragma unroll 1
        for (u32 r = 0; r < rows; r++) {
            const CfRow k = cf_row_consts(nd, r);
            const u32 rb = (r + 1) * ev;
            const u32 z = ch ? zc : (((u32)SWL - 1 + r) * e) << 16;
            const u32 left = quad_shr<4>(z, T[NDW - 1]);
            u32 b[NDW], gn[NDW], br[NDW];
#pragma unroll
            for (int j = 0; j < NDW; j++) {
                br[j] = p_add(bias0[j], rb);
                const u32 sh = __builtin_amdgcn_alignbit(T[j], j ? T[j - 1] : left, 16);
                u32 mm, mb;
                cf_match<UPPER>(k, hw[j], bonus[j], casev, mm, mb);
                const u32 D = p_subs(p_add(sh, mb), xqv);
                const u32 U = p_subs(T[j], g[j]);
                b[j] = p_max3_s(D, U, br[j]);  // (all three below 0x7C00: cfm_ok)
                gn[j] = p_mul(mm, gopmv);
            }
            if (last_chunk && r + 1 == rows) {  // only the row's maximum is read: no propagation
#pragma unroll
                for (int j = 0; j < NDW; j++) mx = p_max(mx, p_subs(b[j], br[j]));
                break;
            }
            // the previous chunk's row r in this chunk's frame, charged (zero in the first chunk: nothing can flow in, 0 < every bias)
            if (!AHEAD && reads) park_load(r);
            u32 pa[NDW], pb[NDW];
#pragma unroll
            for (int j = 0; j < NDW; j++) pb[j] = pw[j], pa[j] = pw[NDW + j];
            zc = pw[NP];
            if (AHEAD && reads) {  // (behind the last use of this row's words: the same registers take the next row's)
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
                for (int j = 0; j < NDW; j++) asm volatile("" : "+v"(pa[j]), "+v"(pb[j]) : : "memory");
                asm volatile("" : "+v"(zc) : : "memory");
#endif
                if (r + 1 < rows) park_load(r + 1);
            }
            // ---- propagate_horizontal_gaps over [top half of the previous chunk | this chunk] ---------------------------------------------
            {  // one column
                u32 cc[NDW], nb[NDW];
#pragma unroll
                for (int j = 0; j < NDW; j++) cc[j] = p_subs(b[j], gn[j]);
                const u32 lc = quad_shr<4>(pa[NDW - 1], cc[NDW - 1]);
#pragma unroll
                for (int j = 0; j < NDW; j++) nb[j] = p_max(b[j], __builtin_amdgcn_alignbit(cc[j], j ? cc[j - 1] : lc, 16));
#pragma unroll
                for (int j = 0; j < NDW; j++) b[j] = nb[j];
            }
#pragma unroll
            for (int off = 1; off <= 2 * NDW; off *= 2) {  // dwords
                u32 cc[NDW], src[NDW];
#pragma unroll
                for (int j = 0; j < NDW; j++) cc[j] = p_subs(b[j], gn[j]);
#pragma unroll
                for (int j = 0; j < NDW; j++) {
                    if (off == 2 * NDW) src[j] = quad_shr<8>(pb[j], cc[j]);
                    else if (off == NDW) src[j] = quad_shr<4>(pa[j], cc[j]);
                    else src[j] = j >= off ? cc[j - off] : quad_shr<4>(pa[NDW + j - off], cc[NDW + j - off]);
                }
#pragma unroll
                for (int j = 0; j < NDW; j++) b[j] = p_max(b[j], src[j]);
            }
            if (!last_chunk) {
                if (L >= 2) {  // this row for the next chunk: its frame (column - SWL), charged; and the uncharged top dword
                    const u32 wbase = (L - 2) * (u32)NDW;
                    u32* srow = park + (size_t)r * rpitch;
#pragma unroll
                    for (int j = 0; j < NDW; j++) srow[(size_t)(wbase + j) * pstride] = p_subs(p_sub(b[j], swlev), gn[j]);
                    if (L == 3) srow[(size_t)NP * pstride] = p_sub(b[NDW - 1], swlev);
                }
                if (r + 1 == rows) {  // last row of a chunk that is not the last: its maximum, unbiased
#pragma unroll
                    for (int j = 0; j < NDW; j++) mx = p_max(mx, p_sub(b[j], br[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < NDW; j++) T[j] = b[j], g[j] = gn[j];
        }
    }
    mx = max(mx & 0xFFFFu, mx >> 16);
    mx = max(mx, quad_ror<4>(mx));
    mx = max(mx, quad_ror<8>(mx));
    return mx;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same window ROW by row instead of chunk by chunk, for windows of at most MAXC chunks: every chunk's previous row (T, charges), bytes and
// bonuses stay in registers (4 * MAXC * NDW of them), so NOTHING is parked - what chunk c needs of chunk c - 1 in row r (its final row's top half,
// charged and in c's frame; its last dword of row r - 1 for the diagonal) was computed a moment ago by quad lanes 2 and 3 and comes over by a row
// ROTATE (row_ror:4 / :8 - lane 0 <- lane 3, lanes 0, 1 <- lanes 2, 3) into the `old` operand of the same shifts.  The chunk-by-chunk form's slab
// traffic (72 bytes per window, row and chunk: 1.5 GB per query of the bench's 80-row needle) and its loads ahead are gone; the arithmetic is the
// same statement for statement.  Windows of more chunks take dp_quad_window.
template <int SWL, bool UPPER, int MAXC, typename ND>
__device__ __forceinline__ u32 dp_quad_rows(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const CfTables& tab) {
    static_assert(SWL == 64 || SWL == 32, "four lanes x 8 or 4 dwords");
    constexpr int NDW = SWL / 8, CPL = SWL / 4;
    const u32 L = (threadIdx.x >> 2) & 3u;
    const u32 rows = (u32)nd.rows;
    const u32 e = nd.gex, x = nd.mismatch, o = nd.gopm;
    const u32 ev = splat16(e), gopmv = splat16(o), casev = splat16(nd.matching_case), xqv = splat16(x - 2 * e), swlev = splat16((u32)SWL * e);
    const u32 nchunks = (m + SWL - 1) / SWL;  // <= MAXC (the caller's business)
    u32 bias0[NDW];
#pragma unroll
    for (int j = 0; j < NDW; j++) {
        const u32 col = 2 * ((u32)NDW * L + j);
        bias0[j] = ((u32)SWL + col) * e + ((((u32)SWL + col + 1) * e) << 16);
    }
    u32 hw[MAXC][NDW], bonus[MAXC][NDW], T[MAXC][NDW], g[MAXC][NDW];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
#pragma unroll
        for (int j = 0; j < NDW; j++) hw[c][j] = 0, bonus[c][j] = 0, T[c][j] = bias0[j], g[c][j] = 0;
        if ((u32)c < nchunks) {
            const u32 p0 = (u32)c * SWL + L * CPL;
            u32 cprev = 0;
            if (p0) cprev = tab.cls2[p0 - 1 < m ? th[p0 - 1] : 0];
#pragma unroll
            for (int k = 0; k < CPL / 4; k++) {
                const u32 p = p0 + 4 * k;
                u32 w = 0;
                if (p < m) {
                    w = load_u32_unaligned(th, p);
                    const u32 rem = m - p;
                    if (rem < 4) w &= (1u << (8 * rem)) - 1;
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int d = 2 * k + h;
                    hw[c][d] = __builtin_amdgcn_perm(0u, w, h ? 0x0c030c02u : 0x0c010c00u);
                    const u32 c0 = tab.cls2[hw[c][d] & 0xFF], c1 = tab.cls2[hw[c][d] >> 16];
                    const u32 i0 = (cprev << 2) | c0, i1 = (c0 << 2) | c1;
                    bonus[c][d] = (u32) * (const u16*)((const u8*)tab.bon + i0) | ((u32) * (const u16*)((const u8*)tab.bon + i1) << 16);
                    cprev = c1;
                }
            }
            if (p0 == 0 && include_prefix) bonus[c][0] = p_add(bonus[c][0], (u32)nd.prefix);
        }
    }
    u32 mx = 0;
#pragma unroll 1
    for (u32 r = 0; r < rows; r++) {
        const CfRow k = cf_row_consts(nd, r);
        const u32 rb = (r + 1) * ev;
        const bool last_row = r + 1 == rows;
        u32 br[NDW];
#pragma unroll
        for (int j = 0; j < NDW; j++) br[j] = p_add(bias0[j], rb);
        u32 dz = 0;        // the previous chunk's last dword of row r - 1 in this chunk's frame (quad lane 3's copy is the one that is read)
        u32 ccp[NDW];      // the previous chunk's final row r in this chunk's frame, charged
#pragma unroll
        for (int j = 0; j < NDW; j++) ccp[j] = 0;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            if ((u32)c < nchunks) {
                const bool last_chunk = (u32)c + 1 == nchunks;
                const u32 z = c ? quad_ror<4>(dz) : (((u32)SWL - 1 + r) * e) << 16;
                const u32 left = quad_shr<4>(z, T[c][NDW - 1]);
                dz = p_sub(T[c][NDW - 1], swlev);
                u32 b[NDW], gn[NDW];
#pragma unroll
                for (int j = 0; j < NDW; j++) {
                    const u32 sh = __builtin_amdgcn_alignbit(T[c][j], j ? T[c][j - 1] : left, 16);
                    u32 mm, mb;
                    cf_match<UPPER>(k, hw[c][j], bonus[c][j], casev, mm, mb);
                    const u32 D = p_subs(p_add(sh, mb), xqv);
                    const u32 U = p_subs(T[c][j], g[c][j]);
                    b[j] = p_max3_s(D, U, br[j]);
                    gn[j] = p_mul(mm, gopmv);
                }
                if (last_chunk && last_row) {  // only the row's maximum is read: no propagation
#pragma unroll
                    for (int j = 0; j < NDW; j++) mx = p_max(mx, p_subs(b[j], br[j]));
                } else {
                    u32 pa[NDW], pb[NDW];
#pragma unroll
                    for (int j = 0; j < NDW; j++) pa[j] = c ? quad_ror<4>(ccp[j]) : 0u, pb[j] = c ? quad_ror<8>(ccp[j]) : 0u;
                    {  // one column
                        u32 cc[NDW], nb[NDW];
#pragma unroll
                        for (int j = 0; j < NDW; j++) cc[j] = p_subs(b[j], gn[j]);
                        const u32 lc = quad_shr<4>(pa[NDW - 1], cc[NDW - 1]);
#pragma unroll
                        for (int j = 0; j < NDW; j++) nb[j] = p_max(b[j], __builtin_amdgcn_alignbit(cc[j], j ? cc[j - 1] : lc, 16));
#pragma unroll
                        for (int j = 0; j < NDW; j++) b[j] = nb[j];
                    }
#pragma unroll
                    for (int off = 1; off <= 2 * NDW; off *= 2) {  // dwords
                        u32 cc[NDW], src[NDW];
#pragma unroll
                        for (int j = 0; j < NDW; j++) cc[j] = p_subs(b[j], gn[j]);
#pragma unroll
                        for (int j = 0; j < NDW; j++) {
                            if (off == 2 * NDW) src[j] = quad_shr<8>(pb[j], cc[j]);
                            else if (off == NDW) src[j] = quad_shr<4>(pa[j], cc[j]);
                            else src[j] = j >= off ? cc[j - off] : quad_shr<4>(pa[NDW + j - off], cc[NDW + j - off]);
                        }
#pragma unroll
                        for (int j = 0; j < NDW; j++) b[j] = p_max(b[j], src[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NDW; j++) ccp[j] = p_subs(p_sub(b[j], swlev), gn[j]);
                    if (last_row) {  // last row of a chunk that is not the last: its maximum, unbiased
#pragma unroll
                        for (int j = 0; j < NDW; j++) mx = p_max(mx, p_sub(b[j], br[j]));
                    }
#pragma unroll
                    for (int j = 0; j < NDW; j++) T[c][j] = b[j], g[c][j] = gn[j];
                }
            }
        }
    }
    mx = max(mx & 0xFFFFu, mx >> 16);
    mx = max(mx, quad_ror<4>(mx));
    mx = max(mx, quad_ror<8>(mx));
    return mx;
}
