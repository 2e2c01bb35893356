// Sub-wave cooperative Smith-Waterman (round 6): SIXTEEN lanes - one DPP row - score one window, four windows per wavefront.
//
// Why: the thread-per-window scorers (dp_cf.h / dp_cfm.h) are built for throughput - a thread walks its window's chunks and needle rows alone,
// ~ 1 000 wave-instructions per (row, chunk), which is the right trade when there are hundreds of thousands of windows.  A SHORT queue (the
// 27 k two- and three-chunk windows of the paths-shaped list: 426 wavefronts on 1 024 SIMDs) leaves most of the chip idle while every wave
// walks a dependent chain of ~ 12 000 instructions: 44.6 us for that slice, one wave's instruction latency.  Here lane l of a row holds the
// CPL = SWL / 16 columns CPL * l .. CPL * l + CPL - 1 of the emulated backend's SWL-lane vector as CPL / 2 packed-u16 dwords, so a (row, chunk)
// is ~ 130 instructions and four times as many wavefronts share the work.  The total instruction count is still above the thread form's (no
// biased domain, no closed-form padding, every shift a v_mov_dpp or two), which is why the DEVICE picks this form only below a queue length
// (kernels_dp.hip, k2_classes_all; profiles/r06_coop.txt has the measurements on both sides of it).
//
// The arithmetic is the reference's, statement by statement (score_haystack src/smith_waterman/algo/ascii.rs:10-158, propagate_horizontal_gaps
// ascii_gap.rs:11-105), in packed u16 (u8-class lanes never wrap: score_fits_in_u8, src/smith_waterman/mod.rs:92-116; u16 adds wrap as the
// reference's do).  `shift_right_padded::<K>` is
//   * K = 1: v_alignbit of each dword with its left neighbour dword (the lane's own, or the left lane's last by row_shr:1);
//   * K >= 2: a move by K / 2 dwords = whole lanes (DPP row_shr) and, for K = 2 at four columns per lane, one dword inside the lane;
//   the lanes a shift leaves empty take the previous chunk's final row by a row ROTATE (row_ror) of the lane's parked copy - no LDS round trip,
//   no wave-wide shuffle; in the window's first chunk that row is zero and the rotate is not issued.
// What a (row, chunk) leaves for the next chunk - the row's final values and its match flags - is PRIVATE to the lane (lane l of the next chunk
// needs exactly columns CPL * l ..), so it is parked per thread (LDS, [row][word][thread]) and needs no barrier.
#pragma once
#include "dp_body.h"

template <int K>
__device__ __forceinline__ u32 coop_shr(u32 old, u32 src) {  // lane i <- src of lane i - K inside its row of 16; lanes i < K keep `old`
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x110 + K, 0xF, 0xF, false);
}
template <int K>
__device__ __forceinline__ u32 coop_ror(u32 src) {  // lane i <- src of lane (i - K) mod 16
    return (u32)__builtin_amdgcn_update_dpp(0, (int)src, 0x120 + K, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 coop_subs(u32 a, u32 b) { return a > b ? a - b : 0; }

// words a thread parks per needle row: CPL / 2 dwords of values + one word of match flags
template <int SWL>
struct CoopPark {
    static constexpr int CPL = SWL / 16;
    static constexpr int NDW = CPL / 2;
    static constexpr int WORDS = NDW + 1;
};
__device__ __forceinline__ u32 coop_mad(u32 a, u32 b, u32 c) { return as_u32(as_us2(a) * as_us2(b) + as_us2(c)); }
// 1 per 16-bit lane where x == y, else 0
__device__ __forceinline__ u32 coop_eq01(u32 x, u32 y) { return p_min(x ^ y, 0x00010001u) ^ 0x00010001u; }
// the left neighbour lane's copy of `x`; lane 0 of the row gets lane 15's copy of `adjx` (the previous chunk) - zero in a window's first chunk
template <bool FIRST>
__device__ __forceinline__ u32 coop_left(u32 x, u32 adjx) { return coop_shr<1>(FIRST ? 0u : coop_ror<1>(adjx), x); }
template <int KL, bool FIRST>
__device__ __forceinline__ u32 coop_lanes(u32 x, u32 adjx) { return coop_shr<KL>(FIRST ? 0u : coop_ror<KL>(adjx), x); }

// One gap step of the log-step scan, K = the shift in columns.  row / adjv: this lane's dwords of the row being propagated / of the previous
// chunk's final row; mm / amm: the row's match flags (0 / 1 per 16-bit lane) of this chunk / of the previous one; kgv = K * gap_extend.
template <int SWL, int K, bool FIRST>
__device__ __forceinline__ void coop_gap_step(u32 (&row)[SWL / 32], const u32 (&adjv)[SWL / 32], const u32 (&mm)[SWL / 32], const u32 (&amm)[SWL / 32], u32 kgv, u32 gopmv) {
    constexpr int NDW = SWL / 32;
    u32 srow[NDW], smm[NDW];
    if constexpr (K == 1) {
        const u32 lr = coop_left<FIRST>(row[NDW - 1], adjv[NDW - 1]), lm = coop_left<FIRST>(mm[NDW - 1], amm[NDW - 1]);
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            srow[d] = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : lr, 16);
            smm[d] = __builtin_amdgcn_alignbit(mm[d], d ? mm[d - 1] : lm, 16);
        }
    } else if constexpr ((K / 2) % NDW != 0) {  // (four columns per lane, K = 2: one dword)
        srow[0] = coop_left<FIRST>(row[NDW - 1], adjv[NDW - 1]);
        smm[0] = coop_left<FIRST>(mm[NDW - 1], amm[NDW - 1]);
#pragma unroll
        for (int d = 1; d < NDW; d++) srow[d] = row[d - 1], smm[d] = mm[d - 1];
    } else {
        constexpr int KL = (K / 2) / NDW;  // whole lanes
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            srow[d] = coop_lanes<KL, FIRST>(row[d], adjv[d]);
            smm[d] = coop_lanes<KL, FIRST>(mm[d], amm[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < NDW; d++) row[d] = p_max(row[d], p_subs(srow[d], coop_mad(smm[d], gopmv, kgv)));
}

// the needle rows of one chunk; FIRST = the window's first chunk (nothing parked yet: the previous chunk is the zero chunk)
template <int SWL, bool FIRST, typename ND>
__device__ __forceinline__ u32 coop_chunk_rows(const ND& nd, const u32 (&hw)[SWL / 32], const u32 (&bonus)[SWL / 32], bool park_out, u32* __restrict__ park, u32 pstride) {
    constexpr int NDW = SWL / 32;
    constexpr int WORDS = NDW + 1;
    const u32 ONE = 0x00010001u;
    const u32 rows = (u32)nd.rows;
    const u32 Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm), casev = splat16(nd.matching_case);
    u32 prev[NDW], upg[NDW], row[NDW];
#pragma unroll
    for (int d = 0; d < NDW; d++) prev[d] = 0, upg[d] = 0, row[d] = 0;
    u32 carry = 0;  // S(r - 1, previous chunk), this lane's last dword: lane 15's copy holds the diagonal source of lane 0, column 0
    for (u32 r = 1; r <= rows; r++) {
        const u32 cr = nd.c[r - 1], fr = nd.f[r - 1];
        const u32 crv = splat16(cr), frv = splat16(fr);
        u32 ex[NDW], mm[NDW];
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            ex[d] = coop_eq01(hw[d], crv);
            mm[d] = cr != fr ? (ex[d] | coop_eq01(hw[d], frv)) : ex[d];
        }
        // what this lane parked for row r in the previous chunk
        u32 adjv[NDW], amm[NDW];
#pragma unroll
        for (int d = 0; d < NDW; d++) adjv[d] = 0, amm[d] = 0;
        if (!FIRST) {
#pragma unroll
            for (int d = 0; d < NDW; d++) adjv[d] = park[(r * WORDS + d) * pstride];
            const u32 pm = park[(r * WORDS + NDW) * pstride];
#pragma unroll
            for (int d = 0; d < NDW; d++) amm[d] = (pm >> d) & ONE;
        }
        // diagonal (ascii.rs:118-127) and up (:130-133)
        const u32 lp = coop_left<FIRST>(prev[NDW - 1], carry);
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : lp, 16);
            const u32 diag = coop_mad(ex[d], casev, p_subs(coop_mad(mm[d], bonus[d], sh), Xv));
            const u32 up = p_subs(p_subs(prev[d], gexv), upg[d]);
            row[d] = p_max(diag, up);
        }
        // propagate_horizontal_gaps (ascii_gap.rs:11-105): shifts 1, 2, 4, ... SWL / 2, each over the row as the previous step left it
        u32 kgv = gexv;
        coop_gap_step<SWL, 1, FIRST>(row, adjv, mm, amm, kgv, gopmv); kgv = p_add(kgv, kgv);
        coop_gap_step<SWL, 2, FIRST>(row, adjv, mm, amm, kgv, gopmv); kgv = p_add(kgv, kgv);
        coop_gap_step<SWL, 4, FIRST>(row, adjv, mm, amm, kgv, gopmv); kgv = p_add(kgv, kgv);
        coop_gap_step<SWL, 8, FIRST>(row, adjv, mm, amm, kgv, gopmv); kgv = p_add(kgv, kgv);
        coop_gap_step<SWL, 16, FIRST>(row, adjv, mm, amm, kgv, gopmv); kgv = p_add(kgv, kgv);
        if constexpr (SWL == 64) coop_gap_step<SWL, 32, FIRST>(row, adjv, mm, amm, kgv, gopmv);
        carry = adjv[NDW - 1];
        if (park_out) {
            u32 pm = 0;
#pragma unroll
            for (int d = 0; d < NDW; d++) {
                park[(r * WORDS + d) * pstride] = row[d];
                pm |= mm[d] << d;
            }
            park[(r * WORDS + NDW) * pstride] = pm;
        }
#pragma unroll
        for (int d = 0; d < NDW; d++) prev[d] = row[d], upg[d] = p_mul(mm[d], gopmv);
    }
    u32 mx = 0;
#pragma unroll
    for (int d = 0; d < NDW; d++) mx = p_max(mx, row[d]);  // every lane of the chunk, padding included (ascii.rs:152-156)
    return mx;
}

// The window th[0 .. m) (1 <= m <= 1024, any number of chunks) scored by the 16 lanes of the caller's DPP row; every lane returns the score.
// park: this thread's slot of the per-row parking area, word w of row r at park[(r * WORDS + w) * pstride].
template <int SWL, typename ND>
__device__ __forceinline__ u32 dp_coop_window(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, u32* __restrict__ park, u32 pstride) {
    constexpr int CPL = SWL / 16;
    constexpr int NDW = CPL / 2;
    static_assert(SWL == 64 || SWL == 32, "16 lanes x 4 or 2 columns");
    const u32 gl = threadIdx.x & 15u;
    const u32 Mv = splat16(nd.match_plus_mismatch), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const u32 nchunks = (m + SWL - 1) / SWL;
    auto load_cols = [&](u32 base) -> u32 {  // this lane's CPL bytes of the chunk at `base`, zero behind the window's end
        const u32 p = base + CPL * gl;
        if (p >= m) return 0u;
        u32 w = load_u32_unaligned(th, p);
        const u32 rem = m - p;
        if (rem < 4) w &= (1u << (8 * rem)) - 1;
        if (CPL == 2) w &= 0xFFFFu;
        return w;
    };
    u32 maxs = 0;
    u32 prevflags = 0;  // this lane's last column of the previous chunk: bit 0 lowercase, bit 1 delimiter (lane 15's copy is what the next chunk's lane 0 reads)
    u32 nxt = load_cols(0);
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 w = nxt;
        nxt = ch + 1 < nchunks ? load_cols((ch + 1) * SWL) : 0u;  // one chunk ahead: a window of ten chunks is otherwise a chain of ten exposed loads
        u32 lowerb = 0, upperb = 0, delimb_ = 0;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const u32 b = (w >> (8 * c)) & 0xFFu;
            const bool lower = b >= 'a' && b <= 'z', upper = b >= 'A' && b <= 'Z', digit = b >= '0' && b <= '9';
            const bool delim = !(lower || upper || digit || b > 127);  // NUL padding is a delimiter (ascii.rs:86-89)
            lowerb |= (u32)lower << c;
            upperb |= (u32)upper << c;
            delimb_ |= (u32)delim << c;
        }
        const u32 myflags = ((lowerb >> (CPL - 1)) & 1u) | (((delimb_ >> (CPL - 1)) & 1u) << 1);
        const u32 nbflags = coop_shr<1>(coop_ror<1>(prevflags), myflags);  // column -1: the left neighbour's last column / the previous chunk's last lane (chunk 0: false)
        prevflags = myflags;
        u32 hw[NDW], bonus[NDW];
#pragma unroll
        for (int d = 0; d < NDW; d++) {
            u32 cap01 = 0, dl01 = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int c = 2 * d + h;
                const u32 pl = c ? (lowerb >> (c - 1)) & 1u : nbflags & 1u;
                const u32 pd = c ? (delimb_ >> (c - 1)) & 1u : (nbflags >> 1) & 1u;
                cap01 |= (((upperb >> c) & 1u) & pl) << (16 * h);
                dl01 |= (pd & ~(delimb_ >> c) & 1u) << (16 * h);
            }
            hw[d] = ((w >> (16 * d)) & 0xFFu) | (((w >> (16 * d + 8)) & 0xFFu) << 16);
            bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
        }
        if (ch == 0 && gl == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);  // lane 0 of chunk 0 only (ascii.rs:50-54, 153)
        const bool park_out = ch + 1 < nchunks;
        const u32 mx = ch == 0 ? coop_chunk_rows<SWL, true>(nd, hw, bonus, park_out, park, pstride) : coop_chunk_rows<SWL, false>(nd, hw, bonus, park_out, park, pstride);
        maxs = p_max(maxs, mx);
    }
    maxs = max(maxs & 0xFFFFu, maxs >> 16);
    maxs = max(maxs, coop_ror<8>(maxs));
    maxs = max(maxs, coop_ror<4>(maxs));
    maxs = max(maxs, coop_ror<2>(maxs));
    maxs = max(maxs, coop_ror<1>(maxs));
    return maxs;
}
