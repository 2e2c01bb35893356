// Sub-wave cooperative Smith-Waterman (round 6): SIXTEEN lanes - one DPP row - score one window, four windows per wavefront.
//
// Why: the thread-per-window scorers (dp_cf.h / dp_cfm.h) are built for throughput - a thread walks its window's chunks and needle rows alone,
// ~ 1 000 wave-instructions per (row, chunk), which is the right trade when there are hundreds of thousands of windows.  A SMALL queue (the
// 27 k two- and three-chunk windows of the paths-shaped list: 426 wavefronts on 1 024 SIMDs) leaves most of the chip idle while every wave
// walks a dependent chain of ~ 12 000 instructions: 44.6 us for that slice, one wave's instruction latency.  Here lane l of a row holds the
// CPL = SWL / 16 columns CPL * l .. CPL * l + CPL - 1 of the emulated backend's SWL-lane vector, so a (row, chunk) is ~ 360 instructions and
// four times as many wavefronts share the work; the price is ~ 5 x the instructions in total (nothing is packed, nothing is in closed form,
// every shift is two v_mov_dpp), which is why the DEVICE picks this form only below a queue length (kernels_dp.hip, k2_classes_all: about one
// and a half rounds of the resident groups - lists of ordinary size, 100 k - 300 k paths: -17 % per step; at 27 k queued windows it loses).
//
// The arithmetic is the reference's, literally (the same statements as the wave-per-haystack kernel k2c_generic, which is one column per lane):
// score_haystack src/smith_waterman/algo/ascii.rs:10-158, propagate_horizontal_gaps ascii_gap.rs:11-105.  `shift_right_padded::<K>` is
//   * K a multiple of CPL: a DPP row shift by K / CPL lanes (row_shr), the lanes it leaves empty filled from the previous chunk's final row by a
//     row rotate (row_ror) of the lane's parked copy - one v_mov_dpp each, no LDS round trip, no wave-wide shuffle;
//   * K < CPL: a move between the lane's own columns, the first K columns from the left neighbour's last K (one row_shr:1 each).
// What a (row, chunk) leaves for the next chunk - the row's final values and its match bits - is PRIVATE to the lane (lane l of the next chunk
// needs exactly columns CPL * l ..), so it is parked per thread (LDS, [row][word][thread]) and needs no barrier.
#pragma once
#include "kernels_common.h"

template <int K>
__device__ __forceinline__ u32 coop_shr(u32 old, u32 src) {  // lane i <- src of lane i - K inside its row of 16; lanes i < K keep `old`
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x110 + K, 0xF, 0xF, false);
}
template <int K>
__device__ __forceinline__ u32 coop_ror(u32 src) {  // lane i <- src of lane (i - K) mod 16
    return (u32)__builtin_amdgcn_update_dpp(0, (int)src, 0x120 + K, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 coop_subs(u32 a, u32 b) { return a > b ? a - b : 0; }

// words a thread parks per needle row: CPL u16 values (one or two words) + one word of match bits
template <int SWL>
struct CoopPark {
    static constexpr int CPL = SWL / 16;
    static constexpr int WORDS = CPL / 2 + 1;
};

// One gap step of the log-step scan, K = the shift in columns.  row / adjv: this lane's CPL columns of the row being propagated / of the previous
// chunk's final row; mmb / amm: the row's match bits of this chunk / of the previous one (bit c = column c of the lane).
template <int SWL, int K>
__device__ __forceinline__ void coop_gap_step(u32 (&row)[SWL / 16], const u32 (&adjv)[SWL / 16], u32 mmb, u32 amm, u32 kg, u32 gopm, u32 LM) {
    constexpr int CPL = SWL / 16;
    u32 srow[CPL], smm;
    if constexpr (K < CPL) {
        // the left neighbour's columns (lane 0: the previous chunk's last lane, by rotation of the parked copy)
        u32 nb[CPL];
#pragma unroll
        for (int c = CPL - K; c < CPL; c++) nb[c] = coop_shr<1>(coop_ror<1>(adjv[c]), row[c]);
        const u32 nbm = coop_shr<1>(coop_ror<1>(amm), mmb);
#pragma unroll
        for (int c = 0; c < CPL; c++) srow[c] = c >= K ? row[c - K] : nb[CPL - K + c];
        smm = ((mmb << K) | (nbm >> (CPL - K))) & ((1u << CPL) - 1u);
    } else {
        constexpr int KL = K / CPL;  // lanes
#pragma unroll
        for (int c = 0; c < CPL; c++) srow[c] = coop_shr<KL>(coop_ror<KL>(adjv[c]), row[c]);
        smm = coop_shr<KL>(coop_ror<KL>(amm), mmb);
    }
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const u32 pen = (kg + (((smm >> c) & 1u) ? gopm : 0u)) & LM;
        row[c] = max(row[c], coop_subs(srow[c], pen));
    }
}

// The window th[0 .. m) (1 <= m <= 1024, any number of chunks) scored by the 16 lanes of the caller's DPP row; every lane returns the score.
// park: this thread's slot of the per-row parking area, word w of row r at park[(r * WORDS + w) * pstride].
template <int SWL, typename ND>
__device__ __forceinline__ u32 dp_coop_window(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, u32* __restrict__ park, u32 pstride) {
    constexpr int CPL = SWL / 16;
    constexpr int WORDS = CoopPark<SWL>::WORDS;
    static_assert(SWL == 64 || SWL == 32, "16 lanes x 4 or 2 columns");
    const u32 gl = threadIdx.x & 15u;
    const u32 LM = (u32)nd.lane_mask;
    const u32 rows = (u32)nd.rows;
    const u32 Mc = nd.match_plus_mismatch & LM, X = nd.mismatch & LM, gex = nd.gex & LM, gopm = nd.gopm & LM;
    const u32 caseb = nd.matching_case & LM, capb = nd.capitalization & LM, delimb = nd.delimiter & LM, prefixb = nd.prefix & LM;
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
        u32 b[CPL], bonus[CPL];
        u32 lowerb = 0, upperb = 0, delimb_ = 0;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            b[c] = (w >> (8 * c)) & 0xFFu;
            const bool lower = b[c] >= 'a' && b[c] <= 'z', upper = b[c] >= 'A' && b[c] <= 'Z', digit = b[c] >= '0' && b[c] <= '9';
            const bool delim = !(lower || upper || digit || b[c] > 127);  // NUL padding is a delimiter (ascii.rs:86-89)
            lowerb |= (u32)lower << c;
            upperb |= (u32)upper << c;
            delimb_ |= (u32)delim << c;
        }
        const u32 myflags = ((lowerb >> (CPL - 1)) & 1u) | (((delimb_ >> (CPL - 1)) & 1u) << 1);
        const u32 nbflags = coop_shr<1>(coop_ror<1>(prevflags), myflags);  // column -1: the left neighbour's last column / the previous chunk's last lane (chunk 0: false)
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const u32 pl = c ? (lowerb >> (c - 1)) & 1u : nbflags & 1u;
            const u32 pd = c ? (delimb_ >> (c - 1)) & 1u : (nbflags >> 1) & 1u;
            const u32 cap = (((upperb >> c) & 1u) && pl) ? capb : 0u, dl = (pd && !((delimb_ >> c) & 1u)) ? delimb : 0u;
            u32 bn = (dl + cap) & LM;
            bn = (bn + ((ch == 0 && gl == 0 && c == 0 && include_prefix) ? prefixb : 0u)) & LM;
            bonus[c] = (bn + Mc) & LM;
        }
        prevflags = myflags;
        u32 prev_row[CPL], row[CPL];
#pragma unroll
        for (int c = 0; c < CPL; c++) prev_row[c] = 0, row[c] = 0;
        u32 up_mm = 0;      // match bits of the row above (ascii.rs:130-133: the gap-open surcharge of `up` is keyed on them)
        u32 carry_src = 0;  // S(r - 1, previous chunk), this lane's last column: lane 15's copy is the diagonal source of lane 0, column 0
        for (u32 r = 1; r <= rows; r++) {
            const u32 cr = nd.c[r - 1], fr = nd.f[r - 1];
            u32 mmb = 0, exb = 0;
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const bool ex = b[c] == cr;
                exb |= (u32)ex << c;
                mmb |= (u32)(ex || b[c] == fr) << c;
            }
            // what this lane parked for row r in the previous chunk
            u32 adjv[CPL], amm = 0;
#pragma unroll
            for (int c = 0; c < CPL; c++) adjv[c] = 0;
            if (ch) {
                const u32 v0 = park[(r * WORDS + 0) * pstride];
                adjv[0] = v0 & 0xFFFFu;
                adjv[1] = v0 >> 16;
                if (CPL == 4) {
                    const u32 v1 = park[(r * WORDS + 1) * pstride];
                    adjv[CPL - 2] = v1 & 0xFFFFu;
                    adjv[CPL - 1] = v1 >> 16;
                }
                amm = park[(r * WORDS + WORDS - 1) * pstride];
            }
            // diagonal (ascii.rs:118-127) and up (:130-133)
            const u32 nbp = coop_shr<1>(coop_ror<1>(carry_src), prev_row[CPL - 1]);
#pragma unroll
            for (int c = 0; c < CPL; c++) {
                const u32 dsrc = c ? prev_row[c - 1] : nbp;
                u32 diag = (dsrc + (((mmb >> c) & 1u) ? bonus[c] : 0u)) & LM;
                diag = coop_subs(diag, X);
                diag = (diag + (((exb >> c) & 1u) ? caseb : 0u)) & LM;
                const u32 up = coop_subs(coop_subs(prev_row[c], gex), ((up_mm >> c) & 1u) ? gopm : 0u);
                row[c] = max(diag, up);
            }
            // propagate_horizontal_gaps (ascii_gap.rs:11-105): shifts 1, 2, 4, ... SWL / 2, each over the row as the previous step left it
            u32 kg = gex;
            coop_gap_step<SWL, 1>(row, adjv, mmb, amm, kg, gopm, LM); kg = (kg + kg) & LM;
            coop_gap_step<SWL, 2>(row, adjv, mmb, amm, kg, gopm, LM); kg = (kg + kg) & LM;
            coop_gap_step<SWL, 4>(row, adjv, mmb, amm, kg, gopm, LM); kg = (kg + kg) & LM;
            coop_gap_step<SWL, 8>(row, adjv, mmb, amm, kg, gopm, LM); kg = (kg + kg) & LM;
            coop_gap_step<SWL, 16>(row, adjv, mmb, amm, kg, gopm, LM); kg = (kg + kg) & LM;
            if constexpr (SWL == 64) coop_gap_step<SWL, 32>(row, adjv, mmb, amm, kg, gopm, LM);
            carry_src = adjv[CPL - 1];
            if (ch + 1 < nchunks) {
                park[(r * WORDS + 0) * pstride] = row[0] | (row[1] << 16);
                if (CPL == 4) park[(r * WORDS + 1) * pstride] = row[CPL - 2] | (row[CPL - 1] << 16);
                park[(r * WORDS + WORDS - 1) * pstride] = mmb;
            }
#pragma unroll
            for (int c = 0; c < CPL; c++) prev_row[c] = row[c];
            up_mm = mmb;
        }
#pragma unroll
        for (int c = 0; c < CPL; c++) maxs = max(maxs, row[c]);  // every lane of the chunk, padding included (ascii.rs:152-156)
    }
    maxs = max(maxs, coop_ror<8>(maxs));
    maxs = max(maxs, coop_ror<4>(maxs));
    maxs = max(maxs, coop_ror<2>(maxs));
    maxs = max(maxs, coop_ror<1>(maxs));
    return maxs;
}
