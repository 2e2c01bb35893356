// gfx950 kernels, stage 1: the streaming filter over the whole haystack list (HBM-bound) and the small
// order-preserving compaction kernels (scan / map) that turn its bitmap into a dense survivor list.
//
// Reference semantics being implemented:
//   * `match_list_into_impl` outer loop: length check + prefilter accept (src/matcher/algo.rs:78-103)
//   * 0-typo ASCII accept  == needle is a case-folded ordered subsequence (src/prefilter/algo/ascii.rs:6-54)
//   * k-typo accept        is implied by LCS(needle, haystack) + k >= rows (src/prefilter/mod.rs:1013-1084);
//     the converse does not hold at every lane width (oracle/selfcheck.cpp), so with typos - and on the unicode
//     path, where this stage only looks at each scalar's LAST byte - this stage is a conservative superset
//     and the lane-exact prefilter (kernels_window.hip) re-decides every survivor.
#include "kernels_common.h"
#include "dfa_lds.h"
#include <algorithm>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------
// K1: one thread per haystack.  Bytes are streamed from HBM as aligned 16-byte vectors (padded-16 layout),
// each byte indexes a 256-entry table in LDS whose entry is the bitmask of needle rows that byte can
// match (either case); the per-thread state is one machine word:
//   MODE 1 (ordered subsequence): st is one-hot at the next row to match; `st += st & T[b]` advances it.
//   MODE 2 (bit-vector LCS, Allison-Dix/Hyyro): V' = (V + (V & M)) | (V & ~M); LCS = #zero bits.
// Output: 1 bit per haystack (wave ballot -> one u64 store per wave) + a count per 1024-haystack tile.
// ---------------------------------------------------------------------------------------------------
template <typename TW, int MODE>
__device__ __forceinline__ void filter_step(TW& st, u32 b, const TW* T) {
    TW t = T[b];
    if (MODE == 1) {
        st += st & t;
    } else {
        TW u = st & t;
        st = (st + u) | (st & ~t);
    }
}

template <typename TW, int MODE>
__device__ __forceinline__ void filter_word(TW& st, u32 w, u32 nbytes, const TW* T) {
    if (nbytes >= 4) {
        filter_step<TW, MODE>(st, w & 0xFF, T);
        filter_step<TW, MODE>(st, (w >> 8) & 0xFF, T);
        filter_step<TW, MODE>(st, (w >> 16) & 0xFF, T);
        filter_step<TW, MODE>(st, w >> 24, T);
    } else {
        for (u32 k = 0; k < nbytes; k++) filter_step<TW, MODE>(st, (w >> (8 * k)) & 0xFF, T);
    }
}

// MARG (MODE 2 only): a second bit per haystack, "accepted with nothing to spare" (LCS == need exactly), with its own per-tile
// counts - the only inputs on which the reference's chunked typo prefilter can differ from the LCS criterion
// (tests/test_oracle_reference_properties.py), so only those are re-decided at the exact lane width (k2a_window, decide form).
// The kernel also clears the per-haystack reject bits / per-tile reject counts that the decide pass sets.
struct MargOut {
    u64* bitmap_m;
    u32* tile_counts_m;
    u64* reject_bits;
    u32* tile_rejects;
};
template <typename TW, int MODE, typename ET, bool MARG = false>
__global__ __launch_bounds__(256) void k1_filter(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                 const u64* __restrict__ Tg, int rows, int need, u32 min_len,
                                                 u64* __restrict__ bitmap, u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, MargOut mo = MargOut{}, u32 ulen = 0) {
    // the per-call counter block is cleared here (first workgroup) instead of by a separate memset launch: nothing before the
    // compaction kernel reads it
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    __shared__ TW T[256];
    __shared__ u32 s_cnt, s_cnt_m;
    const int tid = threadIdx.x;
    T[tid] = (TW)Tg[tid];
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0, s_cnt_m = 0;
        __syncthreads();
        u32 cnt = 0, cnt_m = 0;
#pragma unroll 1
        for (int p = 0; p < FZB_TILE / 256; p++) {
            const u32 li = tile * FZB_TILE + p * 256 + tid;
            bool matched = false, marginal = false;
            if (li < count) {
                u64 s;
                u32 L;
                haystack_span_u(ends, ulen, first + li, s, L);
                if (L >= min_len) {
                    const uint4* vp = (const uint4*)(bytes + s);
                    TW st = (MODE == 1) ? (TW)1 : (TW)~(TW)0;
                    const u32 nvec = (L + 15) >> 4;
                    for (u32 v = 0; v < nvec; v++) {
                        const uint4 q = vp[v];
                        const u32 rem = L - 16 * v;
                        filter_word<TW, MODE>(st, q.x, rem, T);
                        if (rem > 4) filter_word<TW, MODE>(st, q.y, rem - 4, T);
                        if (rem > 8) filter_word<TW, MODE>(st, q.z, rem - 8, T);
                        if (rem > 12) filter_word<TW, MODE>(st, q.w, rem - 12, T);
                    }
                    if (MODE == 1) {
                        matched = (st >> rows) & 1;
                    } else {
                        const TW low = rows >= (int)(8 * sizeof(TW)) ? (TW)~(TW)0 : (((TW)1 << rows) - 1);
                        const TW z = ~st & low;
                        const int lcs = sizeof(TW) == 8 ? __popcll((u64)z) : __popc((u32)z);
                        matched = lcs >= need;
                        marginal = lcs == need;
                    }
                }
            }
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + p * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
            if (MARG) {
                const u64 bm = __ballot(marginal);
                if (lane_id() == 0) {
                    mo.bitmap_m[(tile * FZB_TILE + p * 256) / 64 + (tid >> 6)] = bm;
                    mo.reject_bits[(tile * FZB_TILE + p * 256) / 64 + (tid >> 6)] = 0;
                    cnt_m += __popcll(bm);
                }
            }
        }
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        if (MARG && lane_id() == 0 && cnt_m) atomicAdd(&s_cnt_m, cnt_m);
        __syncthreads();
        if (tid == 0) {
            tile_counts[tile] = s_cnt;
            if (MARG) mo.tile_counts_m[tile] = s_cnt_m, mo.tile_rejects[tile] = 0;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA: the ordered-subsequence test (MODE 1) as a table-driven DFA.  State s = number of needle rows
// already matched; dfa[s * 256 + b] = s + 1 if byte b matches row s (either case), else s; state `rows`
// is absorbing.  One step is ONE v_perm_b32 (builds the LDS address (s << 8) | byte) + ONE ds_read_u8,
// instead of extract / lookup / and / add.  The lookup chain is serial per haystack, so every thread runs
// the 4 haystacks it owns in a tile as 4 interleaved chains, with all their 16-byte vectors requested
// from HBM up front.  The table has (rows + 1) * 256 bytes; 4 byte values share a dword, so the
// alphanumerics of one state row spread over distinct LDS banks.
// ---------------------------------------------------------------------------------------------------
template <typename ET>
__global__ __launch_bounds__(256) void k1_dfa(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                              const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                              u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, u32 ulen) {
    // the per-call counter block is cleared here (first workgroup) instead of by a separate memset launch: nothing before the
    // compaction kernel reads it
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    // the table is the ONLY LDS object of the kernel and therefore sits at LDS address 0: a lookup's address is the v_perm result itself
    // (behind a static __shared__ variable every lookup paid a v_add of the table's offset); the tile counter lives behind the table
    extern __shared__ __attribute__((aligned(16))) u8 dfa[];
    u32& s_cnt = *(u32*)(dfa + FZB_DFA_LDS_BYTES(rows));
    const int tid = threadIdx.x;
    dfa_require_lds_base0(dfa);
    dfa_load_lds(dfa, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u64 hs[4];
        u32 hl[4];
        uint4 v0[4], v1[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const u32 li = tile * FZB_TILE + p * 256 + tid;
            hs[p] = 0;
            hl[p] = 0;
            if (li < count) haystack_span_u(ends, ulen, first + li, hs[p], hl[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            v0[p] = make_uint4(0, 0, 0, 0);
            v1[p] = make_uint4(0, 0, 0, 0);
            const uint4* vp = (const uint4*)(bytes + hs[p]);
            if (hl[p] > 0) v0[p] = vp[0];
            if (hl[p] > 16) v1[p] = vp[1];
        }
        u32 st[4] = {0, 0, 0, 0};
        // short haystacks (<= 32 bytes): both vectors are already in flight
        if (hl[0] >= 16 && hl[1] >= 16 && hl[2] >= 16 && hl[3] >= 16) {
            { const u32 w[4] = {v0[0].x, v0[1].x, v0[2].x, v0[3].x}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v0[0].y, v0[1].y, v0[2].y, v0[3].y}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v0[0].z, v0[1].z, v0[2].z, v0[3].z}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v0[0].w, v0[1].w, v0[2].w, v0[3].w}; dfa_word4(st, w, dfa); }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) st[p] = dfa_partial(st[p], v0[p], hl[p] >= 16 ? 16u : hl[p], dfa);
        }
        if (hl[0] >= 32 && hl[1] >= 32 && hl[2] >= 32 && hl[3] >= 32) {
            { const u32 w[4] = {v1[0].x, v1[1].x, v1[2].x, v1[3].x}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v1[0].y, v1[1].y, v1[2].y, v1[3].y}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v1[0].z, v1[1].z, v1[2].z, v1[3].z}; dfa_word4(st, w, dfa); }
            { const u32 w[4] = {v1[0].w, v1[1].w, v1[2].w, v1[3].w}; dfa_word4(st, w, dfa); }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (hl[p] > 16) st[p] = dfa_partial(st[p], v1[p], hl[p] >= 32 ? 16u : hl[p] - 16, dfa);
        }
        u32 cnt = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const u32 L = hl[p];
            const u32 li = tile * FZB_TILE + p * 256 + tid;
            const bool matched = li < count && L >= min_len && st[p] >= acc_lo;
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + p * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists (haystacks longer than the two pre-requested vectors).  Same DFA; P = haystacks a
// thread runs interleaved (a 1024-haystack tile takes 4 / P sub-passes).  With long haystacks the quantity to
// control is the cache footprint, not memory-level parallelism: a wave's load touches 64 haystacks = a
// contiguous ~5 KB of which only 16 B per haystack are consumed, and the rest of those lines must still be in
// L2 when the lane comes back for its next vector.  Footprint per CU = resident waves x P x ~5 KB, so P and the
// number of resident workgroups are launch parameters (fzb_launch_filter).
// ---------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void dfa_wordP(u32 (&st)[P], const u32 (&w)[P], const u8* dfa) {
#pragma unroll
    for (int p = 0; p < P; p++) st[p] = dfa_step<0>(st[p], w[p], dfa);
    if (P > 1) FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < P; p++) st[p] = dfa_step<1>(st[p], w[p], dfa);
    if (P > 1) FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < P; p++) st[p] = dfa_step<2>(st[p], w[p], dfa);
    if (P > 1) FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < P; p++) st[p] = dfa_step<3>(st[p], w[p], dfa);
    if (P > 1) FZB_WAIT_LDS();
}

// SAN = false (needle without a NUL byte): the zero fill between a haystack's end and its 16-byte boundary - and the zero vectors a lane
// sees after its haystack ended - match no needle row, so the vectors go through the DFA unmasked (a third of the loop's instructions).
template <typename ET, int P, bool SAN = true>
__global__ __launch_bounds__(256) void k1_dfa_ragged(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                     const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                     u32* __restrict__ tile_counts, u32* __restrict__ reset_counters) {
    // the per-call counter block is cleared here (first workgroup) instead of by a separate memset launch: nothing before the
    // compaction kernel reads it
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    // the table is the ONLY LDS object of the kernel and therefore sits at LDS address 0: a lookup's address is the v_perm result itself
    // (behind a static __shared__ variable every lookup paid a v_add of the table's offset); the tile counter lives behind the table
    extern __shared__ __attribute__((aligned(16))) u8 dfa[];
    u32& s_cnt = *(u32*)(dfa + FZB_DFA_LDS_BYTES(rows));
    const int tid = threadIdx.x;
    dfa_require_lds_base0(dfa);
    dfa_load_lds(dfa, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int sub = 0; sub < 4 / P; sub++) {
            const u32 base = tile * FZB_TILE + sub * (256 * P);
            u64 hs[P];
            u32 hl[P];
            uint4 cur[P], nxt[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                const u32 li = base + p * 256 + tid;
                hs[p] = 0;
                hl[p] = 0;
                if (li < count) haystack_span(ends, first + li, hs[p], hl[p]);
            }
#pragma unroll
            for (int p = 0; p < P; p++) {
                cur[p] = make_uint4(0, 0, 0, 0);
                nxt[p] = make_uint4(0, 0, 0, 0);
                const uint4* vp = (const uint4*)(bytes + hs[p]);
                if (hl[p] > 0) cur[p] = vp[0];
                if (hl[p] > 16) nxt[p] = vp[1];
            }
            u32 st[P];
            u32 nvmax = 0;
#pragma unroll
            for (int p = 0; p < P; p++) {
                st[p] = 0;
                nvmax = max(nvmax, (hl[p] + 15) >> 4);
            }
            // Bytes past a haystack's end are replaced by `dead`, a byte value no needle row can match, so every vector runs
            // the same branch-free unrolled steps.
            for (u32 v = 0; v < nvmax; v++) {
                uint4 nn[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    nn[p] = make_uint4(0, 0, 0, 0);
                    if (hl[p] > 16 * (v + 2)) nn[p] = ((const uint4*)(bytes + hs[p]))[v + 2];
                }
                u32 wx[P], wy[P], wz[P], ww[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const u32 rem = hl[p] > 16 * v ? hl[p] - 16 * v : 0u;  // valid bytes from this vector on
                    auto san = [&](u32 w, u32 off) {
                        const u32 nv = rem > off ? rem - off : 0u;
                        const u32 mask = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1);
                        return (w & mask) | (deadv & ~mask);
                    };
                    wx[p] = SAN ? san(cur[p].x, 0) : cur[p].x;
                    wy[p] = SAN ? san(cur[p].y, 4) : cur[p].y;
                    wz[p] = SAN ? san(cur[p].z, 8) : cur[p].z;
                    ww[p] = SAN ? san(cur[p].w, 12) : cur[p].w;
                }
                dfa_wordP<P>(st, wx, dfa);
                dfa_wordP<P>(st, wy, dfa);
                dfa_wordP<P>(st, wz, dfa);
                dfa_wordP<P>(st, ww, dfa);
#pragma unroll
                for (int p = 0; p < P; p++) cur[p] = nxt[p], nxt[p] = nn[p];
            }
#pragma unroll
            for (int p = 0; p < P; p++) {
                const u32 li = base + p * 256 + tid;
                const bool matched = li < count && hl[p] >= min_len && st[p] >= acc_lo;
                const u64 b = __ballot(matched);
                if (lane_id() == 0) {
                    bitmap[(base + p * 256) / 64 + (tid >> 6)] = b;
                    cnt += __popcll(b);
                }
            }
        }
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists, burst form: a thread requests ALL vectors of its haystack (up to 8 = 128 bytes per round) back to
// back and only then runs the DFA over them from registers.  In the rolling form above a 128-byte line is touched by ~8 loads
// of a wave that are separated by the DFA work of every resident wave, and the CU's footprint (24 waves x 64 haystacks x up to
// 128 B) is far beyond the vector L1, so most of those touches re-fetch the line from L2; issued back to back they hit.
// C4 shard: 297 -> 242 us.  (Two or four haystacks per thread as interleaved DFA chains on top of it: 276 / 300 us, and again 281 / 405 us after
// the lookups were cut to v_perm + ds_read with one wait per round - with 8 waves per SIMD the lookup chain is already hidden; neither that
// nor removing a quarter of the loop's instructions moved the kernel, so it is bound by its cache transactions: every 16-byte piece of a
// lane is its own request, ~35 lines per wave instruction.)
// ---------------------------------------------------------------------------------------------------
// PERM (round 3): `bytes` / `ends` are the corpus' length-sorted filter view (CorpusDev::fbytes / fends; `first` a multiple of the tile
// size): the haystack at sorted position g of a tile came from position perm[g] of that tile, where its decision bit belongs - collected
// in LDS (atomic-or) and written per tile.  A wave's 64 lanes then hold haystacks of one vector-count class: the loop below runs no
// lookups for lanes whose haystack has ended.
template <typename ET, bool SAN, bool PERM = false>
__global__ __launch_bounds__(256) void k1_dfa_ragged_burst(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                           const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                           u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, const u16* __restrict__ perm = nullptr) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    // the table is the ONLY LDS object of the kernel and therefore sits at LDS address 0: a lookup's address is the v_perm result itself
    // (behind a static __shared__ variable every lookup paid a v_add of the table's offset); the tile counter (and, PERM, the tile's 32
    // words of decision bits) live behind the table
    extern __shared__ __attribute__((aligned(16))) u8 dfa[];
    u32& s_cnt = *(u32*)(dfa + FZB_DFA_LDS_BYTES(rows));
    u32* const s_bits = &s_cnt + 4;
    const int tid = threadIdx.x;
    dfa_require_lds_base0(dfa);
    dfa_load_lds(dfa, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        if (PERM && tid < 32) s_bits[tid] = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int sub = 0; sub < 4; sub++) {
            const u32 li = tile * FZB_TILE + sub * 256 + tid;
            u64 hs = 0;
            u32 hl = 0;
            u32 orig = 0;
            if (li < count) {
                haystack_span(ends, first + li, hs, hl);
                if (PERM) orig = perm[first + li];
            }
            const uint4* vp = (const uint4*)(bytes + hs);
            u32 st = 0;
            for (u32 v0 = 0; 16 * v0 < hl; v0 += 8) {  // rounds of 8 vectors (one round for haystacks up to 128 bytes)
                uint4 q[8];
#pragma unroll
                for (int k = 0; k < 8; k++) q[k] = hl > 16 * (v0 + k) ? vp[v0 + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (hl <= 16 * (v0 + k)) break;
                    u32 w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                    if (SAN) {
                        const u32 rem = hl - 16 * (v0 + k);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                            const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                            w[j] = (w[j] & mask) | (deadv & ~mask);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        st = dfa_step<0>(st, w[j], dfa);
                        st = dfa_step<1>(st, w[j], dfa);
                        st = dfa_step<2>(st, w[j], dfa);
                        st = dfa_step<3>(st, w[j], dfa);
                    }
                }
            }
            const bool matched = li < count && hl >= min_len && st >= acc_lo;
            if (PERM) {
                if (matched) atomicOr(&s_bits[orig >> 5], 1u << (orig & 31));
                continue;
            }
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + sub * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (PERM) {
            __syncthreads();
            if (tid < FZB_TILE / 64) {
                const u64 word = (u64)s_bits[2 * tid] | ((u64)s_bits[2 * tid + 1] << 32);
                bitmap[(size_t)tile * (FZB_TILE / 64) + tid] = word;
                cnt = (u32)__popcll(word);
            }
            if (tid < FZB_TILE / 64 && cnt) atomicAdd(&s_cnt, cnt);
        } else if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists over the CLASS-COMPOSITE automaton (round 3).  The four structural variants below left the burst form's time
// where it was, which pointed at the one thing they share: every wave is a serial chain of one dependent LDS lookup PER BYTE (~100 cycles
// each under load), and with all eight wave slots of a SIMD taken the throughput is waves / chain latency.  Here the chain has one link
// per G bytes: the automaton only distinguishes K byte CLASSES (bytes with identical columns: the needle's letters in either case, and
// "anything else" - 6 for `deadbeef`), so the host composes G transitions into one table, next = comp[state][c0 + K c1 + K^2 c2 + K^3 c3]
// (states x K^G bytes; G = 4 when that fits 16 KB, else 2).  Per dword: four class lookups cls[byte] that do NOT depend on the state (issued
// back to back for the whole 16-byte vector, off the chain), three multiply-adds for the offset, and ONE dependent lookup.
// LDS: [0, 256) byte -> class, [256, 256 + states * K^G) the composite table; lookups address LDS directly (the object starts at 0).
// ---------------------------------------------------------------------------------------------------
template <typename ET, bool SAN, int G>
__global__ __launch_bounds__(256) void k1_cdfa_ragged(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                      const u8* __restrict__ cdfa_g, u32 cdfa_bytes, u32 K, u32 KG, u32 min_len, u32 dead, u32 acc_lo,
                                                      u64* __restrict__ bitmap, u32* __restrict__ tile_counts, u32* __restrict__ reset_counters) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tab_bytes = (cdfa_bytes + 15u) & ~15u;
    u32& s_cnt = *(u32*)(lds + tab_bytes);
    const int tid = threadIdx.x;
    dfa_require_lds_base0(lds);
    for (u32 i = tid * 4; i < tab_bytes; i += 256 * 4) *(u32*)(lds + i) = i < cdfa_bytes ? *(const u32*)(cdfa_g + i) : 0u;  // (the blob is padded to 16 bytes)
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    auto cls_of = [](u32 b) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)b; };
    auto comp_at = [](u32 a) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)(256u + a); };
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int sub = 0; sub < 4; sub++) {
            const u32 li = tile * FZB_TILE + sub * 256 + tid;
            u64 hs = 0;
            u32 hl = 0;
            if (li < count) haystack_span(ends, first + li, hs, hl);
            const uint4* vp = (const uint4*)(bytes + hs);
            u32 st = 0;
            for (u32 v0 = 0; 16 * v0 < hl; v0 += 8) {  // rounds of 8 vectors (one round for haystacks up to 128 bytes)
                uint4 q[8];
#pragma unroll
                for (int k = 0; k < 8; k++) q[k] = hl > 16 * (v0 + k) ? vp[v0 + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (hl <= 16 * (v0 + k)) break;
                    u32 w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                    if (SAN) {
                        const u32 rem = hl - 16 * (v0 + k);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                            const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                            w[j] = (w[j] & mask) | (deadv & ~mask);
                        }
                    }
                    if (K == 0xFFFFu) {  // measurement knob (FZB_CDFA_NODFA=1, results meaningless): the loads alone
                        st ^= w[0] ^ w[1] ^ w[2] ^ w[3];
                        continue;
                    }
                    // the vector's sixteen class lookups: independent of the state
                    u32 c[4][4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        c[j][0] = cls_of(w[j] & 0xFF);
                        c[j][1] = cls_of((w[j] >> 8) & 0xFF);
                        c[j][2] = cls_of((w[j] >> 16) & 0xFF);
                        c[j][3] = cls_of(w[j] >> 24);
                    }
                    if (G == 4) {
                        u32 off[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) off[j] = c[j][0] + K * (c[j][1] + K * (c[j][2] + K * c[j][3]));
#pragma unroll
                        for (int j = 0; j < 4; j++) st = comp_at(st * KG + off[j]);  // the chain: one dependent lookup per dword
                    } else {
                        u32 off[8];
#pragma unroll
                        for (int j = 0; j < 4; j++) off[2 * j] = c[j][0] + K * c[j][1], off[2 * j + 1] = c[j][2] + K * c[j][3];
#pragma unroll
                        for (int j = 0; j < 8; j++) st = comp_at(st * KG + off[j]);  // one dependent lookup per byte pair
                    }
                }
            }
            const bool matched = li < count && hl >= min_len && st >= acc_lo;
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + sub * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The class-composite automaton over the corpus' filter VIEW (CorpusDev::vbytes: tiles sorted by vector count, groups of 64 interleaved by
// vector).  One wave per group: "vector v of my haystack" is ONE coalesced 1 KiB load for the whole wave, every lane of the wave has the
// same number of vectors, and all of a group's loads are issued back to back.  The decision bit goes to the haystack's ORIGINAL position
// in its tile (vperm) through an LDS atomic-or, so the bitmap, the per-tile counts and every later stage see nothing of the view.
// NV = vectors held in registers per haystack (8: lists up to 128 bytes, 16: up to 256).
// ---------------------------------------------------------------------------------------------------
template <bool SAN, int G, int NV>
__global__ __launch_bounds__(256) void k1_cdfa_view(const u8* __restrict__ vbytes, const u32* __restrict__ vgofs, const u8* __restrict__ vgnv, const u16* __restrict__ vlen,
                                                    const u16* __restrict__ vperm, u64 first, u32 count, const u8* __restrict__ cdfa_g, u32 cdfa_bytes, u32 K, u32 KG,
                                                    u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap, u32* __restrict__ tile_counts, u32* __restrict__ reset_counters) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tab_bytes = (cdfa_bytes + 15u) & ~15u;
    u32& s_cnt = *(u32*)(lds + tab_bytes);
    u32* const s_bits = (u32*)(lds + tab_bytes + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dfa_require_lds_base0(lds);
    for (u32 i = tid * 4; i < tab_bytes; i += 256 * 4) *(u32*)(lds + i) = i < cdfa_bytes ? *(const u32*)(cdfa_g + i) : 0u;
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    auto cls_of = [](u32 b) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)b; };
    auto comp_at = [](u32 a) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)(256u + a); };
    const u64 g_first = first / 64;  // `first` is a multiple of the tile size
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        if (tid < 32) s_bits[tid] = 0;
        __syncthreads();
#pragma unroll 1
        for (int gi = 0; gi < 4; gi++) {
            const u32 p = tile * FZB_TILE + (u32)(gi * 4 + wave) * 64 + lane;  // sorted position (relative to `first`)
            const u32 gl = tile * (FZB_TILE / 64) + (u32)(gi * 4 + wave);       // its group
            if (gl * 64 >= count) continue;
            const u32 nv = __builtin_amdgcn_readfirstlane((u32)vgnv[g_first + gl]);
            const u8* base = vbytes + (size_t)__builtin_amdgcn_readfirstlane(vgofs[g_first + gl]) * 16 + (u32)lane * 16;
            u32 hl = 0, orig = 0;
            if (p < count) { hl = vlen[first + p]; orig = vperm[first + p]; }
            uint4 q[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = (u32)k < nv ? *(const uint4*)(base + (size_t)k * 1024) : make_uint4(0, 0, 0, 0);
            u32 st = 0;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if ((u32)k >= nv) continue;  // wave-uniform
                u32 w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                if (K == 0xFFFFu) {  // measurement knob (FZB_CDFA_NODFA=1, results meaningless): the loads alone
                    st ^= w[0] ^ w[1] ^ w[2] ^ w[3];
                    continue;
                }
                if (SAN) {
                    const u32 rem = hl > 16u * k ? hl - 16u * k : 0u;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                        const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                        w[j] = (w[j] & mask) | (deadv & ~mask);
                    }
                }
                u32 c[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    c[j][0] = cls_of(w[j] & 0xFF);
                    c[j][1] = cls_of((w[j] >> 8) & 0xFF);
                    c[j][2] = cls_of((w[j] >> 16) & 0xFF);
                    c[j][3] = cls_of(w[j] >> 24);
                }
                if (G == 4) {
                    u32 off[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) off[j] = c[j][0] + K * (c[j][1] + K * (c[j][2] + K * c[j][3]));
#pragma unroll
                    for (int j = 0; j < 4; j++) st = comp_at(st * KG + off[j]);
                } else {
                    u32 off[8];
#pragma unroll
                    for (int j = 0; j < 4; j++) off[2 * j] = c[j][0] + K * c[j][1], off[2 * j + 1] = c[j][2] + K * c[j][3];
#pragma unroll
                    for (int j = 0; j < 8; j++) st = comp_at(st * KG + off[j]);
                }
            }
            if (p < count && hl >= min_len && st >= acc_lo) atomicOr(&s_bits[orig >> 5], 1u << (orig & 31));
        }
        __syncthreads();
        if (tid < FZB_TILE / 64) {
            const u64 word = (u64)s_bits[2 * tid] | ((u64)s_bits[2 * tid + 1] << 32);
            bitmap[(size_t)tile * (FZB_TILE / 64) + tid] = word;
            const u32 cnt = (u32)__popcll(word);
            if (cnt) atomicAdd(&s_cnt, cnt);
        }
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists, software-pipelined burst form (round 3).  Measured: neither removing the dead lookups (length-sorted view),
// nor making every load fully coalesced (cooperative form below), nor staging in LDS moves the burst form's ~235 us - every wave runs
// span loads -> vector loads -> a serial chain of DFA lookups one after the other, all eight wave slots of a SIMD are taken, and the
// throughput is (waves in flight) x (work per wave) / (that serial latency).  Here a thread has THREE haystacks in flight: the DFA runs
// over the vectors of haystack i (registers) while the vectors of haystack i+1 and the end offsets of haystack i+2 are on their way.
// The price is registers (two sets of eight vectors): fewer waves per SIMD, each of them never waiting for memory.
// ---------------------------------------------------------------------------------------------------
template <typename ET, bool SAN, bool PERM>
__global__ __launch_bounds__(256) void k1_dfa_ragged_pipe(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                          const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                          u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, const u16* __restrict__ perm) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) u8 dfa[];
    u32& s_cnt = *(u32*)(dfa + FZB_DFA_LDS_BYTES(rows));
    u32* const s_bits = &s_cnt + 4;
    const int tid = threadIdx.x;
    dfa_require_lds_base0(dfa);
    dfa_load_lds(dfa, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    // item = (tile, sub-pass): this workgroup's items in order; li = the thread's haystack of the item (>= count: none)
    auto item_li = [&](u32 it) -> u32 {
        const u32 tile = blockIdx.x + (it >> 2) * gridDim.x;
        return tile < ntiles ? tile * FZB_TILE + (it & 3) * 256 + tid : 0xFFFFFFFFu;
    };
    auto load_span = [&](u32 li, u64& hs, u32& hl, u32& orig) {
        hs = 0; hl = 0; orig = 0;
        if (li < count) {
            haystack_span(ends, first + li, hs, hl);
            if (PERM) orig = perm[first + li];
        }
    };
    auto load_vecs = [&](u64 hs, u32 hl, uint4 (&q)[8]) {
        const uint4* vp = (const uint4*)(bytes + hs);
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = hl > 16u * k ? vp[k] : make_uint4(0, 0, 0, 0);
    };
    u64 hs_c, hs_n, hs_f;
    u32 hl_c, hl_n, hl_f, or_c, or_n, or_f;
    uint4 qc[8], qn[8];
    load_span(item_li(0), hs_c, hl_c, or_c);
    load_span(item_li(1), hs_n, hl_n, or_n);
    load_vecs(hs_c, hl_c, qc);
    u32 cnt = 0;
    for (u32 it = 0;; it++) {
        const u32 tile = blockIdx.x + (it >> 2) * gridDim.x;
        if (tile >= ntiles) break;
        const int sub = (int)(it & 3);
        if (sub == 0) {
            if (tid == 0) s_cnt = 0;
            if (PERM && tid < 32) s_bits[tid] = 0;
            cnt = 0;
            __syncthreads();
        }
        load_vecs(hs_n, hl_n, qn);                       // item it + 1: its vectors
        load_span(item_li(it + 2), hs_f, hl_f, or_f);     // item it + 2: its span
        const u32 li = tile * FZB_TILE + sub * 256 + tid;
        u32 st = 0;
        {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (hl_c <= 16u * k) continue;
                u32 w[4] = {qc[k].x, qc[k].y, qc[k].z, qc[k].w};
                if (SAN) {
                    const u32 rem = hl_c - 16u * k;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                        const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                        w[j] = (w[j] & mask) | (deadv & ~mask);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    st = dfa_step<0>(st, w[j], dfa);
                    st = dfa_step<1>(st, w[j], dfa);
                    st = dfa_step<2>(st, w[j], dfa);
                    st = dfa_step<3>(st, w[j], dfa);
                }
            }
            // haystacks beyond 128 bytes: the rest in rounds of eight vectors, loaded here (not a case of the streaming lists)
            const uint4* vp = (const uint4*)(bytes + hs_c);
            for (u32 v0 = 8; 16 * v0 < hl_c; v0 += 8) {
                uint4 q[8];
#pragma unroll
                for (int k = 0; k < 8; k++) q[k] = hl_c > 16 * (v0 + k) ? vp[v0 + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (hl_c <= 16 * (v0 + k)) continue;
                    u32 w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                    if (SAN) {
                        const u32 rem = hl_c - 16 * (v0 + k);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                            const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                            w[j] = (w[j] & mask) | (deadv & ~mask);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        st = dfa_step<0>(st, w[j], dfa);
                        st = dfa_step<1>(st, w[j], dfa);
                        st = dfa_step<2>(st, w[j], dfa);
                        st = dfa_step<3>(st, w[j], dfa);
                    }
                }
            }
        }
        const bool matched = li < count && hl_c >= min_len && st >= acc_lo;
        if (PERM) {
            if (matched) atomicOr(&s_bits[or_c >> 5], 1u << (or_c & 31));
        } else {
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + sub * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (sub == 3) {
            if (PERM) {
                __syncthreads();
                cnt = 0;
                if (tid < FZB_TILE / 64) {
                    const u64 word = (u64)s_bits[2 * tid] | ((u64)s_bits[2 * tid + 1] << 32);
                    bitmap[(size_t)tile * (FZB_TILE / 64) + tid] = word;
                    cnt = (u32)__popcll(word);
                }
                if (tid < FZB_TILE / 64 && cnt) atomicAdd(&s_cnt, cnt);
            } else if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
            __syncthreads();
            if (tid == 0) tile_counts[tile] = s_cnt;
            __syncthreads();
        }
        hs_c = hs_n; hl_c = hl_n; or_c = or_n;
#pragma unroll
        for (int k = 0; k < 8; k++) qc[k] = qn[k];
        hs_n = hs_f; hl_n = hl_f; or_n = or_f;
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists, cooperative loads (round 3).  The burst form's per-lane 16-byte loads make EVERY piece its own request: a
// wave's 64 haystacks are a contiguous ~5 KB, but one load instruction touches 64 different 128-byte lines and uses 16 bytes of each, and
// with 32 waves per CU (~150 KB of live lines against a 32 KB L1) the lane's next piece misses L1 again - 50 M L1 accesses and as many
// L2 requests for 0.85 GB (profiles/r02_pmc_ragged_filter.txt), ~70 % of what the L2 channels take per cycle.  Removing the 47 % of dead
// lookups alone (the length-sorted view, PERM) made the kernel SLOWER (236 -> 257 us): it is bound by requests, not by instructions.
// Here the WAVE loads its 64 haystacks' byte range with fully coalesced instructions (lane l takes bytes 16 l of each KiB: 8 lines per
// instruction, every line once), passes it through a 4 KiB LDS window of its own (ds_write_b128), and every lane picks its haystack's
// vectors out of the window (ds_read_b128 at its own offset) into the same registers the burst form fills.  No workgroup barrier: LDS
// operations of one wave execute in order.  NV = vectors per haystack held in registers (8: <= 128 bytes, 16: <= 256 bytes).
// ---------------------------------------------------------------------------------------------------
#define FZB_COOP_WIN 4096u
template <typename ET, bool SAN, bool PERM, int NV>
__global__ __launch_bounds__(256) void k1_dfa_ragged_coop(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                          const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                          u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, const u16* __restrict__ perm) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) u8 dfa[];  // table at LDS address 0, then the tile counter, the tile's decision bits, the waves' windows
    const u32 tab_bytes = (FZB_DFA_LDS_BYTES(rows) + 15u) & ~15u;
    u32& s_cnt = *(u32*)(dfa + tab_bytes);
    u32* const s_bits = (u32*)(dfa + tab_bytes + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u8* const wbuf = dfa + tab_bytes + 16 + 128 + (u32)wave * FZB_COOP_WIN;
    dfa_require_lds_base0(dfa);
    dfa_load_lds(dfa, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 deadv = dead * 0x01010101u;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        if (PERM && tid < 32) s_bits[tid] = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int sub = 0; sub < 4; sub++) {
            const u32 li = tile * FZB_TILE + sub * 256 + tid;
            u64 hs = 0;
            u32 hl = 0, orig = 0;
            if (li < count) {
                haystack_span(ends, first + li, hs, hl);
                if (PERM) orig = perm[first + li];
            }
            // the wave's byte range: from its first haystack's start to the (16-byte rounded) end of its last one
            const u32 wfirst = tile * FZB_TILE + sub * 256 + (u32)wave * 64;
            u32 nvalid = wfirst < count ? min(64u, count - wfirst) : 0u;
            nvalid = __builtin_amdgcn_readfirstlane(nvalid);
            uint4 q[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = make_uint4(0, 0, 0, 0);
            const u32 nv = min((hl + 15u) >> 4, (u32)NV);
            if (nvalid) {
                const u32 lo_lo = __builtin_amdgcn_readlane((u32)hs, 0), lo_hi = __builtin_amdgcn_readlane((u32)(hs >> 32), 0);
                const u64 rstart = ((u64)lo_hi << 32) | lo_lo;
                const u64 my_end = (hs + hl + 15) & ~(u64)15;
                const u32 e_lo = __builtin_amdgcn_readlane((u32)my_end, nvalid - 1), e_hi = __builtin_amdgcn_readlane((u32)(my_end >> 32), nvalid - 1);
                const u32 rlen = (u32)((((u64)e_hi << 32) | e_lo) - rstart);  // a wave's range: 64 haystacks of <= 16 NV bytes
                const u32 my_off = (u32)(hs - rstart);
                const u8* rbase = bytes + rstart;
                for (u32 w0 = 0; w0 < rlen; w0 += FZB_COOP_WIN) {
                    uint4 g[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const u32 a = w0 + 1024u * k + 16u * lane;
                        g[k] = a < rlen ? *(const uint4*)(rbase + a) : make_uint4(0, 0, 0, 0);
                    }
                    __builtin_amdgcn_wave_barrier();  // (the previous window's reads are issued before these writes: LDS runs a wave's operations in order)
#pragma unroll
                    for (int k = 0; k < 4; k++) *(uint4*)(wbuf + 1024u * k + 16u * lane) = g[k];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        const u32 rel = my_off + 16u * k - w0;  // < window size only if the vector lies in this window (unsigned wrap otherwise)
                        if ((u32)k < nv && rel < FZB_COOP_WIN) q[k] = *(const uint4*)(wbuf + rel);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            u32 st = 0;
#pragma unroll
            for (int k = 0; k < NV; k++) {  // (no early exit: with a break the loop is not unrolled and q[] goes to scratch memory)
                if (hl <= 16u * k) continue;
                u32 w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
                if (SAN) {
                    const u32 rem = hl - 16u * k;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u32 nvb = rem > 4u * j ? rem - 4u * j : 0u;
                        const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                        w[j] = (w[j] & mask) | (deadv & ~mask);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    st = dfa_step<0>(st, w[j], dfa);
                    st = dfa_step<1>(st, w[j], dfa);
                    st = dfa_step<2>(st, w[j], dfa);
                    st = dfa_step<3>(st, w[j], dfa);
                }
            }
            const bool matched = li < count && hl >= min_len && st >= acc_lo;
            if (PERM) {
                if (matched) atomicOr(&s_bits[orig >> 5], 1u << (orig & 31));
                continue;
            }
            const u64 b = __ballot(matched);
            if (lane == 0) {
                bitmap[(tile * FZB_TILE + sub * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (PERM) {
            __syncthreads();
            if (tid < FZB_TILE / 64) {
                const u64 word = (u64)s_bits[2 * tid] | ((u64)s_bits[2 * tid + 1] << 32);
                bitmap[(size_t)tile * (FZB_TILE / 64) + tid] = word;
                cnt = (u32)__popcll(word);
            }
            if (tid < FZB_TILE / 64 && cnt) atomicAdd(&s_cnt, cnt);
        } else if (lane == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// K1-DFA for ragged lists, LDS-staged and length-sorted (round 3).  What bounded the burst form above (profiles/r02_pmc_ragged_filter.txt):
// every lane's 16-byte vector is its own L1 access (50 M accesses for 0.85 GB: neighbouring lanes read different haystacks), and a wave
// runs to its longest haystack - 47 % of its DFA lookups are issued for lanes whose haystack has ended, in a kernel whose instruction
// issue is 75 % busy.  Here a workgroup
//   1. loads a SEGMENT of the list - a run of consecutive haystacks, up to seg_bytes of the padded-16 stream - into LDS with fully
//      coalesced 16-byte loads (1 KiB per wave instruction, each line fetched once),
//   2. orders the segment's haystacks by their number of 16-byte vectors (descending; wave ballots per class, no atomics),
//   3. hands blocks of 64 consecutive SORTED haystacks to (chain, wave) pairs - block b to chain b / 4 of wave b % 4 - so that the 64
//      lanes of a chain have the same vector count (+- 1 at a class boundary) and a thread's chains get shorter with the chain number:
//      at vector v the live chains of a wave are a PREFIX, run interleaved by dfa_wordP<A> (A = 4, 3, 2, 1),
//   4. reads the bytes from LDS (ds_read_b128; scattered addresses cost nothing there), runs the same v_perm + ds_read_u8 DFA, and
//      sets the decision bit at the haystack's ORIGINAL position with an LDS atomic-or, so the bitmap, the per-tile counts and every
//      later stage are unchanged.
// A haystack longer than the segment buffer is run straight from global memory by one thread (not a case the streaming lists have).
// ---------------------------------------------------------------------------------------------------
#define FZB_RL_NCLS 10  // vector-count classes 0..8 and ">= 9" (haystacks beyond 128 bytes: lengths inside the class differ)
template <int A>
__device__ __forceinline__ void rl_round(u32 (&st)[4], const uint4 (&q)[4], const u8* dfa) {
    u32 sa[A], w[A];
#pragma unroll
    for (int p = 0; p < A; p++) sa[p] = st[p];
#pragma unroll
    for (int p = 0; p < A; p++) w[p] = q[p].x;
    dfa_wordP<A>(sa, w, dfa);
#pragma unroll
    for (int p = 0; p < A; p++) w[p] = q[p].y;
    dfa_wordP<A>(sa, w, dfa);
#pragma unroll
    for (int p = 0; p < A; p++) w[p] = q[p].z;
    dfa_wordP<A>(sa, w, dfa);
#pragma unroll
    for (int p = 0; p < A; p++) w[p] = q[p].w;
    dfa_wordP<A>(sa, w, dfa);
#pragma unroll
    for (int p = 0; p < A; p++) st[p] = sa[p];
}

template <typename ET, bool SAN>
__global__ __launch_bounds__(256) void k1_dfa_ragged_lds(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                         const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                         u32* __restrict__ tile_counts, u32* __restrict__ reset_counters, u32 seg_bytes) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    // ONE dynamic LDS object with the DFA table at address 0 (dfa_lds.h); behind it: the tile's end offsets, the sorted order, the
    // per-(pass, wave) class counts, the tile's decision bits, and the segment buffer
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u8* dfa = lds;
    const u32 tab_bytes = (FZB_DFA_LDS_BYTES(rows) + 15u) & ~15u;
    u32* const s_end = (u32*)(lds + tab_bytes);          // [1024 + 4]: exclusive end of haystack j, relative to the tile's first byte
    u16* const s_order = (u16*)(s_end + FZB_TILE + 4);   // [1024]: sorted position -> index inside the segment
    u32* const s_cnt = (u32*)(s_order + FZB_TILE);       // [16][NCLS]: haystacks of class c in (pass i, wave w); then its exclusive prefix
    u32* const s_tot = s_cnt + 16 * FZB_RL_NCLS;         // [NCLS + 2]
    u32* const s_bits = s_tot + FZB_RL_NCLS + 2;         // [32]: decision bits of the tile
    u32* const s_sum = s_bits + 32;                      // [4]
    u8* const buf = (u8*)(s_sum + 4);                    // [seg_bytes], 16-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dfa_require_lds_base0(lds);
    dfa_load_lds(lds, dfa_g, rows);
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const u32 fillv = SAN ? dead * 0x01010101u : 0u;  // what a chain reads once its haystack has ended: bytes no needle row matches
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u32 i0 = tile * FZB_TILE;
        const u32 nt = min((u32)FZB_TILE, count - i0);
        const u64 g0 = first + i0;
        const u64 tile_base = g0 ? (((u64)ends[g0 - 1] + 15) & ~(u64)15) : 0;
#pragma unroll
        for (int k = 0; k < FZB_TILE / 256; k++) {
            const u32 j = tid + 256 * k;
            if (j < nt) s_end[j] = (u32)((u64)ends[g0 + j] - tile_base);
        }
        if (tid < 32) s_bits[tid] = 0;
        if (tid == 0) s_sum[0] = 0;
        __syncthreads();
        auto start_of = [&](u32 j) -> u32 { return j ? (s_end[j - 1] + 15u) & ~15u : 0u; };
        u32 j0 = 0;
        while (j0 < nt) {
            const u32 start0 = start_of(j0);
            // the segment: haystacks [j0, j1) with start_of(j1) - start0 <= seg_bytes (largest such j1; wave-uniform, everybody computes it)
            u32 lo = j0, hi = nt;
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (((s_end[mid - 1] + 15u) & ~15u) - start0 <= seg_bytes) lo = mid;
                else hi = mid - 1;
            }
            const u32 j1 = lo;
            if (j1 == j0) {
                // one haystack larger than the buffer: straight from global memory, one thread
                if (tid == 0) {
                    const u32 len = s_end[j0] - start0;
                    const u8* h = bytes + tile_base + start0;
                    u32 st = 0;
                    for (u32 k = 0; k < len; k++) st = *(const __attribute__((address_space(3))) u8*)(uintptr_t)(st * FZB_DFA_STRIDE + h[k]);
                    if (len >= min_len && st >= acc_lo) atomicOr(&s_bits[j0 >> 5], 1u << (j0 & 31));
                }
                j0++;
                continue;
            }
            const u32 nseg = j1 - j0;
            const u32 nvec_seg = (((s_end[j1 - 1] + 15u) & ~15u) - start0) >> 4;
            // ---- 1. the segment's bytes: coalesced 16-byte loads, eight in flight per thread -----------------------------------
            {
                const uint4* src = (const uint4*)(bytes + tile_base + start0);
                uint4* dst = (uint4*)buf;
                for (u32 b0 = 0; b0 < nvec_seg; b0 += 256 * 8) {
                    uint4 q[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const u32 idx = b0 + u * 256 + tid;
                        q[u] = idx < nvec_seg ? src[idx] : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const u32 idx = b0 + u * 256 + tid;
                        if (idx < nvec_seg) dst[idx] = q[u];
                    }
                }
            }
            // ---- 2. class (vector count) and rank inside (pass, wave, class) of every haystack of the segment --------------------
            u32 cls[4], rank[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 sidx = tid + 256 * i;
                const bool valid = sidx < nseg;
                u32 c = 0;
                if (valid) {
                    const u32 j = j0 + sidx;
                    const u32 nv = (s_end[j] - start_of(j) + 15u) >> 4;
                    c = min(nv, (u32)FZB_RL_NCLS - 1);
                }
                cls[i] = c;
                u32 rk = 0;
#pragma unroll 1
                for (int k = 0; k < FZB_RL_NCLS; k++) {  // (not unrolled: forty live ballot masks would not fit the scalar registers)
                    const u64 mk = __ballot(valid && c == (u32)k);
                    const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u));  // set bits below my lane
                    rk = c == (u32)k ? below : rk;
                    if (lane == 0) s_cnt[(i * 4 + wave) * FZB_RL_NCLS + k] = (u32)__popcll(mk);
                }
                rank[i] = rk;
            }
            __syncthreads();  // counts complete; the segment's bytes are in LDS
            if (tid < FZB_RL_NCLS) {  // exclusive prefix of class `tid` over the 16 (pass, wave) slots
                u32 sum = 0;
                for (int q = 0; q < 16; q++) {
                    const u32 v = s_cnt[q * FZB_RL_NCLS + tid];
                    s_cnt[q * FZB_RL_NCLS + tid] = sum;
                    sum += v;
                }
                s_tot[tid] = sum;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 sidx = tid + 256 * i;
                if (sidx < nseg) {
                    u32 base = 0;  // descending vector count: classes above mine come first
                    for (int k = FZB_RL_NCLS - 1; k > (int)cls[i]; k--) base += s_tot[k];
                    s_order[base + s_cnt[(i * 4 + wave) * FZB_RL_NCLS + cls[i]] + rank[i]] = (u16)sidx;
                }
            }
            __syncthreads();
            // ---- 3. + 4. the DFA over blocks of 64 sorted haystacks: block p * 4 + wave is chain p of this wave ---------------------
            u32 off[4], len[4], nv[4], jj[4], st[4];
            u32 nvw[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const u32 slot = (u32)(p * 4 + wave) * 64 + lane;
                const bool act = slot < nseg;
                jj[p] = act ? j0 + s_order[slot] : 0xFFFFFFFFu;
                const u32 sj = act ? start_of(jj[p]) : start0;
                off[p] = sj - start0;
                len[p] = act ? s_end[jj[p]] - sj : 0u;
                nv[p] = (len[p] + 15u) >> 4;
                st[p] = 0;
                u32 m = nv[p];  // the chain's vector count for the wave: max over its lanes (equal but for class boundaries / the open class)
                for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_xor(m, o));
                nvw[p] = __builtin_amdgcn_readfirstlane(m);
            }
            // (inside the open class ">= 9 vectors" the blocks are not ordered among themselves: the bound is the maximum, and the live
            // set is taken up to the highest live chain - a finished chain inside it reads fill bytes)
            const u32 vmax = max(max(nvw[0], nvw[1]), max(nvw[2], nvw[3]));
            for (u32 v = 0; v < vmax; v++) {
                uint4 q[4];
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    q[p] = make_uint4(fillv, fillv, fillv, fillv);
                    if (v < nv[p]) {
                        q[p] = *(const uint4*)(buf + off[p] + 16 * v);
                        if (SAN && v + 1 == nv[p]) {  // the zero fill behind the haystack's end could match a needle with a NUL byte
                            const u32 rem = len[p] - 16 * v;
                            u32 w4[4] = {q[p].x, q[p].y, q[p].z, q[p].w};
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const u32 nvb = rem > 4u * k ? rem - 4u * k : 0u;
                                const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                                w4[k] = (w4[k] & mask) | (fillv & ~mask);
                            }
                            q[p] = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        }
                    }
                }
                // live chains are a prefix (sorted descending: nvw[0] >= nvw[1] >= nvw[2] >= nvw[3] up to the open class)
                if (v < nvw[3]) rl_round<4>(st, q, dfa);
                else if (v < nvw[2]) rl_round<3>(st, q, dfa);
                else if (v < nvw[1]) rl_round<2>(st, q, dfa);
                else rl_round<1>(st, q, dfa);
            }
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (jj[p] != 0xFFFFFFFFu && len[p] >= min_len && st[p] >= acc_lo) atomicOr(&s_bits[jj[p] >> 5], 1u << (jj[p] & 31));
            __syncthreads();  // the buffer and the order array are free again
            j0 = j1;
        }
        __syncthreads();
        if (tid < FZB_TILE / 64) {
            const u64 word = (u64)s_bits[2 * tid] | ((u64)s_bits[2 * tid + 1] << 32);
            bitmap[(size_t)tile * (FZB_TILE / 64) + tid] = word;
            atomicAdd(&s_sum[0], (u32)__popcll(word));
        }
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_sum[0];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Level-1 compaction in ONE kernel: every workgroup owns a
// contiguous run of tiles, obtains the number of survivors before its run by reducing the (small, L2-resident)
// per-tile count array itself - redundant across workgroups but far cheaper than a dependent scan launch -
// scans its own tiles' counts in LDS, and expands its bitmap words into the index-ordered survivor list.
// The last workgroup also publishes the total.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact1(const u64* __restrict__ bitmap, const u32* __restrict__ counts, u32 n_items_host, const u32* __restrict__ n_items_ptr,
                                                  const u32* __restrict__ src, u32* __restrict__ out_idx, u32* __restrict__ total_out) {
    // n_items_ptr (device) overrides the host count; src, if given, maps a bit position to the value that is listed
    // (the item-list form of the filter: positions in a candidate list -> haystack indices)
    __shared__ u32 red[4];
    __shared__ u32 pre[256];
    const u32 n_items = n_items_ptr ? *n_items_ptr : n_items_host;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 ntiles = (n_items + FZB_TILE - 1) / FZB_TILE;
    const u32 T = (ntiles + gridDim.x - 1) / gridDim.x;
    const u32 t0 = min(blockIdx.x * T, ntiles), t1 = min(t0 + T, ntiles);
    // Everything this workgroup reads is requested up front - its first batch of tile counts, the first 256 bitmap words of that batch and
    // the counts before its run - so that the three dependent round trips of the straightforward order (prefix, batch counts, words)
    // overlap into one; for the 10 M-haystack list a workgroup has ~10 tiles = 152 words, i.e. nothing is left to load afterwards.
    const u32 nwords = (n_items + 63) / 64;
    const u32 nt_first = min(256u, t1 - t0);
    const u32 c_first = (u32)tid < nt_first ? counts[t0 + tid] : 0u;
    const u32 w_first = t0 * (FZB_TILE / 64) + tid;
    const u64 bits_first = (w_first < (t0 + nt_first) * (FZB_TILE / 64) && w_first < nwords) ? bitmap[w_first] : 0ull;
    // survivors before tile t0
    // (this sum is the kernel's critical path - up to ntiles counts per workgroup: 16-byte loads, TWELVE in flight per thread, so that the
    // 10 M-haystack list's 9766 counts are one round trip for every workgroup instead of three)
    u32 part = 0;
    {
        const uint4* c4 = (const uint4*)counts;
        const u32 n4 = t0 / 4;
        for (u32 i0 = 0; i0 < n4; i0 += 12 * 256) {
            uint4 v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const u32 i = i0 + k * 256 + tid;
                v[k] = i < n4 ? c4[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 12; k++) part += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        for (u32 k = 4 * n4 + tid; k < t0; k += 256) part += counts[k];
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    u32 base = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    for (u32 tb = t0; tb < t1; tb += 256) {  // batches of up to 256 tiles
        const u32 nt = min(256u, t1 - tb);
        const u32 c = tb == t0 ? c_first : ((u32)tid < nt ? counts[tb + tid] : 0u);
        // exclusive scan of c over the batch
        u32 incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; w++) wb += red[w];
        pre[tid] = base + wb + incl - c;
        const u32 batch_total = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
        // expand the bitmap words of these tiles: the 16 words of a tile sit in 16 consecutive lanes (w0 is a multiple of 16),
        // so a word's offset inside its tile is a 16-lane segmented scan of the popcounts - no re-reading of the earlier words
        const u32 w0 = tb * (FZB_TILE / 64), w1 = (tb + nt) * (FZB_TILE / 64);
        for (u32 wb0 = w0; wb0 < w1; wb0 += 256) {  // uniform trip count: the shuffles below need every lane
            const u32 w = wb0 + tid;
            const bool live = w < w1 && w < nwords;
            u64 bits = (tb == t0 && wb0 == w0) ? bits_first : (live ? bitmap[w] : 0ull);
            const u32 c = (u32)__popcll(bits);
            u32 incl = c;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const u32 v = __shfl_up(incl, off, 16);
                if ((lane & 15) >= off) incl += v;
            }
            if (!bits) continue;
            const u32 tile = w / (FZB_TILE / 64);
            u32 pos = pre[tile - tb] + incl - c;
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                out_idx[pos++] = src ? src[w * 64 + b] : w * 64 + b;
            }
        }
        base += batch_total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *total_out = base;  // every earlier workgroup's tiles precede this one's
}

// ---------------------------------------------------------------------------------------------------
// Level-2 compaction (after the lane-exact prefilter re-decided the filter's survivors), same scheme, one kernel:
// here most items are kept, so the work is laid out per ITEM - the gathers of the kept items' haystack index and
// window are coalesced - with the per-word offsets of a batch of tiles prepared once in LDS.
// ---------------------------------------------------------------------------------------------------
#define FZB_C2_BATCH 64  // tiles per batch: 64 x 16 word offsets in LDS
__global__ __launch_bounds__(256) void k_compact2(const u64* __restrict__ bitmap, const u32* __restrict__ counts, const u32* __restrict__ n_items_ptr,
                                                  const u32* __restrict__ in_idx, const u32* __restrict__ in_win, u32* __restrict__ out_idx, u32* __restrict__ out_win,
                                                  u32* __restrict__ total_out) {
    __shared__ u32 red[4];
    __shared__ u32 pre[FZB_C2_BATCH];
    __shared__ u32 woff[FZB_C2_BATCH * (FZB_TILE / 64)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 n_items = *n_items_ptr;
    const u32 ntiles = (n_items + FZB_TILE - 1) / FZB_TILE;
    const u32 T = (ntiles + gridDim.x - 1) / gridDim.x;
    const u32 t0 = min(blockIdx.x * T, ntiles), t1 = min(t0 + T, ntiles);
    // (16-byte loads, four in flight per thread: this sum is the kernel's critical path - up to ntiles counts per workgroup)
    u32 part = 0;
    {
        const uint4* c4 = (const uint4*)counts;
        const u32 n4 = t0 / 4;
        u32 i = tid;
        for (; i + 768 < n4; i += 1024) {
            const uint4 a = c4[i], b = c4[i + 256], c = c4[i + 512], d = c4[i + 768];
            part += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        }
        for (; i < n4; i += 256) {
            const uint4 a = c4[i];
            part += a.x + a.y + a.z + a.w;
        }
        for (u32 k = 4 * n4 + tid; k < t0; k += 256) part += counts[k];
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    u32 base = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const u32 nwords = (n_items + 63) / 64;
    for (u32 tb = t0; tb < t1; tb += FZB_C2_BATCH) {
        const u32 nt = min((u32)FZB_C2_BATCH, t1 - tb);
        // exclusive scan of the batch's tile counts (nt <= 64: one wave)
        if (wave == 0) {
            const u32 c = (u32)lane < nt ? counts[tb + lane] : 0u;
            u32 incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const u32 v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if ((u32)lane < nt) pre[lane] = base + incl - c;
            if (lane == 63) red[0] = incl;
        }
        // offset of every bitmap word inside its tile: 16-lane segmented scan of the popcounts
        const u32 w0 = tb * (FZB_TILE / 64), w1 = (tb + nt) * (FZB_TILE / 64);
        for (u32 wb0 = w0; wb0 < w1; wb0 += 256) {
            const u32 w = wb0 + tid;
            const u32 c = (w < w1 && w < nwords) ? (u32)__popcll(bitmap[w]) : 0u;
            u32 incl = c;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const u32 v = __shfl_up(incl, off, 16);
                if ((lane & 15) >= off) incl += v;
            }
            if (w < w1) woff[w - w0] = incl - c;
        }
        __syncthreads();
        const u32 j0 = tb * FZB_TILE, j1 = min((tb + nt) * FZB_TILE, n_items);
        for (u32 j = j0 + tid; j < j1; j += 256) {
            const u32 w = j >> 6;
            const u64 bits = bitmap[w];
            if (!((bits >> (j & 63)) & 1)) continue;
            const u32 pos = pre[j / FZB_TILE - tb] + woff[w - w0] + (u32)__popcll(bits & (((u64)1 << (j & 63)) - 1));
            out_idx[pos] = in_idx ? in_idx[j] : j;
            *(uint2*)(out_win + 2 * (size_t)pos) = *(const uint2*)(in_win + 2 * (size_t)j);
        }
        base += red[0];
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *total_out = base;
}

// ---------------------------------------------------------------------------------------------------
// The same accept tests over an ITEM LIST (positions j -> haystack items[j]) whose length lives in device memory:
// the narrowing step of the multi-pattern composition (src/matcher/multi.rs:102-118 re-matches each further pattern
// against only the haystacks that survived the previous ones).  Candidates are a small, sparse subset, so this is
// one plain thread per item; the decisions leave in the usual bitmap + per-1024 counts, indexed by list position.
// ---------------------------------------------------------------------------------------------------
template <typename TW, int MODE, typename ET>
__global__ __launch_bounds__(256) void k1_items(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, const u32* __restrict__ items,
                                                const u32* __restrict__ n_items_ptr, const u64* __restrict__ Tg, int rows, int need, u32 min_len,
                                                u64* __restrict__ bitmap, u32* __restrict__ tile_counts) {
    __shared__ TW T[256];
    __shared__ u32 s_cnt;
    const int tid = threadIdx.x;
    T[tid] = (TW)Tg[tid];
    const u32 count = *n_items_ptr;
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int p = 0; p < FZB_TILE / 256; p++) {
            const u32 j = tile * FZB_TILE + p * 256 + tid;
            bool matched = false;
            if (j < count) {
                u64 s;
                u32 L;
                haystack_span(ends, first + items[j], s, L);
                if (L >= min_len) {
                    const uint4* vp = (const uint4*)(bytes + s);
                    TW st = (MODE == 1) ? (TW)1 : (TW)~(TW)0;
                    const u32 nvec = (L + 15) >> 4;
                    for (u32 v = 0; v < nvec; v++) {
                        const uint4 q = vp[v];
                        const u32 rem = L - 16 * v;
                        filter_word<TW, MODE>(st, q.x, rem, T);
                        if (rem > 4) filter_word<TW, MODE>(st, q.y, rem - 4, T);
                        if (rem > 8) filter_word<TW, MODE>(st, q.z, rem - 8, T);
                        if (rem > 12) filter_word<TW, MODE>(st, q.w, rem - 12, T);
                    }
                    if (MODE == 1) {
                        matched = (st >> rows) & 1;
                    } else {
                        const TW low = rows >= (int)(8 * sizeof(TW)) ? (TW)~(TW)0 : (((TW)1 << rows) - 1);
                        const TW z = ~st & low;
                        const int lcs = sizeof(TW) == 8 ? __popcll((u64)z) : __popc((u32)z);
                        matched = lcs >= need;
                    }
                }
            }
            const u64 b = __ballot(matched);
            if (lane_id() == 0) {
                bitmap[(tile * FZB_TILE + p * 256) / 64 + (tid >> 6)] = b;
                cnt += __popcll(b);
            }
        }
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts[tile] = s_cnt;
        __syncthreads();
    }
}

void fzb_launch_filter_items(const CorpusDev& c, u64 first, const u32* items, const u32* n_items_ptr, const u64* table, int rows, int mode, int need, u32 min_len,
                             u64* bitmap, u32* tile_counts, int grid, hipStream_t st) {
    // MODE 1 with one-hot state needs rows + 1 bits, MODE 2 rows bits
    const bool w64 = (mode == 1) ? rows > 31 : rows > 32;
#define FZB_K1I(TW, MODE, ET) hipLaunchKernelGGL((k1_items<TW, MODE, ET>), dim3(grid), dim3(256), 0, st, c.bytes, (const ET*)c.ends, first, items, n_items_ptr, table, rows, need, min_len, bitmap, tile_counts)
    if (c.ends_u64) {
        if (mode == 1) { if (w64) FZB_K1I(u64, 1, u64); else FZB_K1I(u32, 1, u64); }
        else           { if (w64) FZB_K1I(u64, 2, u64); else FZB_K1I(u32, 2, u64); }
    } else {
        if (mode == 1) { if (w64) FZB_K1I(u64, 1, u32); else FZB_K1I(u32, 1, u32); }
        else           { if (w64) FZB_K1I(u64, 2, u32); else FZB_K1I(u32, 2, u32); }
    }
#undef FZB_K1I
}

// ---------------------------------------------------------------------------------------------------
// host-side launch wrappers (called from host.hip)
// ---------------------------------------------------------------------------------------------------
void fzb_launch_filter(const CorpusDev& c, u64 first, u32 count, const u64* table, const u8* dfa, u32 dead, int rows, int mode, int need, u32 min_len,
                       u64* bitmap, u32* tile_counts, u32* reset_counters, int grid, hipStream_t st, u64* bitmap_m, u32* tile_counts_m, u64* reject_bits, u32* tile_rejects, int nul_safe,
                       int acc_lo, const u8* cdfa, u32 cdfa_bytes, int cdfa_K, int cdfa_G) {
    // mode 1: `dfa` has rows + 1 states, start state 0, and accepts in the states >= acc (the subsequence / unicode / KMP automata: the last
    // state; the LCS automaton of a typo configuration: every state whose LCS reaches the need)
    const u32 acc = acc_lo < 0 ? (u32)rows : (u32)acc_lo;
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    if (grid > (int)ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    if (mode == 1) {
        const size_t lds = (size_t)(rows + 1) * FZB_DFA_STRIDE + 16;  // table + the tile counter
        const bool shortc = c.max_len != 0 && c.max_len <= 32;  // every haystack fits the two pre-requested vectors
        if (shortc) {
            if (c.ends_u64) hipLaunchKernelGGL((k1_dfa<u64>), dim3(grid), dim3(256), lds, st, c.bytes, (const u64*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, c.uniform_len);
            else hipLaunchKernelGGL((k1_dfa<u32>), dim3(grid), dim3(256), lds, st, c.bytes, (const u32*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, c.uniform_len);
        } else {
            // ragged lists: one haystack per thread, 6 resident workgroups per CU (measured on the 8..128-byte list: 311 us
            // vs 498 us for the 4-way kernel at full occupancy, whose L2 footprint re-fetched every line 2-4 times).
            // Tried in round 2 and dropped: sorting a tile's haystacks by length class in LDS before the DFA (no SIMT length
            // divergence, no sanitising of the zero fill) - 460-490 us instead of 298: a wave's 64 lanes then touch 64 scattered
            // 128-byte lines instead of ~34 adjacent ones, and the kernel is bound by cache transactions, not by instruction issue.
            int rgrid = std::min<int>((grid / 8) * 6, (int)ntiles);
            if (rgrid < 1) rgrid = 1;
            // the class-composite automaton of `dfa` (host-built: fzb_matcher_create): one dependent lookup per 4 (or 2) bytes
            static const bool no_cdfa = getenv("FZB_NO_CDFA") != nullptr;
            static const int cwg = getenv("FZB_CDFA_WGS") ? atoi(getenv("FZB_CDFA_WGS")) : 5;  // resident workgroups per CU (C4 shard: 8 -> 247 us, 5 -> 232, 4 -> 229)
            static const bool no_view = getenv("FZB_FILTER_VIEW") && atoi(getenv("FZB_FILTER_VIEW")) == 0;
            static const int vwg = getenv("FZB_VIEW_WGS") ? atoi(getenv("FZB_VIEW_WGS")) : 6;  // (C4 shard: 8 -> 190 us, 6 -> 187, 4 -> 194)
            if (cdfa && !no_cdfa && (cdfa_G == 4 || cdfa_G == 2) && c.vbytes && !no_view && first % FZB_TILE == 0 && (first + count == c.n || count % FZB_TILE == 0) &&
                c.max_len <= 256) {
                const size_t lds_v = ((cdfa_bytes + 15) & ~(size_t)15) + 16 + 128;
                const int g = std::max(1, std::min<int>((grid / 8) * vwg, (int)ntiles));
                u32 kg = 1;
                for (int i = 0; i < cdfa_G; i++) kg *= (u32)cdfa_K;
                if (getenv("FZB_CDFA_NODFA")) cdfa_K = 0xFFFF;
#define FZB_K1V(SAN, G, NV) hipLaunchKernelGGL((k1_cdfa_view<SAN, G, NV>), dim3(g), dim3(256), lds_v, st, c.vbytes, c.vgofs, c.vgnv, c.vlen, c.vperm, first, count, cdfa, cdfa_bytes, (u32)cdfa_K, kg, min_len, dead, acc, bitmap, tile_counts, reset_counters)
#define FZB_K1V_NV(SAN, G) do { if (c.max_len <= 128) FZB_K1V(SAN, G, 8); else FZB_K1V(SAN, G, 16); } while (0)
#define FZB_K1V_G(SAN) do { if (cdfa_G == 4) FZB_K1V_NV(SAN, 4); else FZB_K1V_NV(SAN, 2); } while (0)
                if (nul_safe) FZB_K1V_G(false); else FZB_K1V_G(true);
#undef FZB_K1V_G
#undef FZB_K1V_NV
#undef FZB_K1V
                return;
            }
            if (cdfa && !no_cdfa && (cdfa_G == 4 || cdfa_G == 2)) {
                const size_t lds_c = ((cdfa_bytes + 15) & ~(size_t)15) + 16;
                const int g = std::max(1, std::min<int>((grid / 8) * cwg, (int)ntiles));
                u32 kg = 1;
                for (int i = 0; i < cdfa_G; i++) kg *= (u32)cdfa_K;
                static const bool nodfa = getenv("FZB_CDFA_NODFA") != nullptr;
                if (nodfa) cdfa_K = 0xFFFF;
#define FZB_K1CD(ET, SAN, G) hipLaunchKernelGGL((k1_cdfa_ragged<ET, SAN, G>), dim3(g), dim3(256), lds_c, st, c.bytes, (const ET*)c.ends, first, count, cdfa, cdfa_bytes, (u32)cdfa_K, kg, min_len, dead, acc, bitmap, tile_counts, reset_counters)
#define FZB_K1CD_G(ET, SAN) do { if (cdfa_G == 4) FZB_K1CD(ET, SAN, 4); else FZB_K1CD(ET, SAN, 2); } while (0)
                if (c.ends_u64) { if (nul_safe) FZB_K1CD_G(u64, false); else FZB_K1CD_G(u64, true); }
                else            { if (nul_safe) FZB_K1CD_G(u32, false); else FZB_K1CD_G(u32, true); }
#undef FZB_K1CD_G
#undef FZB_K1CD
                return;
            }
            static const int burst = getenv("FZB_RAGGED_BURST") ? atoi(getenv("FZB_RAGGED_BURST")) : 1;  // 0 = the rolling form, for comparison
            static const int bwgs = getenv("FZB_RAGGED_WGS") ? atoi(getenv("FZB_RAGGED_WGS")) : 8;
            // LDS-staged, length-sorted form (k1_dfa_ragged_lds), OPT-IN (FZB_RAGGED_LDS=1): measured 0.53-0.99 ms against the burst form's
            // 0.234 ms on the C4 shard (profiles/r03_ragged_filter_variants.txt) - with the bytes resident in LDS only ~2 100 haystacks fit a
            // CU, 3 waves per SIMD with 1-2 live DFA chains each, and the dependent lookup chain is latency-bound.
            // FZB_RAGGED_SEG = segment buffer in bytes, FZB_RAGGED_LDS_WGS = resident workgroups per CU it is launched with
            static const int use_lds = getenv("FZB_RAGGED_LDS") ? atoi(getenv("FZB_RAGGED_LDS")) : 0;
            static const int seg_env = getenv("FZB_RAGGED_SEG") ? atoi(getenv("FZB_RAGGED_SEG")) : 32768;
            static const int lwgs = getenv("FZB_RAGGED_LDS_WGS") ? atoi(getenv("FZB_RAGGED_LDS_WGS")) : 3;
            const u32 seg = ((u32)std::max(seg_env, 1024) + 15u) & ~15u;
            const size_t fixed = (((size_t)(rows + 1) * FZB_DFA_STRIDE + 15) & ~(size_t)15) + (FZB_TILE + 4) * 4 + FZB_TILE * 2 + (16 * FZB_RL_NCLS + FZB_RL_NCLS + 2 + 32 + 4) * 4 + 16;
            if (use_lds && fixed + seg <= 64 * 1024) {
                const int g = std::max(1, std::min<int>((grid / 8) * lwgs, (int)ntiles));
#define FZB_K1L(ET, SAN) hipLaunchKernelGGL((k1_dfa_ragged_lds<ET, SAN>), dim3(g), dim3(256), fixed + seg, st, c.bytes, (const ET*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, seg)
                if (c.ends_u64) { if (nul_safe) FZB_K1L(u64, false); else FZB_K1L(u64, true); }
                else            { if (nul_safe) FZB_K1L(u32, false); else FZB_K1L(u32, true); }
#undef FZB_K1L
                return;
            }
            // the corpus' length-sorted filter view (built by fzb_corpus_upload for ragged lists): whole tiles only
            const bool view = false;  // (round 3's first, non-interleaved sorted view is gone: its measurements are in profiles/r03_ragged_filter_variants.txt)
            // software-pipelined burst form (three haystacks in flight per thread)
            static const int pipe = getenv("FZB_RAGGED_PIPE") ? atoi(getenv("FZB_RAGGED_PIPE")) : 0;  // opt-in: measured 244-253 us vs the burst form's 234 us (profiles/r03_ragged_filter_variants.txt)
            static const int pwgs = getenv("FZB_RAGGED_PIPE_WGS") ? atoi(getenv("FZB_RAGGED_PIPE_WGS")) : 5;
            if (pipe) {
                const int g = std::max(1, std::min<int>((grid / 8) * pwgs, (int)ntiles));
                const size_t lds_p = lds + 16 + 32 * 4;
#define FZB_K1P(ET, SAN, PERM, B, E, P) hipLaunchKernelGGL((k1_dfa_ragged_pipe<ET, SAN, PERM>), dim3(g), dim3(256), lds_p, st, B, (const ET*)E, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, P)
                if (view) { if (nul_safe) FZB_K1P(u32, false, true, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1P(u32, true, true, c.bytes, c.ends, (const u16*)nullptr); }
                else if (c.ends_u64) { if (nul_safe) FZB_K1P(u64, false, false, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1P(u64, true, false, c.bytes, c.ends, (const u16*)nullptr); }
                else { if (nul_safe) FZB_K1P(u32, false, false, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1P(u32, true, false, c.bytes, c.ends, (const u16*)nullptr); }
#undef FZB_K1P
                return;
            }
            // cooperative (coalesced, LDS-transposed) loads: every haystack's vectors must fit the registers of the NV = 8 / 16 forms
            static const int coop = getenv("FZB_RAGGED_COOP") ? atoi(getenv("FZB_RAGGED_COOP")) : 0;  // opt-in: measured 234-292 us vs 234 us
            static const int cwgs = getenv("FZB_RAGGED_COOP_WGS") ? atoi(getenv("FZB_RAGGED_COOP_WGS")) : 8;
            if (coop && c.max_len != 0 && c.max_len <= 256) {
                const size_t lds_c = (((size_t)(rows + 1) * FZB_DFA_STRIDE + 15) & ~(size_t)15) + 16 + 128 + 4 * FZB_COOP_WIN;
                const int g = std::max(1, std::min<int>((grid / 8) * cwgs, (int)ntiles));
                const bool nv8 = c.max_len <= 128;
#define FZB_K1C(ET, SAN, PERM, NV, B, E, P) hipLaunchKernelGGL((k1_dfa_ragged_coop<ET, SAN, PERM, NV>), dim3(g), dim3(256), lds_c, st, B, (const ET*)E, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, P)
#define FZB_K1C_NV(ET, SAN, PERM, B, E, P) do { if (nv8) FZB_K1C(ET, SAN, PERM, 8, B, E, P); else FZB_K1C(ET, SAN, PERM, 16, B, E, P); } while (0)
                if (lds_c <= 64 * 1024) {
                    if (view) { if (nul_safe) FZB_K1C_NV(u32, false, true, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1C_NV(u32, true, true, c.bytes, c.ends, (const u16*)nullptr); }
                    else if (c.ends_u64) { if (nul_safe) FZB_K1C_NV(u64, false, false, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1C_NV(u64, true, false, c.bytes, c.ends, (const u16*)nullptr); }
                    else { if (nul_safe) FZB_K1C_NV(u32, false, false, c.bytes, c.ends, (const u16*)nullptr); else FZB_K1C_NV(u32, true, false, c.bytes, c.ends, (const u16*)nullptr); }
                    return;
                }
#undef FZB_K1C_NV
#undef FZB_K1C
            }
            if (view) {
                rgrid = std::max(1, std::min<int>((grid / 8) * bwgs, (int)ntiles));
                const size_t lds_p = lds + 16 + 32 * 4;
                if (nul_safe) hipLaunchKernelGGL((k1_dfa_ragged_burst<u32, false, true>), dim3(rgrid), dim3(256), lds_p, st, c.bytes, (const u32*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, (const u16*)nullptr);
                return;
            }
            if (burst) {
                rgrid = std::max(1, std::min<int>((grid / 8) * bwgs, (int)ntiles));
#define FZB_K1B(ET, SAN) hipLaunchKernelGGL((k1_dfa_ragged_burst<ET, SAN>), dim3(rgrid), dim3(256), lds, st, c.bytes, (const ET*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters)
                if (c.ends_u64) { if (nul_safe) FZB_K1B(u64, false); else FZB_K1B(u64, true); }
                else            { if (nul_safe) FZB_K1B(u32, false); else FZB_K1B(u32, true); }
#undef FZB_K1B
                return;
            }
#define FZB_K1R(ET, SAN) hipLaunchKernelGGL((k1_dfa_ragged<ET, 1, SAN>), dim3(rgrid), dim3(256), lds, st, c.bytes, (const ET*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters)
            if (c.ends_u64) { if (nul_safe) FZB_K1R(u64, false); else FZB_K1R(u64, true); }
            else            { if (nul_safe) FZB_K1R(u32, false); else FZB_K1R(u32, true); }
#undef FZB_K1R
        }
        return;
    }
    const bool w64 = (mode == 1) ? rows > 31 : rows > 32;
    if (mode == 2 && bitmap_m) {  // LCS filter with the "nothing to spare" bit (typo configurations on the short-haystack path)
        const MargOut mo{bitmap_m, tile_counts_m, reject_bits, tile_rejects};
#define FZB_K1M(TW, ET) hipLaunchKernelGGL((k1_filter<TW, 2, ET, true>), dim3(grid), dim3(256), 0, st, c.bytes, (const ET*)c.ends, first, count, table, rows, need, min_len, bitmap, tile_counts, reset_counters, mo, c.uniform_len)
        if (c.ends_u64) { if (w64) FZB_K1M(u64, u64); else FZB_K1M(u32, u64); }
        else            { if (w64) FZB_K1M(u64, u32); else FZB_K1M(u32, u32); }
#undef FZB_K1M
        return;
    }
#define FZB_K1(TW, MODE, ET) hipLaunchKernelGGL((k1_filter<TW, MODE, ET>), dim3(grid), dim3(256), 0, st, c.bytes, (const ET*)c.ends, first, count, table, rows, need, min_len, bitmap, tile_counts, reset_counters, MargOut{}, c.uniform_len)
    if (c.ends_u64) {
        if (mode == 1) { if (w64) FZB_K1(u64, 1, u64); else FZB_K1(u32, 1, u64); }
        else           { if (w64) FZB_K1(u64, 2, u64); else FZB_K1(u32, 2, u64); }
    } else {
        if (mode == 1) { if (w64) FZB_K1(u64, 1, u32); else FZB_K1(u32, 1, u32); }
        else           { if (w64) FZB_K1(u64, 2, u32); else FZB_K1(u32, 2, u32); }
    }
#undef FZB_K1
}

// Exclusive prefix of the per-tile reject counts (decide form of k2a_window) - only when something was rejected at all, which on
// real lists is about one haystack in 1e5 of the marginal ones: one workgroup, and an immediate return otherwise.
__global__ __launch_bounds__(1024) void k_scan_rejects(const u32* __restrict__ tile_rejects, u32 ntiles, const u32* __restrict__ reject_count, u32* __restrict__ rej_prefix) {
    if (*reject_count == 0) return;
    __shared__ u32 wsum[16];
    __shared__ u32 carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < ntiles; t0 += 1024) {
        const u32 t = t0 + tid;
        const u32 c = t < ntiles ? tile_rejects[t] : 0u;
        u32 incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; w++) wb += wsum[w];
        if (t < ntiles) rej_prefix[t] = carry + wb + incl - c;
        __syncthreads();
        if (tid == 1023) carry += wb + incl;
        __syncthreads();
    }
}
void fzb_launch_scan_rejects(const u32* tile_rejects, u32 ntiles, const u32* reject_count, u32* rej_prefix, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_rejects, dim3(1), dim3(1024), 0, st, tile_rejects, ntiles, reject_count, rej_prefix);
}

void fzb_launch_compact1(const u64* bitmap, const u32* counts, u32 n_items, const u32* n_items_ptr, const u32* src, u32* out_idx, u32* total_out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_compact1, dim3(grid), dim3(256), 0, st, bitmap, counts, n_items, n_items_ptr, src, out_idx, total_out);
}

void fzb_launch_compact2(const u64* bitmap, const u32* counts, const u32* n_items_ptr, const u32* in_idx, const u32* in_win, u32* out_idx, u32* out_win, u32* total_out,
                         int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_compact2, dim3(grid), dim3(256), 0, st, bitmap, counts, n_items_ptr, in_idx, in_win, out_idx, out_win, total_out);
}
