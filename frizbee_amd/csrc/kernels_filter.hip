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
#include "compact1.h"
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
// (Round 5 measured an instantiation without per-lane lengths for uniform 32-byte lists and the table at a 256-byte row pitch - 7.5 M instead of
// 10.7 M VALU instructions per launch, the same 57 us, more LDS bank conflicts; round 6 closed the topic: profiles/r06_corun.txt, profiles/HISTORY.md.)
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
            // (plain loads: with the non-temporal hint the second vector's request finds the line gone from the vector cache - a wave's two
            // load instructions share every line of a 32-byte-record list - and the kernel takes 62 us instead of 57)
            if (hl[p] > 0) v0[p] = vp[0];
            if (hl[p] > 16) v1[p] = vp[1];
        }
        u32 st[4] = {0, 0, 0, 0};
        // short haystacks (<= 32 bytes): both vectors are already in flight
        if (hl[0] >= 16 && hl[1] >= 16 && hl[2] >= 16 && hl[3] >= 16) {
            { const u32 w[4] = {v0[0].x, v0[1].x, v0[2].x, v0[3].x}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v0[0].y, v0[1].y, v0[2].y, v0[3].y}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v0[0].z, v0[1].z, v0[2].z, v0[3].z}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v0[0].w, v0[1].w, v0[2].w, v0[3].w}; dfa_word4<true>(st, w, dfa); }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++) st[p] = dfa_partial<true>(st[p], v0[p], hl[p] >= 16 ? 16u : hl[p], dfa);
        }
        if (hl[0] >= 32 && hl[1] >= 32 && hl[2] >= 32 && hl[3] >= 32) {
            { const u32 w[4] = {v1[0].x, v1[1].x, v1[2].x, v1[3].x}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v1[0].y, v1[1].y, v1[2].y, v1[3].y}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v1[0].z, v1[1].z, v1[2].z, v1[3].z}; dfa_word4<true>(st, w, dfa); }
            { const u32 w[4] = {v1[0].w, v1[1].w, v1[2].w, v1[3].w}; dfa_word4<true>(st, w, dfa); }
        } else {
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (hl[p] > 16) st[p] = dfa_partial<true>(st[p], v1[p], hl[p] >= 32 ? 16u : hl[p] - 16, dfa);
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
// K1-DFA for ragged lists, burst form: a thread requests ALL vectors of its haystack (up to 8 = 128 bytes per round) back to
// back and only then runs the DFA over them from registers.  (SAN = false, a needle without a NUL byte: the zero fill behind a haystack's end matches
// no needle row and the vectors go through the DFA unmasked.)  In the rolling form of rounds 1-2 (one vector ahead) a 128-byte line was touched by ~8 loads
// of a wave that are separated by the DFA work of every resident wave, and the CU's footprint (24 waves x 64 haystacks x up to
// 128 B) is far beyond the vector L1, so most of those touches re-fetch the line from L2; issued back to back they hit.
// C4 shard: 297 -> 242 us.  (Two or four haystacks per thread as interleaved DFA chains on top of it: 276 / 300 us, and again 281 / 405 us after
// the lookups were cut to v_perm + ds_read with one wait per round - with 8 waves per SIMD the lookup chain is already hidden; neither that
// nor removing a quarter of the loop's instructions moved the kernel, so it is bound by its cache transactions: every 16-byte piece of a
// lane is its own request, ~35 lines per wave instruction.)
// ---------------------------------------------------------------------------------------------------
template <typename ET, bool SAN>
__global__ __launch_bounds__(256) void k1_dfa_ragged_burst(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count,
                                                           const u8* __restrict__ dfa_g, int rows, u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap,
                                                           u32* __restrict__ tile_counts, u32* __restrict__ reset_counters) {
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
// K1-DFA for ragged lists over the CLASS-COMPOSITE automaton (round 3).  Four structural variants (profiles/r03_ragged_filter_variants.txt) left the burst form's time
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
// LEN: the kernel reads the haystacks' lengths (vlen: 2 of the view's ~83 bytes per haystack on the C4 shard).  They are needed to sanitise a
// last vector (SAN), for the stage's header (STAGE), to skip an outlier's lane (0xFFFF) and for `length >= min_len` - but with the zero fill
// harmless (!SAN) an outlier's lane holds zero vectors and cannot leave state 0, and every automaton the view kernel runs accepts only
// haystacks of at least min_len bytes (subsequence / unicode: the needle's bytes all occurred; LCS >= rows - k bytes matched): without SAN
// the launcher passes LEN = false and the array is not touched.
// (Rounds 4-5 also had a STAGE form - the accepting lanes copied their vectors into a per-tile block for the classifier and the scorers; the
// 48 MB copy-out cost the streaming kernel more than the scorers' gathers returned: profiles/HISTORY.md.)
// TPB = 1024 (short lists: the launcher's choice): sixteen waves, ONE group each - a tile is then one load round trip and one walk instead of four
// in a row, which is what a list of fewer tiles than the chip holds workgroups pays (a fixed ~ 13 us of the 256-thread form).
template <bool SAN, int G, int NV, bool LEN = true, int TPB = 256>
__global__ __launch_bounds__(TPB) void k1_cdfa_view(const u8* __restrict__ vbytes, const u32* __restrict__ vgofs, const u8* __restrict__ vgnv, const u16* __restrict__ vlen,
                                                    const u16* __restrict__ vperm, u64 first, u32 count, const u8* __restrict__ cdfa_g, u32 cdfa_bytes, u32 K, u32 KG,
                                                    u32 min_len, u32 dead, u32 acc_lo, u64* __restrict__ bitmap, u32* __restrict__ tile_counts, u32* __restrict__ reset_counters) {
    if (blockIdx.x == 0 && threadIdx.x < 16) reset_counters[threadIdx.x] = 0;
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tab_bytes = (cdfa_bytes + 15u) & ~15u;
    u32& s_cnt = *(u32*)(lds + tab_bytes);
    u32* const s_bits = (u32*)(lds + tab_bytes + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dfa_require_lds_base0(lds);
    for (u32 i = tid * 4; i < tab_bytes; i += TPB * 4) *(u32*)(lds + i) = i < cdfa_bytes ? *(const u32*)(cdfa_g + i) : 0u;
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
        for (int gi = 0; gi < 16 / (TPB / 64); gi++) {
            const u32 p = tile * FZB_TILE + (u32)(gi * (TPB / 64) + wave) * 64 + lane;  // sorted position (relative to `first`)
            const u32 gl = tile * (FZB_TILE / 64) + (u32)(gi * (TPB / 64) + wave);       // its group
            if (gl * 64 >= count) continue;
            // vgnv: vectors per member | (bytes stored per member of the LAST row / 4 - 1) << 5 (round 5: the group's last row is as narrow as
            // its longest member's tail allows - 4, 8, 12 or 16 bytes per lane, a contiguous 256..1024 bytes for the wave)
            const u32 code = __builtin_amdgcn_readfirstlane((u32)vgnv[g_first + gl]);
            const u32 nv = code & 31u, tw = ((code >> 5) + 1u) * 4u;
            const u8* gblock = vbytes + (size_t)__builtin_amdgcn_readfirstlane(vgofs[g_first + gl]) * 16;
            const u8* base = gblock + (u32)lane * 16;
            u32 hl = 0, orig = 0;
            if (p < count) { hl = LEN ? (u32)vlen[first + p] : 0u; orig = vperm[first + p]; }
            uint4 q[NV];
            uint4 tail = make_uint4(0, 0, 0, 0);
            if (nv) {  // (wave-uniform width: one load instruction of the width the group was stored with)
                const u8* tp = gblock + (size_t)(nv - 1) * 1024 + (size_t)lane * tw;
                if (tw == 16) tail = load16_stream<true>((const uint4*)tp);
                else tail = load_narrow_stream(tp, tw, true);
            }
            // (non-temporal: a wave's load covers whole lines that nothing reads again - the stream no longer displaces what the later stages
            // re-read: C4 shard 190 -> 183 us).  The FULL rows land in q[0 .. nv-2]; the narrow last row stays in `tail` and is the automaton's
            // last step (selecting it into q[nv-1] cost a chain of scalar branches and a full wait per vector)
#pragma unroll
            for (int k = 0; k < NV; k++) q[k] = (u32)k + 1 < nv ? load16_stream<true>((const uint4*)(base + (size_t)k * 1024)) : make_uint4(0, 0, 0, 0);
            u32 st = 0;
            // one vector of the lane's haystack through the class-composite automaton (vector index kk: only the sanitising form needs it)
            auto step = [&](const uint4& v, u32 kk) {
                u32 w[4] = {v.x, v.y, v.z, v.w};
                if (SAN) {
                    const u32 rem = hl > 16u * kk ? hl - 16u * kk : 0u;
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
            };
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if ((u32)k + 1 >= nv) continue;  // wave-uniform
                step(q[k], (u32)k);
            }
            if (nv) step(tail, nv - 1);
            if (p < count && (!LEN || (hl != 0xFFFFu && hl >= min_len)) && st >= acc_lo) {  // (0xFFFF: an outlier beyond 256 bytes - k1_cdfa_outliers decides it)
                atomicOr(&s_bits[orig >> 5], 1u << (orig & 31));
            }
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
// The view's OUTLIERS (CorpusDev::vlong: the few haystacks beyond 256 bytes, which have no vectors in the view): decided from the canonical
// layout with the same class-composite automaton, the decision OR-ed into the bitmap the view kernel wrote and added to the tile's count.
// Launched behind k1_cdfa_view on the same stream; a list without outliers does not launch it.
// One WAVE per outlier, because a thread walking 600 bytes alone is a chain of 38 loads and 150 dependent lookups (12 us for the 71
// outliers of the Arabic-shaped list, half of what the view kernel takes for the other 285 516 haystacks): lane k takes vector k and
// runs it from EVERY start state (ns x G/4.. independent lookups - the transition function of its 16 bytes, ns bytes in LDS), then the
// wave composes the 64 functions in order: one dependent LDS byte per vector instead of a load and four lookups.
// ---------------------------------------------------------------------------------------------------
template <typename ET, bool SAN, int G>
__global__ __launch_bounds__(256) void k1_cdfa_outliers(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count, const u32* __restrict__ vlong, u32 n_long,
                                                        const u8* __restrict__ cdfa_g, u32 cdfa_bytes, u32 K, u32 KG, u32 ns, u32 min_len, u32 dead, u32 acc_lo,
                                                        u64* __restrict__ bitmap, u32* __restrict__ tile_counts) {
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 tab_bytes = (cdfa_bytes + 15u) & ~15u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    dfa_require_lds_base0(lds);
    for (u32 i = tid * 4; i < tab_bytes; i += blockDim.x * 4) *(u32*)(lds + i) = i < cdfa_bytes ? *(const u32*)(cdfa_g + i) : 0u;
    __syncthreads();
    u8* fn = lds + tab_bytes + (size_t)wave * ns * 64;  // fn[s * 64 + k]: where vector k of the current round takes state s
    const u32 deadv = dead * 0x01010101u;
    auto cls_of = [](u32 b) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)b; };
    auto comp_at = [](u32 a) -> u32 { return *(const __attribute__((address_space(3))) u8*)(uintptr_t)(256u + a); };
    const u32 wpw = blockDim.x >> 6;  // waves per workgroup: 4, or 1 when four waves' function tables do not fit
    for (u32 j = blockIdx.x * wpw + wave; j < n_long; j += gridDim.x * wpw) {
        const u64 gi = vlong[j];
        if (gi < first || gi >= first + count) continue;  // wave-uniform
        const u32 li = (u32)(gi - first);
        u64 hs;
        u32 hl;
        haystack_span(ends, gi, hs, hl);
        const uint4* vp = (const uint4*)(bytes + hs);
        const u32 nvec = (hl + 15u) >> 4;
        u32 st = 0;
        for (u32 base = 0; base < nvec; base += 64) {
            const u32 v = base + lane;
            if (v < nvec) {
                const uint4 q = vp[v];
                u32 w[4] = {q.x, q.y, q.z, q.w};
                if (SAN) {
                    const u32 rem = hl - 16 * v;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const u32 nvb = rem > 4u * t ? rem - 4u * t : 0u;
                        const u32 mask = nvb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvb)) - 1);
                        w[t] = (w[t] & mask) | (deadv & ~mask);
                    }
                }
                u32 off[G == 4 ? 4 : 8];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const u32 c0 = cls_of(w[t] & 0xFF), c1 = cls_of((w[t] >> 8) & 0xFF), c2 = cls_of((w[t] >> 16) & 0xFF), c3 = cls_of(w[t] >> 24);
                    if (G == 4) off[t] = c0 + K * (c1 + K * (c2 + K * c3));
                    else off[2 * t] = c0 + K * c1, off[2 * t + 1] = c2 + K * c3;
                }
                for (u32 s0 = 0; s0 < ns; s0++) {
                    u32 t = s0;
#pragma unroll
                    for (int q4 = 0; q4 < (G == 4 ? 4 : 8); q4++) t = comp_at(t * KG + off[q4]);
                    fn[s0 * 64 + lane] = (u8)t;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const u32 steps = min(64u, nvec - base);
            for (u32 k = 0; k < steps; k++) st = fn[st * 64 + k];  // (every lane walks the same chain: broadcast reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0 && hl >= min_len && st >= acc_lo) {
            atomicOr((unsigned long long*)&bitmap[li >> 6], 1ull << (li & 63));
            atomicAdd(&tile_counts[li / FZB_TILE], 1u);
        }
    }
}

// (Round 3 measured three more forms of the ragged filter on the canonical layout - software-pipelined burst, cooperative coalesced loads through
// a per-wave LDS window, LDS-staged + length-sorted - at 244-253 / 234-292 / 530-990 us against the burst form's 234; none is kept: the code
// is in the history, the numbers in profiles/r03_ragged_filter_variants.txt, the reading in DESIGN.md "The ragged filter".)

// ---------------------------------------------------------------------------------------------------
// Level-1 compaction in ONE kernel: every workgroup owns a
// contiguous run of tiles, obtains the number of survivors before its run by reducing the (small, L2-resident)
// per-tile count array itself - redundant across workgroups but far cheaper than a dependent scan launch -
// scans its own tiles' counts in LDS, and expands its bitmap words into the index-ordered survivor list.
// The last workgroup also publishes the total.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact1(const u64* __restrict__ bitmap, const u32* __restrict__ counts, u32 n_items_host, const u32* __restrict__ n_items_ptr,
                                                  const u32* __restrict__ src, u32* __restrict__ out_idx, u32* __restrict__ total_out,
                                                  u32* __restrict__ total_out2) {
    compact1_body(bitmap, counts, n_items_host, n_items_ptr, src, out_idx, total_out, total_out2, [](u32, u32) {}, [](u32) {});
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
    const int cus = std::max(1, grid / 8);  // (`grid` = 8 workgroups per CU = every wave slot)
    if (grid > (int)ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    if (mode == 1) {
        const size_t lds = (size_t)(rows + 1) * FZB_DFA_STRIDE + 16;  // table + the tile counter
        const bool shortc = c.max_len != 0 && c.max_len <= 32;  // every haystack fits the two pre-requested vectors
        if (shortc) {
#define FZB_K1D(ET) hipLaunchKernelGGL((k1_dfa<ET>), dim3(grid), dim3(256), lds, st, c.bytes, (const ET*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters, c.uniform_len)
            if (c.ends_u64) FZB_K1D(u64); else FZB_K1D(u32);
#undef FZB_K1D
            return;
        }
        // Ragged lists: one haystack per thread.  Three forms, best first: the class-composite automaton over the corpus' filter VIEW (6 resident
        // workgroups per CU: C4 shard 8 -> 190 us, 6 -> 187, 4 -> 194), the same automaton over the canonical layout (5 per CU: 8 -> 247 us,
        // 5 -> 232) when the corpus has no view or the range is not tile-aligned, and the byte automaton in its burst form (8 per CU) when the
        // composite table does not fit (host.hip: states x K^G <= 16 KB) or FZB_NO_CDFA=1 asks for it.
        const FzbKnobs& kn = fzb_knobs();
        const bool composite = cdfa && !kn.no_cdfa && (cdfa_G == 4 || cdfa_G == 2);
        u32 kg = 1;
        for (int i = 0; i < cdfa_G; i++) kg *= (u32)cdfa_K;
        if (composite && c.vbytes && !kn.no_filter_view && first % FZB_TILE == 0 && (first + count == c.n || count % FZB_TILE == 0) && c.view_nv != 0 && c.view_nv <= 16) {
            const size_t lds_v = ((cdfa_bytes + 15) & ~(size_t)15) + 16 + 128;
            const int g = std::max(1, std::min<int>(cus * 6, (int)ntiles));
            // (the lengths are not read when nothing needs them: acc >= 1 = the start state does not accept)
            const bool len_free = nul_safe && acc >= 1;
            // short lists (fewer tiles than two per CU, the common form only): 1024-thread workgroups, one group per wave.  Filter / step, us:
            // 100 k paths 14.5 -> 8.3 / 39.8 -> 34.0, 300 k 16.5 -> 13.7 / 42.9 -> 40.6, 0.5 M 8..128-byte items 18.1 -> 12.7 / 55.3 -> 49.1; from
            // 1.4 M items (1 374 tiles) on the 256-thread form's six workgroups per CU win (34.5 against 38.3)
            const bool wide = nul_safe && len_free && ntiles < (u32)cus * 2u;
#define FZB_K1VA const_cast<const u8*>(c.vbytes), c.vgofs, c.vgnv, c.vlen, c.vperm, first, count, cdfa, cdfa_bytes, (u32)cdfa_K, kg, min_len, dead, acc, bitmap, tile_counts, reset_counters
#define FZB_K1V(SAN, G, NV, LEN) hipLaunchKernelGGL((k1_cdfa_view<SAN, G, NV, LEN>), dim3(g), dim3(256), lds_v, st, FZB_K1VA)
#define FZB_K1W(G, NV) hipLaunchKernelGGL((k1_cdfa_view<false, G, NV, false, 1024>), dim3(g), dim3(1024), lds_v, st, FZB_K1VA)
#define FZB_K1V_S(SAN, G, NV) do { if (!SAN && wide) FZB_K1W(G, NV); else if (!SAN && len_free) FZB_K1V(SAN, G, NV, false); else FZB_K1V(SAN, G, NV, true); } while (0)
#define FZB_K1V_NV(SAN, G) do { if (c.view_nv <= 8) FZB_K1V_S(SAN, G, 8); else FZB_K1V_S(SAN, G, 16); } while (0)
#define FZB_K1V_G(SAN) do { if (cdfa_G == 4) FZB_K1V_NV(SAN, 4); else FZB_K1V_NV(SAN, 2); } while (0)
            if (nul_safe) FZB_K1V_G(false); else FZB_K1V_G(true);
#undef FZB_K1V_G
#undef FZB_K1V_NV
#undef FZB_K1V_S
#undef FZB_K1W
#undef FZB_K1V
#undef FZB_K1VA
            if (c.n_long) {  // the haystacks beyond 256 bytes: decided from the canonical layout, OR-ed into the view kernel's bitmap
                const u32 ns_o = (cdfa_bytes - 256u) / kg;  // the automaton's states (the table is padded to 16 bytes: at most a phantom state more)
                const u32 wpw = ((cdfa_bytes + 15) & ~(size_t)15) + (size_t)4 * 64 * ns_o + 16 <= 60 * 1024 ? 4u : 1u;
                const size_t lds_o = ((cdfa_bytes + 15) & ~(size_t)15) + (size_t)wpw * 64 * ns_o + 16;
                const int go = (int)std::max<u32>(1u, std::min<u32>((c.n_long + wpw - 1) / wpw, 4096u));
#define FZB_K1O(ET, SAN, G) hipLaunchKernelGGL((k1_cdfa_outliers<ET, SAN, G>), dim3(go), dim3(64 * wpw), lds_o, st, c.bytes, (const ET*)c.ends, first, count, c.vlong, c.n_long, cdfa, cdfa_bytes, (u32)cdfa_K, kg, ns_o, min_len, dead, acc, bitmap, tile_counts)
#define FZB_K1O_G(ET, SAN) do { if (cdfa_G == 4) FZB_K1O(ET, SAN, 4); else FZB_K1O(ET, SAN, 2); } while (0)
                if (c.ends_u64) { if (nul_safe) FZB_K1O_G(u64, false); else FZB_K1O_G(u64, true); }
                else            { if (nul_safe) FZB_K1O_G(u32, false); else FZB_K1O_G(u32, true); }
#undef FZB_K1O_G
#undef FZB_K1O
            }
            return;
        }
        if (composite) {
            const size_t lds_c = ((cdfa_bytes + 15) & ~(size_t)15) + 16;
            const int g = std::max(1, std::min<int>(cus * 5, (int)ntiles));
#define FZB_K1CD(ET, SAN, G) hipLaunchKernelGGL((k1_cdfa_ragged<ET, SAN, G>), dim3(g), dim3(256), lds_c, st, c.bytes, (const ET*)c.ends, first, count, cdfa, cdfa_bytes, (u32)cdfa_K, kg, min_len, dead, acc, bitmap, tile_counts, reset_counters)
#define FZB_K1CD_G(ET, SAN) do { if (cdfa_G == 4) FZB_K1CD(ET, SAN, 4); else FZB_K1CD(ET, SAN, 2); } while (0)
            if (c.ends_u64) { if (nul_safe) FZB_K1CD_G(u64, false); else FZB_K1CD_G(u64, true); }
            else            { if (nul_safe) FZB_K1CD_G(u32, false); else FZB_K1CD_G(u32, true); }
#undef FZB_K1CD_G
#undef FZB_K1CD
            return;
        }
#define FZB_K1B(ET, SAN) hipLaunchKernelGGL((k1_dfa_ragged_burst<ET, SAN>), dim3(grid), dim3(256), lds, st, c.bytes, (const ET*)c.ends, first, count, dfa, rows, min_len, dead, acc, bitmap, tile_counts, reset_counters)
        if (c.ends_u64) { if (nul_safe) FZB_K1B(u64, false); else FZB_K1B(u64, true); }
        else            { if (nul_safe) FZB_K1B(u32, false); else FZB_K1B(u32, true); }
#undef FZB_K1B
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

void fzb_launch_compact1(const u64* bitmap, const u32* counts, u32 n_items, const u32* n_items_ptr, const u32* src, u32* out_idx, u32* total_out, int grid, hipStream_t st,
                         u32* total_out2) {
    hipLaunchKernelGGL(k_compact1, dim3(grid), dim3(256), 0, st, bitmap, counts, n_items, n_items_ptr, src, out_idx, total_out, total_out2);
}

__global__ void k_init_counters(u32* __restrict__ counters, u32 n0) {
    if (threadIdx.x < 16) counters[threadIdx.x] = threadIdx.x == 0 ? n0 : 0u;
}
void fzb_launch_init_counters(u32* counters, u32 n0, hipStream_t st) { hipLaunchKernelGGL(k_init_counters, dim3(1), dim3(64), 0, st, counters, n0); }

void fzb_launch_compact2(const u64* bitmap, const u32* counts, const u32* n_items_ptr, const u32* in_idx, const u32* in_win, u32* out_idx, u32* out_win, u32* total_out,
                         int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_compact2, dim3(grid), dim3(256), 0, st, bitmap, counts, n_items_ptr, in_idx, in_win, out_idx, out_win, total_out);
}
