// gfx950 kernels for the literal matching modes: exact / prefix / suffix / substring.
//
// Reference: src/literal/algo.rs.  The needle must occur as one contiguous run (`matches_at`, :157-176: per byte
// against (original, case-flipped) on the ASCII path, per scalar against the whole original / flipped scalar on the
// unicode path); the score is the bonuses of a gap-free alignment (`score_scalar` :180-200, `score_at` :204-225);
// substring takes the best-scoring occurrence, the earliest on ties (`find_substring` :239-312, whose two-seed-byte
// SIMD scan is only an accelerator: the reference asserts that all its backends agree, src/literal/backend.rs:103-200).
// `max_typos` is ignored (src/literal/mod.rs:7).
//
// Two passes like the fuzzy pipeline: an accept pass over every haystack (streaming, one bit each) and a scoring
// pass over the survivors that writes the Match records at their rank.
#include "kernels_common.h"

#define LIT_EXACT 1
#define LIT_PREFIX 2
#define LIT_SUFFIX 3
#define LIT_SUBSTRING 4

template <typename ND>
__device__ __forceinline__ bool lit_matches_at(const ND& nd, const u8* __restrict__ h, u32 pos) {
    if (nd.unicode) {
        u32 k = pos;
        for (int r = 0; r < nd.rows; r++) {
            const int len = nd.ulen[r];
            bool eq_c = true, eq_f = true;
            for (int b = 0; b < len; b++) {
                const u8 x = h[k + b];
                eq_c = eq_c && x == nd.uc[r][b];
                eq_f = eq_f && x == nd.uf[r][b];
            }
            if (!eq_c && !eq_f) return false;
            k += len;
        }
        return true;
    }
    for (int k = 0; k < nd.nbytes; k++) {
        const u8 x = h[pos + k];
        if (x != nd.c[k] && x != nd.f[k]) return false;
    }
    return true;
}

__device__ __forceinline__ bool lit_is_delim(u8 b) { return b <= 127 && !((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z') || (b >= '0' && b <= '9')); }

template <typename ND>
__device__ __forceinline__ u32 lit_score_scalar(const ND& nd, const u8* __restrict__ h, u32 start, bool exact_case) {
    u32 s = nd.match_score;
    if (exact_case) s += nd.matching_case;
    if (start == 0) {
        s += nd.prefix;
    } else {
        const u8 b = h[start], prev = h[start - 1];
        if (b >= 'A' && b <= 'Z' && prev >= 'a' && prev <= 'z') s += nd.capitalization;
        if (lit_is_delim(prev) && !lit_is_delim(b)) s += nd.delimiter;
    }
    return s;
}

template <typename ND>
__device__ __forceinline__ u32 lit_score_at(const ND& nd, const u8* __restrict__ h, u32 L, u32 pos) {
    u32 score = 0;
    if (nd.unicode) {
        u32 start = pos;
        for (int r = 0; r < nd.rows; r++) {
            const int len = nd.ulen[r];
            bool eq_c = true;
            for (int b = 0; b < len; b++) eq_c = eq_c && h[start + b] == nd.uc[r][b];
            score += lit_score_scalar(nd, h, start, eq_c);
            start += len;
        }
    } else {
        for (int k = 0; k < nd.nbytes; k++) score += lit_score_scalar(nd, h, pos + k, h[pos + k] == nd.c[k]);
    }
    if (pos == 0 && (u32)nd.nbytes == L) score += nd.exact_bonus;
    return score & 0xFFFFu;  // (u16 arithmetic; the overflow guard keeps it below 2^16)
}

// candidate starts in one dword: bytes equal to either form of the needle's first byte -> 4-bit mask
__device__ __forceinline__ u32 lit_first_byte_hits(u32 w, u32 a, u32 b) {
    auto zb = [](u32 x) {
        u32 y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
        y = ~y & 0x80808080u;
        return ((y >> 7) * 0x00204081u >> 21) & 0xFu;
    };
    return zb(w ^ a) | zb(w ^ b);
}

// Accept decision only.  Substring: first-byte candidates from the aligned 16-byte vectors, verified by byte reads
// (which hit the lines the vector load just brought in).
template <typename ND>
__device__ __forceinline__ bool lit_accepts(const ND& nd, int mode, const u8* __restrict__ h, u32 L) {
    const u32 nl = (u32)nd.nbytes;
    if (L < nl) return false;
    if (mode == LIT_EXACT) return L == nl && lit_matches_at(nd, h, 0);
    if (mode == LIT_PREFIX) return lit_matches_at(nd, h, 0);
    if (mode == LIT_SUFFIX) return lit_matches_at(nd, h, L - nl);
    const u32 a = (nd.unicode ? nd.uc[0][0] : nd.c[0]) * 0x01010101u, b = (nd.unicode ? nd.uf[0][0] : nd.f[0]) * 0x01010101u;
    const u32 last_start = L - nl;  // inclusive
    const uint4* vp = (const uint4*)h;
    const u32 nvec = (last_start >> 4) + 1;
    for (u32 v = 0; v < nvec; v++) {
        const uint4 q = vp[v];
        u32 hits = lit_first_byte_hits(q.x, a, b) | (lit_first_byte_hits(q.y, a, b) << 4) | (lit_first_byte_hits(q.z, a, b) << 8) | (lit_first_byte_hits(q.w, a, b) << 12);
        while (hits) {
            const u32 pos = 16 * v + (u32)__builtin_ctz(hits);
            hits &= hits - 1;
            if (pos > last_start) break;
            if (lit_matches_at(nd, h, pos)) return true;
        }
    }
    return false;
}

// position + score of the match the reference reports (`find`, algo.rs:232-253)
template <typename ND>
__device__ __forceinline__ bool lit_find(const ND& nd, int mode, const u8* __restrict__ h, u32 L, u32& pos_out, u32& score_out) {
    const u32 nl = (u32)nd.nbytes;
    if (L < nl) return false;
    if (mode == LIT_EXACT || mode == LIT_PREFIX || mode == LIT_SUFFIX) {
        const u32 pos = mode == LIT_SUFFIX ? L - nl : 0u;
        if (mode == LIT_EXACT && L != nl) return false;
        if (!lit_matches_at(nd, h, pos)) return false;
        pos_out = pos;
        score_out = lit_score_at(nd, h, L, pos);
        return true;
    }
    bool found = false;
    for (u32 pos = 0; pos + nl <= L; pos++) {
        if (!lit_matches_at(nd, h, pos)) continue;
        const u32 sc = lit_score_at(nd, h, L, pos);
        if (!found || sc > score_out) {
            found = true;
            pos_out = pos;
            score_out = sc;
        }
    }
    return found;
}

// pass 1: items == nullptr: haystacks [first, first + count_host); else the listed ones (device-side count)
template <typename ET, typename ND = NeedleDev>
__global__ __launch_bounds__(256) void k_literal_filter(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count_host, const u32* __restrict__ items,
                                                        const u32* __restrict__ n_items_ptr, const ND nd, int mode, u64* __restrict__ bitmap,
                                                        u32* __restrict__ tile_counts) {
    __shared__ u32 s_cnt;
    const u32 count = items ? *n_items_ptr : count_host;
    const u32 ntiles = (count + FZB_TILE - 1) / FZB_TILE;
    const int tid = threadIdx.x;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int p = 0; p < FZB_TILE / 256; p++) {
            const u32 j = tile * FZB_TILE + p * 256 + tid;
            bool keep = false;
            if (j < count) {
                u64 s;
                u32 L;
                haystack_span(ends, first + (items ? items[j] : j), s, L);
                keep = lit_accepts(nd, mode, bytes + s, L);
            }
            const u64 b = __ballot(keep);
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

// pass 2: one thread per survivor (items = local haystack indices), record j at out[j].  With tpos != nullptr also the matched
// byte positions (match_list_indices_impl, algo.rs:129-155: the whole needle run, reversed) at tpos[j * tstride ..], tnpos[j] of them.
template <typename ET, typename ND = NeedleDev>
__global__ __launch_bounds__(256) void k_literal_score(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 index_offset, const u32* __restrict__ items,
                                                       const u32* __restrict__ n_items_ptr, const ND nd, int mode, fzb_match_rec* __restrict__ out, u32 capacity,
                                                       u32* __restrict__ dev_count, u32* __restrict__ tpos, u32* __restrict__ tnpos, u32 tstride) {
    const u32 M = *n_items_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = M < capacity ? M : capacity; dev_count[1] = M; }
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < M && j < capacity; j += gridDim.x * blockDim.x) {
        const u32 li = items[j];
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        u32 pos = 0, score = 0;
        lit_find(nd, mode, bytes + s, L, pos, score);  // always found: li passed the accept pass
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = (pos == 0 && (u32)nd.nbytes == L) ? 1 : 0;  // algo.rs:113
        rec.valid = 0;
        out[j] = rec;
        if (tpos) {
            const u32 n = (u32)nd.nbytes;
            for (u32 k = 0; k < n && k < tstride; k++) tpos[(size_t)j * tstride + k] = pos + (n - 1 - k);
            tnpos[j] = n < tstride ? n : tstride;
        }
    }
}

void fzb_launch_literal_filter(const CorpusDev& c, u64 first, u32 count, const u32* items, const u32* n_items_ptr, const NeedleDev& nd, int mode, u64* bitmap, u32* tile_counts,
                               int grid, hipStream_t st) {
    if (c.ends_u64) hipLaunchKernelGGL((k_literal_filter<u64>), dim3(grid), dim3(256), 0, st, c.bytes, (const u64*)c.ends, first, count, items, n_items_ptr, nd, mode, bitmap, tile_counts);
    else hipLaunchKernelGGL((k_literal_filter<u32>), dim3(grid), dim3(256), 0, st, c.bytes, (const u32*)c.ends, first, count, items, n_items_ptr, nd, mode, bitmap, tile_counts);
}
void fzb_launch_literal_score(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* n_items_ptr, const NeedleDev& nd, int mode, fzb_match_rec* out,
                              u32 capacity, u32* dev_count, u32* tpos, u32* tnpos, u32 tstride, int grid, hipStream_t st) {
    if (c.ends_u64) hipLaunchKernelGGL((k_literal_score<u64>), dim3(grid), dim3(256), 0, st, c.bytes, (const u64*)c.ends, first, index_offset, items, n_items_ptr, nd, mode, out, capacity, dev_count, tpos, tnpos, tstride);
    else hipLaunchKernelGGL((k_literal_score<u32>), dim3(grid), dim3(256), 0, st, c.bytes, (const u32*)c.ends, first, index_offset, items, n_items_ptr, nd, mode, out, capacity, dev_count, tpos, tnpos, tstride);
}

// long needles (NeedleLongDev: needle arrays in device memory): the same two kernels
void fzb_launch_literal_filter_long(const CorpusDev& c, u64 first, u32 count, const u32* items, const u32* n_items_ptr, const NeedleLongDev& nd, int mode, u64* bitmap,
                                    u32* tile_counts, int grid, hipStream_t st) {
    if (c.ends_u64) hipLaunchKernelGGL((k_literal_filter<u64, NeedleLongDev>), dim3(grid), dim3(256), 0, st, c.bytes, (const u64*)c.ends, first, count, items, n_items_ptr, nd, mode, bitmap, tile_counts);
    else hipLaunchKernelGGL((k_literal_filter<u32, NeedleLongDev>), dim3(grid), dim3(256), 0, st, c.bytes, (const u32*)c.ends, first, count, items, n_items_ptr, nd, mode, bitmap, tile_counts);
}
void fzb_launch_literal_score_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* n_items_ptr, const NeedleLongDev& nd, int mode, fzb_match_rec* out,
                                   u32 capacity, u32* dev_count, u32* tpos, u32* tnpos, u32 tstride, int grid, hipStream_t st) {
    if (c.ends_u64) hipLaunchKernelGGL((k_literal_score<u64, NeedleLongDev>), dim3(grid), dim3(256), 0, st, c.bytes, (const u64*)c.ends, first, index_offset, items, n_items_ptr, nd, mode, out, capacity, dev_count, tpos, tnpos, tstride);
    else hipLaunchKernelGGL((k_literal_score<u32, NeedleLongDev>), dim3(grid), dim3(256), 0, st, c.bytes, (const u32*)c.ends, first, index_offset, items, n_items_ptr, nd, mode, out, capacity, dev_count, tpos, tnpos, tstride);
}
