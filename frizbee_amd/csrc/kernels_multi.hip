// gfx950 kernels for the multi-pattern composition.
//
// Reference: src/matcher/multi.rs:84-152 `match_list_multi_into` - the first non-negated pattern is matched against
// every haystack, every further pattern only against the survivors ("candidates"), and its hits either
//   * replace the candidates, with score = hit.score.saturating_add(candidate.score) and exact |= candidate.exact, or
//   * (negated pattern) are removed from the candidates.
// All lists are in haystack-index order, on both sides, so "the candidate this hit belongs to" is a binary search on
// `index` instead of the reference's position bookkeeping; list lengths stay in device memory throughout.
#include "kernels_common.h"

// candidates -> local haystack indices for the next pattern's pipeline
__global__ __launch_bounds__(256) void k_records_to_items(const fzb_match_rec* __restrict__ cand, const u32* __restrict__ n_ptr, u32 index_offset,
                                                          u32* __restrict__ items) {
    const u32 n = *n_ptr;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) items[j] = cand[j].index - index_offset;
}

// every haystack is a candidate with score 0 (all patterns negated, multi.rs:99-103)
__global__ __launch_bounds__(256) void k_identity_records(fzb_match_rec* __restrict__ out, u32 n, u32 index_offset, u32* __restrict__ count_out) {
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        fzb_match_rec r;
        r.index = index_offset + j;
        r.score = 0;
        r.exact = 0;
        r.valid = 0;
        out[j] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { count_out[0] = n; count_out[1] = n; }
}

// position of the record with `index` in the index-ascending list, or n if absent
__device__ __forceinline__ u32 find_index(const fzb_match_rec* __restrict__ list, u32 n, u32 index) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (list[mid].index < index) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && list[lo].index == index) ? lo : n;
}

// non-negated pattern: hits become the candidates, carrying the accumulated score / exact flag (multi.rs:133-148)
__global__ __launch_bounds__(256) void k_join_add(fzb_match_rec* __restrict__ hits, const u32* __restrict__ n_hits_ptr, const fzb_match_rec* __restrict__ cand,
                                                  const u32* __restrict__ n_cand_ptr) {
    const u32 nh = *n_hits_ptr, nc = *n_cand_ptr;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += gridDim.x * blockDim.x) {
        fzb_match_rec h = hits[j];
        const u32 p = find_index(cand, nc, h.index);
        if (p < nc) {  // always: the hits are a subset of the candidates
            const fzb_match_rec c = cand[p];
            const u32 s = (u32)h.score + (u32)c.score;
            h.score = (u16)(s > 0xFFFFu ? 0xFFFFu : s);
            h.exact = h.exact | c.exact;
            hits[j] = h;
        }
    }
}

// negated pattern: bit j = candidate j is NOT among the hits (multi.rs:121-131); bitmap + per-1024 counts as everywhere
__global__ __launch_bounds__(256) void k_flag_absent(const fzb_match_rec* __restrict__ cand, const u32* __restrict__ n_cand_ptr, const fzb_match_rec* __restrict__ hits,
                                                     const u32* __restrict__ n_hits_ptr, u64* __restrict__ bitmap, u32* __restrict__ tile_counts) {
    __shared__ u32 s_cnt;
    const u32 nc = *n_cand_ptr, nh = *n_hits_ptr;
    const u32 ntiles = (nc + FZB_TILE - 1) / FZB_TILE;
    const int tid = threadIdx.x;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
        for (int p = 0; p < FZB_TILE / 256; p++) {
            const u32 j = tile * FZB_TILE + p * 256 + tid;
            const bool keep = j < nc && find_index(hits, nh, cand[j].index) == nh;
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

// order-preserving compaction of records by the bitmap (same scheme as k_compact1: every workgroup derives the number of
// kept records before its run of tiles from the per-tile counts)
__global__ __launch_bounds__(256) void k_compact_records(const u64* __restrict__ bitmap, const u32* __restrict__ counts, const u32* __restrict__ n_ptr,
                                                         const fzb_match_rec* __restrict__ in, fzb_match_rec* __restrict__ out, u32* __restrict__ total_out) {
    __shared__ u32 red[4];
    __shared__ u32 pre[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 n = *n_ptr;
    const u32 ntiles = (n + FZB_TILE - 1) / FZB_TILE;
    const u32 T = (ntiles + gridDim.x - 1) / gridDim.x;
    const u32 t0 = min(blockIdx.x * T, ntiles), t1 = min(t0 + T, ntiles);
    u32 part = 0;
    for (u32 i = tid; i < t0; i += 256) part += counts[i];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    u32 base = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    for (u32 tb = t0; tb < t1; tb += 256) {
        const u32 nt = min(256u, t1 - tb);
        const u32 c = (u32)tid < nt ? counts[tb + tid] : 0u;
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
        const u32 w0 = tb * (FZB_TILE / 64), w1 = (tb + nt) * (FZB_TILE / 64);
        const u32 nwords = (n + 63) / 64;
        for (u32 w = w0 + tid; w < w1 && w < nwords; w += 256) {
            u64 bits = bitmap[w];
            if (!bits) continue;
            const u32 tile = w / (FZB_TILE / 64);
            u32 pos = pre[tile - tb];
            for (u32 k = tile * (FZB_TILE / 64); k < w; k++) pos += __popcll(bitmap[k]);
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                out[pos++] = in[w * 64 + b];
            }
        }
        base += batch_total;
        __syncthreads();
    }
    // (n == 0: no workgroup has tiles; the last one still publishes base = 0)
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *total_out = base;
}

__global__ __launch_bounds__(256) void k_copy_records(const fzb_match_rec* __restrict__ in, const u32* __restrict__ n_ptr, fzb_match_rec* __restrict__ out, u32 capacity,
                                                      u32* __restrict__ count_out) {
    const u32 total = *n_ptr;
    const u32 n = min(total, capacity);
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) out[j] = in[j];
    if (blockIdx.x == 0 && threadIdx.x == 0) { count_out[0] = n; count_out[1] = total; }  // [1] = the untruncated total
}

void fzb_launch_records_to_items(const fzb_match_rec* cand, const u32* n_ptr, u32 index_offset, u32* items, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_records_to_items, dim3(grid), dim3(256), 0, st, cand, n_ptr, index_offset, items);
}
void fzb_launch_identity_records(fzb_match_rec* out, u32 n, u32 index_offset, u32* count_out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_identity_records, dim3(grid), dim3(256), 0, st, out, n, index_offset, count_out);
}
void fzb_launch_join_add(fzb_match_rec* hits, const u32* n_hits_ptr, const fzb_match_rec* cand, const u32* n_cand_ptr, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_join_add, dim3(grid), dim3(256), 0, st, hits, n_hits_ptr, cand, n_cand_ptr);
}
void fzb_launch_remove_hits(const fzb_match_rec* cand, const u32* n_cand_ptr, const fzb_match_rec* hits, const u32* n_hits_ptr, u64* bitmap, u32* tile_counts,
                            fzb_match_rec* out, u32* total_out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_flag_absent, dim3(grid), dim3(256), 0, st, cand, n_cand_ptr, hits, n_hits_ptr, bitmap, tile_counts);
    hipLaunchKernelGGL(k_compact_records, dim3(grid), dim3(256), 0, st, bitmap, tile_counts, n_cand_ptr, cand, out, total_out);
}
void fzb_launch_copy_records(const fzb_match_rec* in, const u32* n_ptr, fzb_match_rec* out, u32 capacity, u32* count_out, int grid, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_records, dim3(grid), dim3(256), 0, st, in, n_ptr, out, capacity, count_out);
}
