// Level-1 compaction, the body shared by k_compact1 (kernels_filter.hip) and the fused compaction + classification kernel of ragged ASCII lists
// (k_compact1_classify, kernels_dp.hip): see k_compact1 for the scheme.  `after_batch(first_position, survivors)` is called by the WHOLE workgroup
// (workgroup-uniform arguments) once the survivor indices [first_position, first_position + survivors) of a batch of this workgroup's tiles are in
// out_idx and visible to the workgroup; `at_end(total)` by thread 0 of the last workgroup with the list's total.
#pragma once
#include "fzb_internal.h"

template <class AfterBatch, class AtEnd>
__device__ __forceinline__ void compact1_body(const u64* __restrict__ bitmap, const u32* __restrict__ counts, u32 n_items_host, const u32* __restrict__ n_items_ptr,
                                              const u32* __restrict__ src, u32* __restrict__ out_idx, u32* __restrict__ total_out, u32* __restrict__ total_out2,
                                              AfterBatch after_batch, AtEnd at_end) {

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
        after_batch(base, batch_total);
        base += batch_total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {  // every earlier workgroup's tiles precede this one's
        *total_out = base;
        if (total_out2) *total_out2 = base;  // (a second counter that starts as the same number: the unicode path's "kept by the exact prefilter")
        at_end(base);
    }
}
