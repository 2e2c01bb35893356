// gfx950 kernels, stage 3 (ordering): `radix_sort_matches` on the device.
//
// Reference: src/sort.rs:6-40 - a STABLE least-significant-digit radix sort of the Match records by `score`,
// descending, two passes of 8 bits; ties keep their input order (the input is index-ascending, or index-descending
// after the reverse that `match_list` applies for the *Desc strategies, src/matcher/mod.rs:215-221).
// Same algorithm here, each pass as histogram -> exclusive scan -> stable scatter:
//   * a tile = 2048 consecutive records, processed in order by one workgroup, 256 at a time;
//   * digit' = 255 - digit so ascending bucket order is descending score order;
//   * inside a wave the rank among equal digits is found with 8 ballots (one per digit bit: lanes holding the same
//     digit = AND of the matching ballot masks), waves of a 256-record slab are ordered through a per-wave histogram
//     in LDS, slabs through a running per-digit counter - so equal keys never overtake each other.
// The record count lives in device memory (no host round trip); grids are sized for the capacity.
#include "kernels_common.h"

#define SORT_TILE 2048

__device__ __forceinline__ u32 sort_digit(const fzb_match_rec& r, int shift) { return 255u - ((r.score >> shift) & 0xFFu); }

// hist[d * ntiles_cap + tile] = number of records of tile with digit' d.  The per-digit TOTALS are accumulated on the way (one atomic per
// digit and tile into dtot[phase]) so that the scan kernel does not have to reduce the whole histogram in every one of its workgroups
// first (12.2 -> see DESIGN.md): dtot = two sets of 256 counters + a phase word behind them; a pass accumulates into set `phase`, clears the
// other one (the next pass's), and the scatter kernel - which runs when nobody reads the counters any more - flips the phase.
__global__ __launch_bounds__(256) void k_sort_hist(const fzb_match_rec* __restrict__ in, const u32* __restrict__ n_ptr, int shift, u32* __restrict__ hist,
                                                   u32 ntiles_cap, u32* __restrict__ dtot) {
    __shared__ u32 h[256];
    const u32 n = *n_ptr;
    const u32 ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    const u32 phase = dtot[512] & 1u;
    u32* const mine = dtot + 256 * phase;
    if (blockIdx.x == 0) dtot[256 * (phase ^ 1u) + threadIdx.x] = 0;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        h[threadIdx.x] = 0;
        __syncthreads();
        const u32 lo = tile * SORT_TILE, hi = min(lo + SORT_TILE, n);
        for (u32 i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[sort_digit(in[i], shift)], 1u);
        __syncthreads();
        const u32 c = h[threadIdx.x];
        hist[threadIdx.x * ntiles_cap + tile] = c;
        if (c) atomicAdd(&mine[threadIdx.x], c);
        __syncthreads();
    }
}

// exclusive scan of the digit-major histogram (256 rows of `ntiles` live entries, row stride ntiles_cap) into `offs` (same layout)
__global__ __launch_bounds__(1024) void k_sort_scan(const u32* __restrict__ hist, u32* __restrict__ offs, const u32* __restrict__ n_ptr, u32 ntiles_cap,
                                                    const u32* __restrict__ dtot) {
    // Exclusive scan of the digit-major tile histogram, element (digit d, tile t) at hist[d * ntiles_cap + t], in the order
    // (d, t) lexicographic.  16 workgroups of 16 waves.  Every workgroup (1) reads the 256 per-digit totals the histogram kernel
    // accumulated (rounds 2-3 had every workgroup reduce the whole histogram itself: 250 KB of L2 reads and 96 shuffles per wave before
    // anything else could start) and (2) scans them; then (3) workgroup b scans the tiles of ITS 16 digits, one digit per wave, from the
    // digit's base.  Not in place: another workgroup may still be reading the rows this one scans.
    __shared__ u32 dtot_s[256];
    u32* const dtotv = dtot_s;
    const u32 n = *n_ptr;
    const u32 ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    if (threadIdx.x < 256) dtotv[threadIdx.x] = dtot[256 * (dtot[512] & 1u) + threadIdx.x];  // (1)
    __syncthreads();
    if (wave == 0) {  // (2) exclusive scan of the 256 digit totals: 4 per lane
        u32 v[4], s4 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = dtotv[lane * 4 + k]; s4 += v[k]; }
        u32 incl = s4;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        u32 run = incl - s4;
#pragma unroll
        for (int k = 0; k < 4; k++) { dtotv[lane * 4 + k] = run; run += v[k]; }
    }
    __syncthreads();
    {  // (3) this workgroup's digits, one per wave: a wave scan over the digit's tiles from its base
        const u32 d = blockIdx.x * 16 + wave;
        u32 carry = dtotv[d];
        for (u32 t0 = 0; t0 < ntiles; t0 += 64) {  // uniform trip count: every lane takes part in the shuffles
            const u32 t = t0 + lane;
            const u32 v = t < ntiles ? hist[(size_t)d * ntiles_cap + t] : 0u;
            u32 incl = v;
            for (int off = 1; off < 64; off <<= 1) {
                const u32 x = __shfl_up(incl, off);
                if (lane >= off) incl += x;
            }
            if (t < ntiles) offs[(size_t)d * ntiles_cap + t] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
}

__global__ __launch_bounds__(256) void k_sort_scatter(const fzb_match_rec* __restrict__ in, fzb_match_rec* __restrict__ out, const u32* __restrict__ n_ptr,
                                                      int shift, const u32* __restrict__ offs, u32 ntiles_cap, u32* __restrict__ dtot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) dtot[512] ^= 1u;  // the digit totals of this pass have been read: the next pass takes the other set
    __shared__ u32 wave_hist[4][256];
    __shared__ u32 run[256];
    const u32 n = *n_ptr;
    const u32 ntiles = (n + SORT_TILE - 1) / SORT_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        run[tid] = offs[tid * ntiles_cap + tile];  // where this tile's records with digit' = tid start
        const u32 lo = tile * SORT_TILE, hi = min(lo + SORT_TILE, n);
        for (u32 base = lo; base < hi; base += 256) {
#pragma unroll
            for (int w = 0; w < 4; w++) wave_hist[w][tid] = 0;
            __syncthreads();
            const u32 i = base + tid;
            const bool valid = i < hi;
            fzb_match_rec r;
            u32 d = 0;
            if (valid) {
                r = in[i];
                d = sort_digit(r, shift);
            }
            // lanes of this wave holding the same digit (and valid)
            u64 peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const u64 bal = __ballot((d >> b) & 1);
                peers &= ((d >> b) & 1) ? bal : ~bal;
            }
            const u32 rank = __popcll(peers & (((u64)1 << lane) - 1));
            if (valid && rank == 0) wave_hist[wave][d] = __popcll(peers);
            __syncthreads();
            // digit tid: order the four waves, advance the running offset
            const u32 c0 = wave_hist[0][tid], c1 = wave_hist[1][tid], c2 = wave_hist[2][tid], c3 = wave_hist[3][tid];
            const u32 b0 = run[tid];
            __syncthreads();
            wave_hist[0][tid] = b0;
            wave_hist[1][tid] = b0 + c0;
            wave_hist[2][tid] = b0 + c0 + c1;
            wave_hist[3][tid] = b0 + c0 + c1 + c2;
            run[tid] = b0 + c0 + c1 + c2 + c3;
            __syncthreads();
            if (valid) out[wave_hist[wave][d] + rank] = r;
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void k_reverse(fzb_match_rec* __restrict__ a, const u32* __restrict__ n_ptr) {
    const u32 n = *n_ptr;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += gridDim.x * blockDim.x) {
        const fzb_match_rec x = a[i], y = a[n - 1 - i];
        a[i] = y;
        a[n - 1 - i] = x;
    }
}

// Per-shard runs -> one list (fzb_merge_shard_runs: what frizbee_amd.distributed runs on the root after the RCCL gather; more than
// FZB_MAX_RUNS runs go through it in batches, `base_in` carrying the running total).  Shard g's run is index-ordered and the shards are
// contiguous ascending index ranges, so the concatenation in shard order IS the index-ordered record list of the whole query: the
// reverse / stable radix sort that follows reproduces `match_list`'s order exactly (src/matcher/mod.rs:215-221) - the result of
// `match_list_parallel`'s per-run sort + k-way merge (src/matcher/parallel.rs:66-87) without a host heap.
// Every workgroup scans the (<= 64) run lengths itself; a record finds its run by binary search in LDS.
__global__ __launch_bounds__(256) void k_concat_runs(RunSet rs, const u32* __restrict__ base_in, u32* __restrict__ total_out, fzb_match_rec* __restrict__ out, u32 capacity,
                                                     u32* __restrict__ cut_flag) {
    __shared__ u32 pre[FZB_MAX_RUNS + 1];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const u32 c = tid < rs.n ? min(rs.count[tid][0], rs.cap[tid]) : 0u;
        // a run whose producer found more matches than its buffer holds (fzb_match_list_device: count[1] > capacity) is reported, not merged
        const bool cut = tid < rs.n && rs.count[tid][1] > rs.cap[tid];
        // (the first batch - no base - ASSIGNS the flag, later ones only raise it: nothing has to clear the word before the launch)
        const bool any_cut = __ballot(cut) != 0;
        if (tid == 0 && blockIdx.x == 0 && (any_cut || !base_in)) *cut_flag = any_cut ? 1u : 0u;
        u32 incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 t = __shfl_up(incl, off);
            if (tid >= off) incl += t;
        }
        pre[tid + 1] = incl;
        if (tid == 0) pre[0] = 0;
    }
    __syncthreads();
    const u32 base = base_in ? *base_in : 0u;
    const u32 total = pre[rs.n];
    for (u32 i = blockIdx.x * 256u + tid; i < total; i += gridDim.x * 256u) {
        int lo = 0, hi = rs.n - 1;  // last run whose prefix is <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        if (base + i < capacity) out[base + i] = rs.run[lo][i - pre[lo]];
    }
    if (blockIdx.x == 0 && tid == 0) {
        total_out[0] = min(base + total, capacity);
        total_out[1] = base + total;
    }
}

void fzb_launch_concat_runs(const RunSet& rs, const u32* base_in, u32* total_out, fzb_match_rec* out, u32 capacity, int grid, u32* cut_flag, hipStream_t st) {
    hipLaunchKernelGGL(k_concat_runs, dim3(grid), dim3(256), 0, st, rs, base_in, total_out, out, capacity, cut_flag);
}

// records: `buf` (n = *n_ptr records, capacity cap) sorted in place; tmp >= cap records; hist >= 2 * 256 * ntiles_cap words
__global__ __launch_bounds__(256) void k_sort_copy_back(const fzb_match_rec* __restrict__ tmp, fzb_match_rec* __restrict__ buf, const u32* __restrict__ n_ptr) {
    const u32 n = *n_ptr;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = tmp[i];
}

// passes = 1: every score is known (on the host, from the scoring and the needle length) to be below 256, so the pass over the high
// byte would move every record to where it already is - the sorted list is copied back instead (a 4 MB copy instead of a 45 us pass)
void fzb_launch_sort(fzb_match_rec* buf, fzb_match_rec* tmp, const u32* n_ptr, u32* hist, u32 ntiles_cap, int reverse_first, int by_score, int grid, hipStream_t st, int passes) {
    // passes = -1: one pass, and the unsorted records are in `tmp` already (the pipeline wrote them there): tmp -> buf, no copy back
    if (reverse_first) hipLaunchKernelGGL(k_reverse, dim3(grid), dim3(256), 0, st, passes == -1 ? tmp : buf, n_ptr);
    if (!by_score) return;
    u32* offs = hist + (size_t)256 * ntiles_cap;  // the scanned histogram (second half of the buffer)
    u32* dtot = hist + (size_t)512 * ntiles_cap;  // behind both: two sets of 256 digit totals + the phase word (FZB_SORT_HIST_WORDS)
    if (passes == -1) {
        hipLaunchKernelGGL(k_sort_hist, dim3(grid), dim3(256), 0, st, tmp, n_ptr, 0, hist, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scan, dim3(16), dim3(1024), 0, st, hist, offs, n_ptr, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(256), 0, st, tmp, buf, n_ptr, 0, offs, ntiles_cap, dtot);
        return;
    }
    if (passes == 1) {
        hipLaunchKernelGGL(k_sort_hist, dim3(grid), dim3(256), 0, st, buf, n_ptr, 0, hist, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scan, dim3(16), dim3(1024), 0, st, hist, offs, n_ptr, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(256), 0, st, buf, tmp, n_ptr, 0, offs, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_copy_back, dim3(grid), dim3(256), 0, st, tmp, buf, n_ptr);
        return;
    }
    for (int pass = 0; pass < 2; pass++) {
        const fzb_match_rec* src = pass == 0 ? buf : tmp;
        fzb_match_rec* dst = pass == 0 ? tmp : buf;
        const int shift = pass * 8;
        hipLaunchKernelGGL(k_sort_hist, dim3(grid), dim3(256), 0, st, src, n_ptr, shift, hist, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scan, dim3(16), dim3(1024), 0, st, hist, offs, n_ptr, ntiles_cap, dtot);
        hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(256), 0, st, src, dst, n_ptr, shift, offs, ntiles_cap, dtot);
    }
}
