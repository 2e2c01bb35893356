// fzb_corpus_upload: what `match_list(&haystacks)` borrows (src/matcher/mod.rs:212), brought into HBM at link speed.
//
// The caller's two arrays (all haystack bytes back to back + exclusive end offsets) travel AS THEY ARE, one hipMemcpy each: the runtime
// pins the pageable source on the fly and the DMA engines run at 52-53 GB/s (measured on the MI355X box, tools/bench_upload.py:
// 10 M x 32 B = 400 MB of bytes + offsets in 7.6 ms, a 12.5 M-item ragged shard = 950 MB in 18 ms).  Round 2 repacked the list on the
// host first (82 ms for the same 10 M list - the repack, not the copy, was the cost).  Two alternatives are kept behind
// FZB_UPLOAD_MODE for comparison: "register" (hipHostRegister + copy: the same 7.6 ms) and "staged" (worker threads copying through
// their own page-locked staging buffers into asynchronous copies: 20 ms, bound by the host memcpy).  The library's device layout
// ("padded-16": every haystack on a 16-byte boundary, zero
// gaps, end offsets inside that layout) is then built ON the device:
//   k_up_tiles    per 1024 haystacks: sum of the padded lengths, min / max length, "offsets decrease" flag
//   k_up_scan     exclusive scan of the tile sums (one workgroup)
//   k_up_build    per tile: the haystacks' padded starts (scan in LDS), the end offsets in the padded layout, and the bytes - one
//                 thread per 16-byte OUTPUT vector (binary search of its haystack in the tile's starts), so writes are whole aligned
//                 vectors and reads run along each haystack
// A list whose lengths are all multiples of 16 (the 32-byte bench lists) already IS the padded layout: the uploaded buffer is kept and
// only the offsets are rewritten.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "host_internal.h"

namespace {

constexpr int UP_TILE = 1024;
constexpr int UP_THREADS = 256;

// stats block (device, 64 bytes): [0] total padded bytes (u64) [1] min len [2] max len (u64 each) [3] bad flag
struct UpStats {
    u64 total_padded, min_len, max_len, bad;
};

__global__ __launch_bounds__(UP_THREADS) void k_up_tiles(const u64* __restrict__ ends, u64 n, u64 ends_base, u64* __restrict__ tile_sums, UpStats* __restrict__ stats) {
    __shared__ u64 s_sum[UP_THREADS / 64];
    __shared__ u64 s_min[UP_THREADS / 64];
    __shared__ u64 s_max[UP_THREADS / 64];
    const u64 tile = blockIdx.x;
    u64 sum = 0, mn = ~(u64)0, mx = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < UP_TILE / UP_THREADS; k++) {
        const u64 i = tile * UP_TILE + (u64)k * UP_THREADS + threadIdx.x;
        if (i < n) {
            const u64 e = ends[i], p = i ? ends[i - 1] : ends_base;
            if (e < p) bad = true;
            const u64 len = e >= p ? e - p : 0;
            sum += (len + 15) & ~(u64)15;
            mn = min(mn, len);
            mx = max(mx, len);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        mn = min(mn, (u64)__shfl_xor(mn, off));
        mx = max(mx, (u64)__shfl_xor(mx, off));
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr((unsigned long long*)&stats->bad, 1ull);
    if ((threadIdx.x & 63) == 0) {
        s_sum[threadIdx.x >> 6] = sum;
        s_min[threadIdx.x >> 6] = mn;
        s_max[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0, a = ~(u64)0, b = 0;
        for (int w = 0; w < UP_THREADS / 64; w++) {
            t += s_sum[w];
            a = min(a, s_min[w]);
            b = max(b, s_max[w]);
        }
        tile_sums[tile] = t;
        atomicMin((unsigned long long*)&stats->min_len, (unsigned long long)a);
        atomicMax((unsigned long long*)&stats->max_len, (unsigned long long)b);
    }
}

// exclusive scan of the tile sums in place (one workgroup of 1024 threads, chunks of 1024 tiles with a running carry)
__global__ __launch_bounds__(1024) void k_up_scan(u64* __restrict__ tile_sums, u64 ntiles, UpStats* __restrict__ stats) {
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < ntiles; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < ntiles ? tile_sums[i] : 0;
        u64 incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const u64 t = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += t;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        u64 wave_base = 0;
        for (u32 w = 0; w < (threadIdx.x >> 6); w++) wave_base += s_wave[w];
        const u64 carry = s_carry;
        if (i < ntiles) tile_sums[i] = carry + wave_base + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + wave_base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) stats->total_padded = s_carry;
}

// ET = u32 / u64 end offsets of the padded layout.  COPY = false: the raw buffer already is the padded layout (only offsets are written).
template <typename ET, bool COPY>
__global__ __launch_bounds__(UP_THREADS) void k_up_build(const u8* __restrict__ raw, const u64* __restrict__ ends, u64 n, u64 ends_base, const u64* __restrict__ tile_base,
                                                         u8* __restrict__ padded, ET* __restrict__ ends_out) {
    __shared__ u64 s_pstart[UP_TILE + 1];  // padded start of each haystack of the tile, relative to the tile's base
    __shared__ u64 s_wave[UP_THREADS / 64];
    const u64 tile = blockIdx.x;
    const u64 i0 = tile * UP_TILE;
    const u64 base = tile_base[tile];
    // every thread owns 4 CONSECUTIVE haystacks: serial prefix of their padded lengths, then a scan over the threads
    const u64 first = i0 + (u64)threadIdx.x * 4;
    const u64 e_prev = first == 0 ? ends_base : (first <= n ? ends[first - 1] : 0);
    u64 plen[4], e[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = first + k;
        const u64 p = k ? e[k - 1] : e_prev;
        e[k] = i < n ? ends[i] : p;
        plen[k] = ((e[k] - p) + 15) & ~(u64)15;
        mine += plen[k];
    }
    u64 incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(incl, off);
        if ((int)(threadIdx.x & 63) >= off) incl += t;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 run = incl - mine;
    for (u32 w = 0; w < (threadIdx.x >> 6); w++) run += s_wave[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u64 i = first + k;
        s_pstart[threadIdx.x * 4 + k] = run;
        if (i < n) ends_out[i] = (ET)(base + run + (e[k] - (k ? e[k - 1] : e_prev)));
        run += plen[k];
    }
    if (threadIdx.x == UP_THREADS - 1) s_pstart[UP_TILE] = run;
    __syncthreads();
    if (!COPY) return;
    const u64 tile_bytes = s_pstart[UP_TILE];
    const u32 cnt = (u32)min((u64)UP_TILE, n - i0);
    for (u64 v = (u64)threadIdx.x * 16u; v < tile_bytes; v += UP_THREADS * 16u) {
        // the haystack this output vector belongs to: last j with pstart[j] <= v (empty haystacks share a start with their successor:
        // the search lands on the last of them, the one that owns the bytes)
        u32 lo = 0, hi = cnt;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (s_pstart[mid] <= v) lo = mid;
            else hi = mid;
        }
        const u64 i = i0 + lo;
        const u64 src_lo = i ? ends[i - 1] : ends_base, src_hi = ends[i];
        const u64 off = v - s_pstart[lo];
        const u64 len = src_hi - src_lo;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (off < len) {
            const u64 p = src_lo - ends_base + off;  // byte position in `raw`
            const u32 rem = (u32)min((u64)16, len - off);
            const u32* a = (const u32*)(raw + (p & ~(u64)3));
            const u32 sh = (u32)(p & 3);
            // 16 bytes from an arbitrary byte position: five aligned dwords, funnel-shifted (the raw buffer has >= 96 readable bytes of slack)
            const u32 w0 = a[0], w1 = a[1], w2 = a[2], w3 = a[3], w4 = sh ? a[4] : 0u;
            u32 x[4] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh), __builtin_amdgcn_alignbyte(w4, w3, sh)};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 lo_b = 4u * k;
                if (rem <= lo_b) x[k] = 0;
                else if (rem - lo_b < 4) x[k] &= (1u << (8 * (rem - lo_b))) - 1;
            }
            q = make_uint4(x[0], x[1], x[2], x[3]);
        }
        *(uint4*)(padded + base + v) = q;
    }
}

// The streaming filter's VIEW of a ragged list (CorpusDev::vbytes; layout described there): two passes over the canonical layout, one
// workgroup per 1024-haystack tile.
//   k_up_view_sort   ranks the tile's haystacks by descending vector count (counting sort, LDS atomics - upload time), writes vperm / vlen,
//                    the vectors-per-member of each of its 16 groups, and the tile's size in the view (in 16-byte units)
//   (k_up_scan: exclusive scan of the tile sizes - the same kernel the canonical layout uses)
//   k_up_view_fill   writes the groups' block offsets and copies every vector to its interleaved position (the buffer was cleared: the
//                    zero vectors behind a group's shorter members are already there)
#define FV_CLASSES 258  // sort key = the haystack's LENGTH, 0..256 bytes (round 5; rounds 3-4: its vector count, 18 classes), and a guard class
// Round 5: a group's LAST vector row is stored as narrow as its longest member's tail allows - 4, 8, 12 or 16 bytes per member instead of 16
// (vgnv[group] = vectors per member | (tail bytes / 4 - 1) << 5) - and the tile is sorted by exact length, so that the 64 members of a group end
// within a few bytes of each other: the view of the C4 shard shrinks from 1.19 to 1.12 x the haystack bytes (padded-16 alone costs 1.106).
template <typename ET>
__global__ __launch_bounds__(UP_THREADS) void k_up_view_sort(const ET* __restrict__ ends, u64 n, u16* __restrict__ vperm, u16* __restrict__ vlen, u8* __restrict__ vgnv,
                                                             u64* __restrict__ tile_units, UpStats* __restrict__ stats, u32* __restrict__ vlong, u32 long_cap) {
    __shared__ u32 s_len[UP_TILE];
    __shared__ u16 s_inv[UP_TILE];
    __shared__ u32 s_hist[FV_CLASSES], s_base[FV_CLASSES];
    __shared__ u32 s_gunits[UP_TILE / 64];
    const u64 i0 = (u64)blockIdx.x * UP_TILE;
    const u32 nt = (u32)min((u64)UP_TILE, n - i0);
    const int tid = threadIdx.x;
    for (int c = tid; c < FV_CLASSES; c += UP_THREADS) s_hist[c] = 0;
    __syncthreads();
    u32 cls[4], rk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 j = tid + UP_THREADS * k;
        cls[k] = rk[k] = 0;
        if (j < nt) {
            const ET st = (i0 + j) ? (ends[i0 + j - 1] + (ET)15) & ~(ET)15 : (ET)0;
            u32 len = (u32)min((u64)(ends[i0 + j] - st), (u64)0xFFFFu);
            if (len > 256u) {  // an OUTLIER: not in the view (no vectors, sorted last, vlen = 0xFFFF) - listed for the filter's follow-up launch
                const u32 slot = (u32)atomicAdd((unsigned long long*)&stats->bad, 1ull);  // (bad = the outlier count here)
                if (slot < long_cap) vlong[slot] = (u32)(i0 + j);
                len = 0xFFFFu;
            }
            s_len[j] = len;
            cls[k] = len == 0xFFFFu ? 0u : min(len, (u32)FV_CLASSES - 1);
            rk[k] = atomicAdd(&s_hist[cls[k]], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {  // descending: the longest haystacks first
        u32 run = 0, top = 0, low = FV_CLASSES;
        for (int c = FV_CLASSES - 1; c >= 0; c--) {
            s_base[c] = run;
            run += s_hist[c];
            if (s_hist[c]) { top = max(top, (u32)c); low = min(low, (u32)c); }
        }
        atomicMax((unsigned long long*)&stats->max_len, (unsigned long long)((top + 15u) >> 4));  // in VECTORS: the view's widest / narrowest member
        atomicMin((unsigned long long*)&stats->min_len, (unsigned long long)((low + 15u) >> 4));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 j = tid + UP_THREADS * k;
        if (j < nt) s_inv[s_base[cls[k]] + rk[k]] = (u16)j;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 p = tid + UP_THREADS * k;  // sorted position
        if (p < nt) {
            const u32 j = s_inv[p];
            vperm[i0 + p] = (u16)j;
            vlen[i0 + p] = (u16)s_len[j];
        }
    }
    if (tid < UP_TILE / 64) {  // the tile's 16 groups: vectors per member and tail width = the group's first (longest) member's
        const u32 p0 = tid * 64;
        const u32 l0 = p0 < nt ? s_len[s_inv[p0]] : 0u;
        const u32 len0 = l0 == 0xFFFFu ? 0u : l0;
        const u32 nv = (len0 + 15u) >> 4;
        const u32 tw = nv ? ((len0 - 16u * (nv - 1) + 3u) & ~3u) : 0u;  // 4, 8, 12 or 16 bytes of the last vector are stored per member
        vgnv[i0 / 64 + tid] = (u8)(nv ? (nv | ((tw / 4 - 1) << 5)) : 0u);
        s_gunits[tid] = nv ? (nv - 1) * 64 + tw * 4 : 0u;  // 16-byte units of the group's block: nv - 1 rows of 1 KiB + 64 tails of tw bytes
    }
    __syncthreads();
    if (tid == 0) {
        u64 t = 0;
        for (int g = 0; g < UP_TILE / 64; g++) t += s_gunits[g];
        tile_units[blockIdx.x] = t;
    }
}

template <typename ET>
__global__ __launch_bounds__(UP_THREADS) void k_up_view_fill(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 n, const u16* __restrict__ vperm,
                                                             const u8* __restrict__ vgnv, const u64* __restrict__ tile_base_units, u8* __restrict__ vbytes, u32* __restrict__ vgofs) {
    __shared__ u32 s_gofs[UP_TILE / 64], s_gnv[UP_TILE / 64], s_gtw[UP_TILE / 64];
    const u64 i0 = (u64)blockIdx.x * UP_TILE;
    const u32 nt = (u32)min((u64)UP_TILE, n - i0);
    const int tid = threadIdx.x;
    if (tid == 0) {
        u64 run = tile_base_units[blockIdx.x];
        for (int g = 0; g < UP_TILE / 64; g++) {
            const u32 code = vgnv[i0 / 64 + g], nv = code & 31u, tw = nv ? ((code >> 5) + 1u) * 4u : 0u;
            s_gofs[g] = (u32)run;
            s_gnv[g] = nv;
            s_gtw[g] = tw;
            vgofs[i0 / 64 + g] = (u32)run;
            run += nv ? (u64)(nv - 1) * 64 + (u64)tw * 4 : 0;
        }
    }
    __syncthreads();
    // four threads per sorted haystack, each copying every fourth vector; the group's last row holds `tw` bytes per member
    for (u32 p = tid >> 2; p < nt; p += UP_THREADS / 4) {
        const u64 j = i0 + vperm[i0 + p];
        const ET st = j ? (ends[j - 1] + (ET)15) & ~(ET)15 : (ET)0;
        if (ends[j] - st > (ET)256) continue;  // an outlier has no vectors in the view
        const u32 nv = (u32)((ends[j] - st + (ET)15) >> 4);
        const u32 gnv = s_gnv[p >> 6], gtw = s_gtw[p >> 6];
        const uint4* src = (const uint4*)(bytes + st);
        u8* gblock = vbytes + (size_t)s_gofs[p >> 6] * 16;
        for (u32 v = tid & 3; v < nv; v += 4) {
            const uint4 q = src[v];
            if (v + 1 < gnv) {
                *(uint4*)(gblock + (size_t)v * 1024 + (size_t)(p & 63) * 16) = q;
            } else {  // (a member as long in vectors as its group's longest: its last vector goes to the narrow row; bytes beyond `gtw` are zero)
                u32* dst = (u32*)(gblock + (size_t)(gnv - 1) * 1024 + (size_t)(p & 63) * gtw);
                dst[0] = q.x;
                if (gtw > 4) dst[1] = q.y;
                if (gtw > 8) dst[2] = q.z;
                if (gtw > 12) dst[3] = q.w;
            }
        }
    }
}

// ---- host -> device at link speed ---------------------------------------------------------------------------------------------
// One hipMemcpy per array straight from the caller's pageable memory: 52-53 GB/s measured (profiles/r03_upload_modes.txt; hipHostRegister
// first and a pool of threads with pinned staging buffers were both slower there and left with round 6's prune).
struct H2DJob {
    void* dst;
    const void* src;
    size_t bytes;
};

// copies every job; returns hipSuccess or the first error
hipError_t h2d_all(const std::vector<H2DJob>& jobs, int device) {
    (void)device;
    for (const H2DJob& j : jobs)
        if (j.bytes) {
            hipError_t e = hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

}  // namespace

// The streaming filter's view of a corpus whose canonical layout is resident (uploaded or borrowed), on the CURRENT device.  Sets
// c->dev.v* and view_nv on success; leaves the corpus without a view (and returns FZB_OK) when the list does not call for one - more
// than one haystack in 256 (+64) beyond 256 bytes (the few that are become OUTLIERS: listed in vlong, decided by k1_cdfa_outliers from
// the canonical layout), nothing beyond 32, a uniform-length list - or when the device has no room: the view is an accelerator,
// not part of the corpus (the filter then streams the canonical layout).  Any other error is reported.
int fzb_build_filter_view(fzb_corpus* c) {
    const bool want_view = !fzb_knobs().no_filter_view;
    const u64 n = c->dev.n;
    if (!want_view || !n || c->dev.uniform_len || c->dev.vbytes) return FZB_OK;
    const u64 ntiles = (n + UP_TILE - 1) / UP_TILE;
    const size_t ngroups = (size_t)ntiles * (UP_TILE / 64);
    u64* d_vt = nullptr;
    UpStats* d_stats = nullptr;
    // outliers (haystacks beyond 256 bytes) the view tolerates: one in 256, at least 64 - more, and the list is not a short-haystack list
    const u32 long_cap = (u32)std::min<u64>(n / 256 + 64, 0x7FFFFFFFu);
    auto drop_view = [&]() {
        if (d_vt) (void)hipFree(d_vt);
        if (d_stats) (void)hipFree(d_stats);
        d_vt = nullptr;
        d_stats = nullptr;
        for (int q = 0; q < 6; q++) { if (c->own_view[q]) (void)hipFree(c->own_view[q]); c->own_view[q] = nullptr; }
        (void)hipGetLastError();
    };
    auto bail = [&](hipError_t e) {
        drop_view();
        return fzb_fail(FZB_ERR_HIP, std::string("filter view: ") + hipGetErrorString(e));
    };
    hipError_t e = fzb_dev_alloc(&c->own_view[3], n * 2);
    if (e == hipSuccess) e = fzb_dev_alloc(&c->own_view[4], n * 2);
    if (e == hipSuccess) e = fzb_dev_alloc(&c->own_view[2], ngroups);
    if (e == hipSuccess) e = fzb_dev_alloc(&c->own_view[1], ngroups * 4);
    if (e == hipSuccess) e = fzb_dev_alloc(&c->own_view[5], (size_t)long_cap * 4);
    if (e == hipSuccess) e = fzb_dev_alloc((void**)&d_vt, (size_t)ntiles * 8);
    if (e == hipSuccess) e = fzb_dev_alloc((void**)&d_stats, sizeof(UpStats));
    if (e == hipErrorOutOfMemory) { drop_view(); return FZB_OK; }
    if (e != hipSuccess) return bail(e);
    const UpStats init{0, ~(u64)0, 0, 0};
    e = hipMemcpy(d_stats, &init, sizeof(init), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(e);
    if (c->dev.ends_u64)
        hipLaunchKernelGGL((k_up_view_sort<u64>), dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, (const u64*)c->dev.ends, n, (u16*)c->own_view[4], (u16*)c->own_view[3], (u8*)c->own_view[2], d_vt, d_stats, (u32*)c->own_view[5], long_cap);
    else
        hipLaunchKernelGGL((k_up_view_sort<u32>), dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, (const u32*)c->dev.ends, n, (u16*)c->own_view[4], (u16*)c->own_view[3], (u8*)c->own_view[2], d_vt, d_stats, (u32*)c->own_view[5], long_cap);
    hipLaunchKernelGGL(k_up_scan, dim3(1), dim3(1024), 0, nullptr, d_vt, ntiles, d_stats);  // total_padded = the view's size in 16-byte units
    UpStats vst{0, 0, 0, 0};
    e = hipMemcpy(&vst, d_stats, sizeof(vst), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(e);
    // (min_len / max_len are in VECTORS here, outliers excluded; bad = the number of outliers) no view: more outliers beyond 256 bytes than
    // the list holds, nothing beyond 32 bytes (the short kernels serve that list), or group offsets beyond 32 bits of 16-byte units (64 GB)
    if (vst.bad > long_cap || vst.max_len > 16 || vst.max_len <= 2 || vst.total_padded > 0xFFFFFFF0ull) { drop_view(); return FZB_OK; }
    const u64 view_bytes = vst.total_padded * 16;
    e = fzb_dev_alloc(&c->own_view[0], view_bytes + 1024);
    if (e == hipErrorOutOfMemory) { drop_view(); return FZB_OK; }
    if (e == hipSuccess) e = hipMemsetAsync(c->own_view[0], 0, view_bytes + 1024, nullptr);
    if (e != hipSuccess) return bail(e);
    if (c->dev.ends_u64)
        hipLaunchKernelGGL((k_up_view_fill<u64>), dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, c->dev.bytes, (const u64*)c->dev.ends, n, (const u16*)c->own_view[4], (const u8*)c->own_view[2], (const u64*)d_vt, (u8*)c->own_view[0], (u32*)c->own_view[1]);
    else
        hipLaunchKernelGGL((k_up_view_fill<u32>), dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, c->dev.bytes, (const u32*)c->dev.ends, n, (const u16*)c->own_view[4], (const u8*)c->own_view[2], (const u64*)d_vt, (u8*)c->own_view[0], (u32*)c->own_view[1]);
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return bail(e);
    (void)hipFree(d_vt);
    (void)hipFree(d_stats);
    c->dev.vbytes = (const u8*)c->own_view[0];
    c->dev.vgofs = (const u32*)c->own_view[1];
    c->dev.vgnv = (const u8*)c->own_view[2];
    c->dev.vlen = (const u16*)c->own_view[3];
    c->dev.vperm = (const u16*)c->own_view[4];
    c->dev.view_nv = (u32)vst.max_len;
    c->dev.vlong = (const u32*)c->own_view[5];
    c->dev.n_long = (u32)vst.bad;
    return FZB_OK;
}

// For a BORROWED corpus (fzb_corpus_from_device; fzb_corpus_upload builds the view itself): the lengths are read from the end offsets,
// so no hint is needed and a wrong fzb_corpus_set_max_len cannot mislead it.  *out_built (optional) = 1 when the corpus has a view now.
extern "C" int fzb_corpus_build_view(fzb_corpus* c, int* out_built) {
    if (!c) return fzb_fail(FZB_ERR_INVALID, "null argument");
    const int rc = fzb_build_filter_view(c);
    if (out_built) *out_built = c->dev.vbytes != nullptr;
    return rc;
}

// The upload proper, on the CURRENT device: `bytes` points at the first byte of haystack 0 of this list, `end_offsets[i]` are exclusive
// ends counted from `ends_base` (0 for a whole list; a shard passes the end of the haystack before its first one).
int fzb_corpus_upload_impl(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, uint64_t ends_base, fzb_corpus** out) {
    if (!out || (n && (!bytes || !end_offsets))) return fzb_fail(FZB_ERR_INVALID, "null argument");
    if (n > 0xFFFFFFFFull)
        return fzb_fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(n) + " > 4294967295 (index offset: 0)");
    if (n && end_offsets[n - 1] < ends_base) return fzb_fail(FZB_ERR_INVALID, "end_offsets must be non-decreasing");
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    const u64 raw_bytes = n ? end_offsets[n - 1] - ends_base : 0;
    const u64 ntiles = (n + UP_TILE - 1) / UP_TILE;
    auto c = new fzb_corpus();
    u8* d_raw = nullptr;
    u64* d_ends64 = nullptr;
    u64* d_tiles = nullptr;
    UpStats* d_stats = nullptr;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_raw, (void*)d_ends64, (void*)d_tiles, (void*)d_stats})
            if (p) (void)hipFree(p);
    };
    auto bail = [&](hipError_t e, const char* what) {
        cleanup();
        fzb_corpus_free(c);
        return fzb_fail(FZB_ERR_HIP, std::string("corpus upload (") + what + "): " + hipGetErrorString(e));
    };
    hipError_t e = fzb_dev_alloc((void**)&d_raw, raw_bytes + 96);
    if (e == hipSuccess) e = fzb_dev_alloc((void**)&d_ends64, std::max<size_t>(n, 1) * 8);
    if (e == hipSuccess) e = fzb_dev_alloc((void**)&d_tiles, std::max<u64>(ntiles, 1) * 8);
    if (e == hipSuccess) e = fzb_dev_alloc((void**)&d_stats, sizeof(UpStats));
    if (e != hipSuccess) return bail(e, "device buffers");
    const UpStats init{0, ~(u64)0, 0, 0};
    e = hipMemcpy(d_stats, &init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_raw + raw_bytes, 0, 96);
    if (e == hipSuccess) e = h2d_all({H2DJob{d_ends64, end_offsets, n * 8}, H2DJob{d_raw, bytes, (size_t)raw_bytes}}, device);
    if (e != hipSuccess) return bail(e, "host to device");
    UpStats st{0, 0, 0, 0};
    if (n) {
        hipLaunchKernelGGL(k_up_tiles, dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, d_ends64, (u64)n, ends_base, d_tiles, d_stats);
        hipLaunchKernelGGL(k_up_scan, dim3(1), dim3(1024), 0, nullptr, d_tiles, ntiles, d_stats);
        e = hipMemcpy(&st, d_stats, sizeof(st), hipMemcpyDeviceToHost);  // the one synchronisation of the upload: sizes the padded buffer
        if (e != hipSuccess) return bail(e, "layout pass");
        if (st.bad) {
            cleanup();
            fzb_corpus_free(c);
            return fzb_fail(FZB_ERR_INVALID, "end_offsets must be non-decreasing");
        }
    }
    const u64 total = st.total_padded + 96;
    const bool ends_u64 = total > 0xFFFFFFF0ull;
    const bool adopt = st.total_padded == raw_bytes;  // every length a multiple of 16: the upload format is the device layout
    c->dev.n = n;
    c->dev.total_bytes = total;
    c->dev.ends_u64 = ends_u64;
    c->dev.max_len = (u32)std::min<u64>(st.max_len, 0xFFFFFFFFu);
    c->dev.uniform_len = (n && st.min_len == st.max_len && st.max_len && st.max_len < 0xFFFFFFFFu) ? (u32)st.max_len : 0u;  // kernels then skip the end offsets
    e = fzb_dev_alloc(&c->own_ends, std::max<size_t>(n, 1) * (ends_u64 ? 8 : 4));
    if (e == hipSuccess && !adopt) e = fzb_dev_alloc(&c->own_bytes, total);
    if (e != hipSuccess) return bail(e, "padded layout");
    if (adopt) {
        c->own_bytes = d_raw;
        d_raw = nullptr;
    } else {
        e = hipMemsetAsync((u8*)c->own_bytes + st.total_padded, 0, 96, nullptr);
        if (e != hipSuccess) return bail(e, "padded layout");
    }
    if (n) {
#define FZB_UP_BUILD(ET, COPY) \
    hipLaunchKernelGGL((k_up_build<ET, COPY>), dim3((unsigned)ntiles), dim3(UP_THREADS), 0, nullptr, (const u8*)(adopt ? (u8*)c->own_bytes : d_raw), d_ends64, (u64)n, ends_base, d_tiles, (u8*)c->own_bytes, (ET*)c->own_ends)
        if (ends_u64) { if (adopt) FZB_UP_BUILD(u64, false); else FZB_UP_BUILD(u64, true); }
        else { if (adopt) FZB_UP_BUILD(u32, false); else FZB_UP_BUILD(u32, true); }
#undef FZB_UP_BUILD
    }
    c->dev.bytes = (const u8*)c->own_bytes;
    c->dev.ends = c->own_ends;
    // the streaming filter's view (CorpusDev::vbytes): ragged lists whose haystacks are 33..256 bytes.  A second copy of the bytes (+ ~5 %
    // for the zero vectors behind shorter group members, + 4.2 bytes per haystack); FZB_FILTER_VIEW=0 turns it off.
    if (n && !c->dev.uniform_len && c->dev.max_len > 32) {  // (a list with more than a few haystacks beyond 256 bytes gets none: the builder decides)
        int rc = fzb_build_filter_view(c);
        if (rc) {
            const std::string msg = fzb_last_error();
            cleanup();
            fzb_corpus_free(c);
            return fzb_fail(rc, msg);
        }
    }
    e = hipDeviceSynchronize();  // the temporaries are released below; the corpus is complete when the call returns
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return bail(e, "layout kernels");
    cleanup();
    *out = c;
    return FZB_OK;
}

extern "C" int fzb_corpus_upload(const uint8_t* bytes, const uint64_t* end_offsets, size_t n, fzb_corpus** out) {
    return fzb_corpus_upload_impl(bytes, end_offsets, n, 0, out);
}
