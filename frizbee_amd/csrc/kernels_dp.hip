// gfx950 kernels, stage 2b: Smith-Waterman scoring of the survivors whose trimmed window fits ONE chunk of
// the emulated CPU backend (<= SWL bytes; every len-32 haystack).  One THREAD per haystack: the SWL-lane
// score row lives in SWL/2 VGPRs as packed u16 pairs and is updated with v_pk_{add,sub(clamp),max,mul_lo}_u16;
// the reference's `shift_right_padded::<L>` for L >= 2 lanes is a compile-time register renaming, L = 1 is
// one v_alignbit_b32.  No MFMA: this is saturating integer DP, not a contraction.
//
// Restates, bit-exactly for a single chunk (adjacent chunk = the zero column):
//   score_haystack                src/smith_waterman/algo/ascii.rs:10-158
//   propagate_horizontal_gaps     src/smith_waterman/algo/ascii_gap.rs:11-105   (log-step scan, steps 1..SWL/2)
//   trim_haystack / smith_waterman_one / exact bonus   src/matcher/algo.rs:230-263, 332-338
//   0-typo ASCII window (first occurrence of needle[0], last occurrence of needle[n-1])  src/prefilter/algo/ascii.rs:6-72
// Windows wider than one chunk (or > 1024 bytes) are handed to the generic scorer (kernels_generic.hip).
//
// Gap propagation runs in a biased domain r'[L] = r[L] + L*gex, where "shift by k and pay k*gex" becomes a
// plain shift; since every row value is >= 0 the saturating subtract of the reference is preserved exactly
// (see DESIGN.md).  BIAS=false keeps the literal 3-op form for scorings where the bias could overflow u16.
#include "kernels_common.h"

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(u32 x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ u32 as_u32(us2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 p_add(u32 a, u32 b) { return as_u32(as_us2(a) + as_us2(b)); }
__device__ __forceinline__ u32 p_sub(u32 a, u32 b) { return as_u32(as_us2(a) - as_us2(b)); }
__device__ __forceinline__ u32 p_subs(u32 a, u32 b) { return as_u32(__builtin_elementwise_sub_sat(as_us2(a), as_us2(b))); }
__device__ __forceinline__ u32 p_max(u32 a, u32 b) { return as_u32(__builtin_elementwise_max(as_us2(a), as_us2(b))); }
__device__ __forceinline__ u32 p_mul(u32 a, u32 b) { return as_u32(as_us2(a) * as_us2(b)); }
__device__ __forceinline__ u32 splat16(u32 v) { return (v & 0xFFFF) * 0x00010001u; }

__device__ __forceinline__ u32 zero_bytes4_dp(u32 x) {
    u32 y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    y = ~y & 0x80808080u;
    return ((y >> 7) * 0x00204081u >> 21) & 0xF;
}

template <int SWL, bool BIAS, typename ET>
__global__ __launch_bounds__(128) void k2b_dp(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 index_offset,
                                              const u32* __restrict__ items, const u32* __restrict__ win, const u32* __restrict__ n_items_ptr,
                                              const NeedleDev nd, int wmode, fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count,
                                              u32* __restrict__ overflow, u32* __restrict__ counters) {
    constexpr int NW = SWL / 2;  // packed score dwords
    constexpr int NB = SWL / 4;  // haystack byte dwords
    __shared__ u8 cls[256];      // bit0 lower, bit1 upper, bit2 delimiter (ascii.rs:65-89)
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        const bool lower = b >= 'a' && b <= 'z', upper = b >= 'A' && b <= 'Z', digit = b >= '0' && b <= '9';
        const bool delim = !(lower || upper || digit || b > 127);
        cls[b] = (u8)((lower ? 1 : 0) | (upper ? 2 : 0) | (delim ? 4 : 0));
    }
    __syncthreads();
    const u32 M = *n_items_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) *dev_count = M < capacity ? M : capacity;
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);

    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < M; j += gridDim.x * blockDim.x) {
        if (j >= capacity) continue;
        const u32 li = items ? items[j] : j;
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        const u8* hay = bytes + s;
        // ---- window -------------------------------------------------------------------------------
        u32 ws, we;
        if (wmode == 0) {
            ws = win[2 * j];
            we = win[2 * j + 1];
        } else if (wmode == 2) {
            ws = 0;
            we = L;
        } else {
            // first occurrence of needle[0] (either case), 1 + last occurrence of needle[rows-1]
            const u32 a0 = nd.c[0] * 0x01010101u, a1 = nd.f[0] * 0x01010101u;
            const u32 z0 = nd.c[rows - 1] * 0x01010101u, z1 = nd.f[rows - 1] * 0x01010101u;
            ws = 0xFFFFFFFFu;
            we = 0;
            const uint4* vp = (const uint4*)hay;
            const u32 nvec = (L + 15) >> 4;
            for (u32 v = 0; v < nvec; v++) {
                const uint4 q = vp[v];
                const u32 w4[4] = {q.x, q.y, q.z, q.w};
                u32 mf = 0, ml = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    mf |= (zero_bytes4_dp(w4[k] ^ a0) | zero_bytes4_dp(w4[k] ^ a1)) << (4 * k);
                    ml |= (zero_bytes4_dp(w4[k] ^ z0) | zero_bytes4_dp(w4[k] ^ z1)) << (4 * k);
                }
                const u32 rem = L - 16 * v;
                const u32 vm = rem >= 16 ? 0xFFFFu : ((1u << rem) - 1);
                mf &= vm;
                ml &= vm;
                if (ws == 0xFFFFFFFFu && mf) ws = 16 * v + __builtin_ctz(mf);
                if (ml) we = 16 * v + 32 - __builtin_clz(ml);
            }
            if (ws == 0xFFFFFFFFu) ws = 0;  // cannot happen for a survivor of the exact filter
        }
        // ---- trim_haystack (matcher/algo.rs:332-338) --------------------------------------------------
        const u32 sp = ws ? ws - 1 : 0;
        const bool include_exact = sp == 0 && we == L;
        const u32 m = we - sp;
        if (m > (u32)SWL) {
            const u32 slot = atomicAdd(&counters[3], 1u);
            overflow[3 * slot] = j;
            overflow[3 * slot + 1] = ws;
            overflow[3 * slot + 2] = we;
            continue;
        }
        u32 score = 0;
        u32 hb[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const u32 p = 4 * k;
            u32 v = 0;
            if (p < m) {
                v = load_u32_unaligned(hay, sp + p);
                const u32 rem = m - p;
                if (rem < 4) v &= (1u << (8 * rem)) - 1;
            }
            hb[k] = v;
        }
        if (m > 0) {
            // ---- haystack-side vectors (ascii.rs:59-101) ----------------------------------------------
            u32 hw[NW], bonus[NW];
            {
                u32 clsw_prev = 0;
#pragma unroll
                for (int d = 0; d < NW; d++) {
                    const u32 w = hb[d / 2];
                    const u32 b0 = (d & 1) ? (w >> 16) & 0xFF : w & 0xFF;
                    const u32 b1 = (d & 1) ? w >> 24 : (w >> 8) & 0xFF;
                    hw[d] = b0 | (b1 << 16);
                    const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
                    const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);  // class of lane-1 (lane -1 of chunk 0: none)
                    const u32 cap01 = (clsw >> 1) & sh & ONE;                        // upper(j) & lower(j-1)
                    const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;                 // delim(j-1) & !delim(j)
                    bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
                    clsw_prev = clsw;
                }
                if (sp == 0) bonus[0] = p_add(bonus[0], (u32)nd.prefix);  // first_lane(prefix_bonus), include_prefix = (start == 0)
            }
            u32 prev[NW], gprev[NW];
#pragma unroll
            for (int d = 0; d < NW; d++) prev[d] = 0, gprev[d] = 0;
#pragma unroll 1
            for (u32 r = 0; r < rows; r++) {
                const u32 c = nd.c[r], f = nd.f[r];
                const bool ci = c != f;  // case-insensitive ASCII letter: (h | 0x20) == (c | 0x20) <=> h in {c, flip(c)}
                const u32 orv = ci ? 0x00200020u : 0u;
                const u32 cmpv = splat16(ci ? (c | 0x20) : c);
                const u32 cv = splat16(c);
                u32 row[NW], g[NW];
#pragma unroll
                for (int d = 0; d < NW; d++) {
                    const u32 mm = p_subs(ONE, (hw[d] | orv) ^ cmpv);      // match mask as 0/1 per lane
                    const u32 ex = ci ? p_subs(ONE, hw[d] ^ cv) : mm;      // exact-case match
                    const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : 0u, 16);  // S(i-1, j-1)
                    u32 t = p_add(p_mul(mm, bonus[d]), sh);
                    t = p_subs(t, Xv);
                    const u32 diag = p_add(p_mul(ex, casev), t);
                    const u32 up = p_subs(p_subs(prev[d], gexv), gprev[d]);
                    row[d] = p_max(diag, up);
                    g[d] = p_mul(mm, gopmv);
                }
                // ---- propagate_horizontal_gaps: steps 1, 2, 4, ..., SWL/2 -----------------------------
                if (BIAS) {
                    u32 b[NW];
#pragma unroll
                    for (int d = 0; d < NW; d++) b[d] = p_add(row[d], (u32)nd.gex * (u32)(2 * d + ((2 * d + 1) << 16)));
                    {
                        u32 nb[NW];
#pragma unroll
                        for (int d = 0; d < NW; d++) {
                            const u32 sb = __builtin_amdgcn_alignbit(b[d], d ? b[d - 1] : 0u, 16);
                            const u32 sg = __builtin_amdgcn_alignbit(g[d], d ? g[d - 1] : 0u, 16);
                            nb[d] = p_max(b[d], p_subs(sb, sg));
                        }
#pragma unroll
                        for (int d = 0; d < NW; d++) b[d] = nb[d];
                    }
#pragma unroll
                    for (int off = 1; off < NW; off *= 2) {
                        u32 nb[NW];
#pragma unroll
                        for (int d = 0; d < NW; d++) nb[d] = d >= off ? p_max(b[d], p_subs(b[d - off], g[d - off])) : b[d];
#pragma unroll
                        for (int d = 0; d < NW; d++) b[d] = nb[d];
                    }
#pragma unroll
                    for (int d = 0; d < NW; d++) row[d] = p_sub(b[d], (u32)nd.gex * (u32)(2 * d + ((2 * d + 1) << 16)));
                } else {
                    u32 kg = gexv;
                    {
                        u32 nb[NW];
#pragma unroll
                        for (int d = 0; d < NW; d++) {
                            const u32 sb = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : 0u, 16);
                            const u32 sg = __builtin_amdgcn_alignbit(g[d], d ? g[d - 1] : 0u, 16);
                            nb[d] = p_max(row[d], p_subs(sb, p_add(kg, sg)));
                        }
#pragma unroll
                        for (int d = 0; d < NW; d++) row[d] = nb[d];
                    }
#pragma unroll
                    for (int off = 1; off < NW; off *= 2) {
                        kg = p_add(kg, kg);
                        u32 nb[NW];
#pragma unroll
                        for (int d = 0; d < NW; d++) nb[d] = d >= off ? p_max(row[d], p_subs(row[d - off], p_add(kg, g[d - off]))) : row[d];
#pragma unroll
                        for (int d = 0; d < NW; d++) row[d] = nb[d];
                    }
                }
#pragma unroll
                for (int d = 0; d < NW; d++) prev[d] = row[d], gprev[d] = g[d];
            }
            // ---- max over every lane of the last row (ascii.rs:152-156) -------------------------------
            u32 mx = prev[0];
#pragma unroll
            for (int d = 1; d < NW; d++) mx = p_max(mx, prev[d]);
            score = max(mx & 0xFFFF, mx >> 16);
        }
        // ---- exact flag + bonus (matcher/algo.rs:245-248) ---------------------------------------------
        bool exact = include_exact && m == (u32)nd.nbytes;
        if (exact) {
            const u32* raw = (const u32*)nd.raw;
#pragma unroll
            for (int k = 0; k < NB && k < FZB_MAX_NEEDLE_BYTES / 4; k++) exact = exact && (hb[k] == raw[k]);
        }
        if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
        fzb_match_rec rec;
        rec.index = index_offset + li;
        rec.score = (u16)score;
        rec.exact = exact ? 1 : 0;
        rec.valid = 0;
        out[j] = rec;
    }
}

void fzb_launch_dp(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, const u32* n_items_ptr, const NeedleDev& nd,
                   int sw_lanes, int bias_ok, int wmode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* overflow, u32* counters, int grid, hipStream_t st) {
#define FZB_K2B(SWL, B, ET) hipLaunchKernelGGL((k2b_dp<SWL, B, ET>), dim3(grid), dim3(128), 0, st, c.bytes, (const ET*)c.ends, first, index_offset, items, win, n_items_ptr, nd, wmode, out, capacity, dev_count, overflow, counters)
#define FZB_K2B_ET(SWL, B) do { if (c.ends_u64) FZB_K2B(SWL, B, u64); else FZB_K2B(SWL, B, u32); } while (0)
#define FZB_K2B_B(SWL) do { if (bias_ok) FZB_K2B_ET(SWL, true); else FZB_K2B_ET(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2B_B(64); break;
        case 32: FZB_K2B_B(32); break;
        case 16: FZB_K2B_B(16); break;
        default: FZB_K2B_B(8); break;
    }
}
