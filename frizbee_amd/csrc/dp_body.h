// Thread-per-haystack Smith-Waterman bodies of the DP kernels (kernels_dp.hip): single chunk and chunk-by-chunk.
// See kernels_dp.hip for the reference mapping and DESIGN.md for the biased-domain gap propagation.
#pragma once
#include "kernels_common.h"

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(u32 x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ u32 as_u32(us2 x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ u32 p_add(u32 a, u32 b) { return as_u32(as_us2(a) + as_us2(b)); }
__device__ __forceinline__ u32 p_sub(u32 a, u32 b) { return as_u32(as_us2(a) - as_us2(b)); }
__device__ __forceinline__ u32 p_subs(u32 a, u32 b) { return as_u32(__builtin_elementwise_sub_sat(as_us2(a), as_us2(b))); }
__device__ __forceinline__ u32 p_max(u32 a, u32 b) { return as_u32(__builtin_elementwise_max(as_us2(a), as_us2(b))); }
// max(a, b, s) per 16-bit lane in ONE instruction, for values below 0x7C00 (a, b vectors; s wave-uniform): gfx950's v_pk_maximum3_f16.
// Non-negative finite binary16 patterns order like the integers that spell them and the kernels run with f16 denormals preserved
// (amdhsa_float_denorm_mode_16_64 = 3), so the instruction returns one of its inputs bit for bit - tools/ubench/max3_f16.hip checks it
// against two v_pk_max_u16 on 2.9e8 packed triples (0 differences) and times it at 0.62 of the pair's cost.  Callers guarantee the bound
// (LaunchCfg::cf_ok / cfm_ok: biased scores stay below 0x7C00).
__device__ __forceinline__ u32 p_max3_s(u32 a, u32 b, u32 s) {
#ifdef FZB_HOST_SHIM
    return p_max(p_max(a, b), s);
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "p_max3_s is v_pk_maximum3_f16: gfx950 (MI355X) only - this library is built for that one target (csrc/Makefile refuses any other ARCH)"
#endif
    u32 r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s));
    return r;
#endif
}
__device__ __forceinline__ u32 p_min(u32 a, u32 b) { return as_u32(__builtin_elementwise_min(as_us2(a), as_us2(b))); }
__device__ __forceinline__ u32 p_mul(u32 a, u32 b) { return as_u32(as_us2(a) * as_us2(b)); }
__device__ __forceinline__ u32 splat16(u32 v) { return (v & 0xFFFF) * 0x00010001u; }

__device__ __forceinline__ u32 zero_bytes4_dp(u32 x) {
    u32 y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;
    y = ~y & 0x80808080u;
    return ((y >> 7) * 0x00204081u >> 21) & 0xF;
}

// byte class table (ascii.rs:65-89): bit0 lower, bit1 upper, bit2 delimiter
__device__ __forceinline__ void build_cls_table(u8* cls) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) {
        const bool lower = b >= 'a' && b <= 'z', upper = b >= 'A' && b <= 'Z', digit = b >= '0' && b <= '9';
        const bool delim = !(lower || upper || digit || b > 127);
        cls[b] = (u8)((lower ? 1 : 0) | (upper ? 2 : 0) | (delim ? 4 : 0));
    }
}

// ---- 32 byte positions as one word: bit 8j + k = byte j of dword k (position 4k + j) --------------------------------------------------
// (the layout in which eight dwords' SWAR flags merge with one shift-or each; dp_cf.h's cf_window_first_last_regs introduced it)
// positions q with q + len <= L
__device__ __forceinline__ u32 pos32_valid(u32 L, u32 len) {
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        if (L >= len + j) {
            const u32 kmax = (L - len - j) >> 2;  // positions j, j + 4, ..., j + 4 * kmax are valid
            v |= (kmax >= 7 ? 0xFFu : ((2u << kmax) - 1)) << (8 * j);
        }
    }
    return v;
}
// ... for len == 1 (positions q < L) without the per-byte-lane loop: dwords below L/4 are valid in every byte lane, dword L/4 in L%4 of them
__device__ __forceinline__ u32 pos32_valid1(u32 L) {
    const u32 kf = min(L >> 2, 8u), rem = L & 3;
    u32 v = ((1u << kf) - 1) * 0x01010101u;
    if (kf < 8) v |= (0x01010101u & ((1u << (8 * rem)) - 1)) << kf;
    return v;
}
// smallest position set in y (y != 0); the largest is 31 - pos32_first(bitreverse(y)): reversing maps bit 8j + k to 8(3-j) + (7-k)
__device__ __forceinline__ u32 pos32_first(u32 y) {
    u32 best = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 f = (y >> (8 * j)) & 0xFF;
        const u32 k = f ? (u32)__builtin_ctz(f) : 0x3FFFFFFFu;
        best = min(best, 4 * k + j);
    }
    return best;
}
// 0x80 in every byte of x that is zero (exact)
__device__ __forceinline__ u32 zero_flags4(u32 x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }

// 0-typo ASCII window of an ACCEPTED haystack of any length (src/prefilter/algo/ascii.rs:6-72): first occurrence of needle[0], one past
// the last occurrence of needle[rows-1], either case.  Blocks of 32 bytes, four requested together (the scan needs every vector: last
// occurrence); per block one word of positions per needle byte - a case-folded letter is ONE compare, (h | 0x20) == (c | 0x20) <=> h is c or
// its other case - and what follows the haystack in its last block (its own zero padding, then the next haystack) is masked by position.
// (Rounds 1-2 tested both cases of both bytes per dword and kept 16-bit masks per vector: 40 instructions per dword, now 12.)
__device__ __forceinline__ void window_first_last(const NeedleDev& nd, const u8* __restrict__ hay, u32 L, u32& ws, u32& we) {
    const u32 rows = (u32)nd.rows;
    const u32 a = nd.c[0], af = nd.f[0], z = nd.c[rows - 1], zf = nd.f[rows - 1];
    const u32 aor = a != af ? 0x20202020u : 0u, zor = z != zf ? 0x20202020u : 0u;
    const u32 apat = (a != af ? (a | 0x20) : a) * 0x01010101u, zpat = (z != zf ? (z | 0x20) : z) * 0x01010101u;
    // the block of the first / last occurrence and its position word are kept; the positions are extracted once, after the scan
    u32 ya_keep = 0, yz_keep = 0, ba = 0, bz = 0;
    const uint4* vp = (const uint4*)hay;
    const u32 nblk = (L + 31) >> 5;
    for (u32 b0 = 0; b0 < nblk; b0 += 4) {
        uint4 qs[8];
#pragma unroll
        for (int k = 0; k < 8; k++) qs[k] = 32 * b0 + 16 * k < L ? vp[2 * b0 + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32 b = b0 + t;
            if (b >= nblk) break;
            const u32 w8[8] = {qs[2 * t].x, qs[2 * t].y, qs[2 * t].z, qs[2 * t].w, qs[2 * t + 1].x, qs[2 * t + 1].y, qs[2 * t + 1].z, qs[2 * t + 1].w};
            u32 ya = 0, yz = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                ya |= zero_flags4((w8[k] | aor) ^ apat) >> (7 - k);
                yz |= zero_flags4((w8[k] | zor) ^ zpat) >> (7 - k);
            }
            const u32 valid = pos32_valid1(L - 32 * b);
            ya &= valid;
            yz &= valid;
            if (ya_keep == 0 && ya) { ya_keep = ya; ba = b; }
            if (yz) { yz_keep = yz; bz = b; }
        }
    }
    ws = ya_keep ? 32 * ba + pos32_first(ya_keep) : 0u;  // (no occurrence cannot happen for a survivor of the exact filter)
    we = yz_keep ? 32 * bz + 32u - pos32_first(__builtin_bitreverse32(yz_keep)) : 0u;
}

// The same for a haystack of at most 32 bytes whose two 16-byte vectors are already in registers (the DP kernel's
// software pipeline requested them one iteration ahead).
__device__ __forceinline__ void window_first_last_regs(const NeedleDev& nd, const uint4& q0, const uint4& q1, u32 L, u32& ws, u32& we) {
    const u32 rows = (u32)nd.rows;
    const u32 a0 = nd.c[0] * 0x01010101u, a1 = nd.f[0] * 0x01010101u;
    const u32 z0 = nd.c[rows - 1] * 0x01010101u, z1 = nd.f[rows - 1] * 0x01010101u;
    const u32 w8[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    u32 mf = 0, ml = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        mf |= (zero_bytes4_dp(w8[k] ^ a0) | zero_bytes4_dp(w8[k] ^ a1)) << (4 * k);
        ml |= (zero_bytes4_dp(w8[k] ^ z0) | zero_bytes4_dp(w8[k] ^ z1)) << (4 * k);
    }
    const u32 vm = L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1);
    mf &= vm;
    ml &= vm;
    ws = mf ? (u32)__builtin_ctz(mf) : 0u;  // (mf == 0 cannot happen for a survivor of the exact filter)
    we = ml ? 32u - (u32)__builtin_clz(ml) : 0u;
}

// Window bytes th[0..m) -> hb (zero padded), from memory ...
template <int NB>
__device__ __forceinline__ void load_window_mem(const u8* __restrict__ th, u32 m, u32 (&hb)[NB]) {
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const u32 p = 4 * k;
        u32 v = 0;
        if (p < m) {
            v = load_u32_unaligned(th, p);
            const u32 rem = m - p;
            if (rem < 4) v &= (1u << (8 * rem)) - 1;
        }
        hb[k] = v;
    }
}
// ... or from the 32 bytes of a short haystack held in registers: bytes [sp, sp + m), sp + m <= 32
template <int NB>
__device__ __forceinline__ void load_window_regs(const uint4& q0, const uint4& q1, u32 sp, u32 m, u32 (&hb)[NB]) {
    u32 x[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, 0u};
    const u32 ds = sp >> 2, bs = sp & 3;
    if (ds & 4) {
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] = i + 4 < 9 ? x[i + 4] : 0u;
    }
    if (ds & 2) {
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] = i + 2 < 9 ? x[i + 2] : 0u;
    }
    if (ds & 1) {
#pragma unroll
        for (int i = 0; i < 9; i++) x[i] = i + 1 < 9 ? x[i + 1] : 0u;
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
        u32 v = 0;
        if (k < 8) {
            v = __builtin_amdgcn_alignbyte(x[k + 1], x[k], bs);
            const u32 p = 4 * k;
            const u32 rem = m > p ? m - p : 0u;
            if (rem < 4) v &= (1u << (8 * rem)) - 1;
        }
        hb[k] = v;
    }
}

// Scores the trimmed window (1 <= m <= SWL bytes, given zero padded in hb) as ONE chunk of SWL lanes.
// Returns S = max over all lanes of the last row.
//
// REAL = number of packed dwords (2 lanes each) that may hold haystack bytes: the caller guarantees m <= 2 * REAL.
// Dwords >= REAL are zero padding, which the reference scores like any other lane (they enter the final max) but where,
// for a needle without NUL bytes, nothing can match: match mask = 0, so the bonus / case / gap-open terms vanish and
// the same recurrences reduce to `diag = S(i-1,j-1) (-) X`, `up = S(i-1,j) (-) gex`, and a gap step whose source lies in
// the padding is a plain max.  REAL = NW is the fully general form.
// UPPER = the needle has an uppercase letter (decided on the host; keeps the rarely needed code path, and its
// registers, out of the common instantiation).
template <int SWL, bool BIAS, bool UPPER, int REAL = SWL / 2>
__device__ __forceinline__ u32 dp_single_chunk(const NeedleDev& nd, u32 m, bool include_prefix, const u8* cls, const u32 (&hb)[SWL / 4]) {
    constexpr int NW = SWL / 2;  // packed score dwords
    constexpr int NB = SWL / 4;  // haystack byte dwords
    static_assert(REAL >= 1 && REAL <= NW, "REAL");
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    // ---- haystack-side vectors (ascii.rs:59-101) ----------------------------------------------------
    // The matching-case bonus (ascii.rs:121-131) is folded into the per-lane match bonus: a needle byte c earns it on
    // the lanes where h == c.  For a case-folded letter (match: h in {c, flip(c)}) those are the matching lanes that
    // have c's case; for an exactly compared byte, all matching lanes - which then have c's case too, or are
    // non-letters.  So the stored vector carries the case bonus on the lowercase lanes: right for a lowercase c and -
    // plus the constant - for a non-letter c (non-matching lanes are multiplied by 0 either way).  A needle with an
    // uppercase letter (UPPER, rare under smart case) keeps the literal form with a separate exact-case compare.
    u32 hw[REAL], bonus_lo[REAL];
    {
        u32 clsw_prev = 0;
#pragma unroll
        for (int d = 0; d < REAL; d++) {
            const u32 w = hb[d / 2];
            const u32 b0 = (d & 1) ? (w >> 16) & 0xFF : w & 0xFF;
            const u32 b1 = (d & 1) ? w >> 24 : (w >> 8) & 0xFF;
            hw[d] = b0 | (b1 << 16);
            const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
            const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);  // class of lane-1 (lane -1 of chunk 0: none)
            const u32 cap01 = (clsw >> 1) & sh & ONE;                        // upper(j) & lower(j-1)
            const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;                 // delim(j-1) & !delim(j)
            u32 bonus = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
            if (d == 0 && include_prefix) bonus = p_add(bonus, (u32)nd.prefix);  // first_lane(prefix_bonus)
            bonus_lo[d] = UPPER ? bonus : p_add(bonus, p_mul(clsw & ONE, casev));
            clsw_prev = clsw;
        }
    }
    u32 prev[NW], gprev[REAL];
#pragma unroll
    for (int d = 0; d < NW; d++) prev[d] = 0;
#pragma unroll
    for (int d = 0; d < REAL; d++) gprev[d] = 0;
#pragma unroll 1
    for (u32 r = 0; r < rows; r++) {
        // needle bytes through aligned dword reads of the by-value argument: wave-uniform, so they are scalar loads
        const u32 c = (((const u32*)nd.c)[r >> 2] >> (8 * (r & 3))) & 0xFF, f = (((const u32*)nd.f)[r >> 2] >> (8 * (r & 3))) & 0xFF;
        const bool ci = c != f;  // case-insensitive ASCII letter: (h | 0x20) == (c | 0x20) <=> h in {c, flip(c)}
        const u32 orv = ci ? 0x00200020u : 0u;
        const u32 cmpv = splat16(ci ? (c | 0x20) : c);
        const u32 cv = splat16(c);
        const bool c_lower = c >= 'a' && c <= 'z';
        const u32 extra = c_lower ? 0u : casev;  // a non-letter needle byte: every matching lane is an exact-case match
        u32 row[NW], g[REAL];
#pragma unroll
        for (int d = 0; d < REAL; d++) {
            const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : 0u, 16);  // S(i-1, j-1)
            const u32 mm = p_subs(ONE, (hw[d] | orv) ^ cmpv);                             // match mask as 0/1 per lane
            u32 diag;
            if (UPPER) {
                // literal form: separate exact-case compare (bonus_lo holds the bonus WITHOUT the case term here)
                const u32 ex = ci ? p_subs(ONE, hw[d] ^ cv) : mm;
                diag = p_add(p_mul(ex, casev), p_subs(p_add(p_mul(mm, bonus_lo[d]), sh), Xv));
            } else {
                diag = p_subs(p_add(p_mul(mm, p_add(bonus_lo[d], extra)), sh), Xv);
            }
            const u32 up = p_subs(p_subs(prev[d], gexv), gprev[d]);
            row[d] = p_max(diag, up);
            g[d] = p_mul(mm, gopmv);
        }
#pragma unroll
        for (int d = REAL; d < NW; d++) {
            const u32 sh = __builtin_amdgcn_alignbit(prev[d], prev[d - 1], 16);
            row[d] = p_max(p_subs(sh, Xv), p_subs(prev[d], gexv));  // padding lanes: no match, no gap-open surcharge
        }
        // ---- propagate_horizontal_gaps: steps 1, 2, 4, ..., SWL/2 (ascii_gap.rs:11-105) ---------------
        if (BIAS) {
            u32 b[NW];
#pragma unroll
            for (int d = 0; d < NW; d++) b[d] = p_add(row[d], (u32)nd.gex * (u32)(2 * d + ((2 * d + 1) << 16)));
            {
                u32 nb[NW];
#pragma unroll
                for (int d = 0; d < NW; d++) {
                    const u32 sb = __builtin_amdgcn_alignbit(b[d], d ? b[d - 1] : 0u, 16);
                    if (d <= REAL) {  // source lanes 2d-1, 2d: at least one may be a real lane
                        const u32 sg = __builtin_amdgcn_alignbit(d < REAL ? g[d] : 0u, (d && d - 1 < REAL) ? g[d - 1] : 0u, 16);
                        nb[d] = p_max(b[d], p_subs(sb, sg));
                    } else {
                        nb[d] = p_max(b[d], sb);
                    }
                }
#pragma unroll
                for (int d = 0; d < NW; d++) b[d] = nb[d];
            }
#pragma unroll
            for (int off = 1; off < NW; off *= 2) {
                u32 nb[NW];
#pragma unroll
                for (int d = 0; d < NW; d++) nb[d] = d >= off ? ((d - off) < REAL ? p_max(b[d], p_subs(b[d - off], g[d - off])) : p_max(b[d], b[d - off])) : b[d];
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
                    const u32 sg = __builtin_amdgcn_alignbit(d < REAL ? g[d] : 0u, (d && d - 1 < REAL) ? g[d - 1] : 0u, 16);
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
                for (int d = 0; d < NW; d++) nb[d] = d >= off ? p_max(row[d], p_subs(row[d - off], p_add(kg, (d - off) < REAL ? g[d - off] : 0u))) : row[d];
#pragma unroll
                for (int d = 0; d < NW; d++) row[d] = nb[d];
            }
        }
#pragma unroll
        for (int d = 0; d < NW; d++) prev[d] = row[d];
#pragma unroll
        for (int d = 0; d < REAL; d++) gprev[d] = g[d];
    }
    // ---- max over every lane of the last row (ascii.rs:152-156) -------------------------------------
    u32 mx = prev[0];
#pragma unroll
    for (int d = 1; d < NW; d++) mx = p_max(mx, prev[d]);
    return max(mx & 0xFFFF, mx >> 16);
}

// exact flag (matcher/algo.rs:245-248): window spans the whole haystack and equals the needle byte for byte
template <int NB>
__device__ __forceinline__ bool exact_match(const NeedleDev& nd, bool include_exact, u32 m, const u32 (&hb)[NB]) {
    bool exact = include_exact && m == (u32)nd.nbytes;
    if (exact) {
        const u32* raw = (const u32*)nd.raw;
#pragma unroll
        for (int k = 0; k < NB && k < FZB_MAX_NEEDLE_BYTES / 4; k++) exact = exact && (hb[k] == raw[k]);
    }
    return exact;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-chunk form: windows of SWL < m <= 1024 bytes, still one THREAD per haystack.  Chunks are processed left to
// right; for every needle row the reference keeps the previous chunk's final row and match mask (score_matrix /
// match_masks column, src/smith_waterman/matrix.rs) because (a) `shift_right_padded::<k>` refills the low k lanes from
// the previous chunk's top k lanes (ascii.rs:136-143) and (b) lane 0's diagonal reads S(i-1, prev chunk)[last].
// Only the top SWL/2 lanes can ever be shifted in, so per row the thread parks SWL/4 dwords of row values and SWL/4
// dwords of gap-open charges in a global scratch slab laid out [row][dword][thread] (coalesced across the wave).
// In the biased domain the adjacent lanes are simply positions -SWL/2..-1 of one longer vector (bias (p + SWL) * gex).
// ---------------------------------------------------------------------------------------------------------------
// ND: NeedleDev (by value: up to 63 rows) or NeedleLongDev (the same members, the rows' bytes behind pointers into the matcher's device blob:
// any needle the reference's overflow guard accepts - k2d_dp_long)
template <int SWL, bool BIAS, typename ND = NeedleDev>
__device__ __forceinline__ u32 dp_multi_chunk(const ND& nd, const u8* __restrict__ th, u32 m, bool include_prefix, const u8* cls,
                                              u32* __restrict__ scratch, u32 sstride, u32 sidx) {
    constexpr int NW = SWL / 2;
    constexpr int NB = SWL / 4;
    constexpr int HT = NW / 2;  // parked dwords per vector (top half)
    const u32 rows = (u32)nd.rows;
    const u32 ONE = 0x00010001u;
    const u32 Mv = splat16(nd.match_plus_mismatch), Xv = splat16(nd.mismatch), gexv = splat16(nd.gex), gopmv = splat16(nd.gopm);
    const u32 casev = splat16(nd.matching_case), capv = splat16(nd.capitalization), delimv = splat16(nd.delimiter);
    const u32 nchunks = (m + SWL - 1) / SWL;
    const bool u8class = nd.lane_mask == 0xFF;  // score values of the u8 class fit a byte (score_fits_in_u8)
    u32 maxs[NW];
#pragma unroll
    for (int d = 0; d < NW; d++) maxs[d] = 0;
    u32 clsw_carry = 0;  // class of the previous chunk's last lane, in the HIGH half (what alignbit shifts in)
#pragma unroll 1
    for (u32 ch = 0; ch < nchunks; ch++) {
        const u32 cbase = ch * SWL;
        u32 hw[NW], bonus[NW];
        {
            u32 clsw_prev = clsw_carry;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const u32 p = cbase + 4 * k;
                u32 w = 0;
                if (p < m) {
                    w = load_u32_unaligned(th, p);
                    const u32 rem = m - p;
                    if (rem < 4) w &= (1u << (8 * rem)) - 1;
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int d = 2 * k + h;
                    const u32 b0 = h ? (w >> 16) & 0xFF : w & 0xFF;
                    const u32 b1 = h ? w >> 24 : (w >> 8) & 0xFF;
                    hw[d] = b0 | (b1 << 16);
                    const u32 clsw = (u32)cls[b0] | ((u32)cls[b1] << 16);
                    const u32 sh = __builtin_amdgcn_alignbit(clsw, clsw_prev, 16);
                    const u32 cap01 = (clsw >> 1) & sh & ONE;
                    const u32 dl01 = (sh >> 2) & ~(clsw >> 2) & ONE;
                    bonus[d] = p_add(p_add(p_mul(dl01, delimv), p_mul(cap01, capv)), Mv);
                    clsw_prev = clsw;
                }
            }
            clsw_carry = clsw_prev;
            if (ch == 0 && include_prefix) bonus[0] = p_add(bonus[0], (u32)nd.prefix);
        }
        u32 prev[NW], gprev[NW];
#pragma unroll
        for (int d = 0; d < NW; d++) prev[d] = 0, gprev[d] = 0;
        u32 carry = 0;  // S(r-1, previous chunk) top dword (its high half is the last lane), unbiased
#pragma unroll 1
        for (u32 r = 0; r < rows; r++) {
            const u32 c = (((const u32*)nd.c)[r >> 2] >> (8 * (r & 3))) & 0xFF, f = (((const u32*)nd.f)[r >> 2] >> (8 * (r & 3))) & 0xFF;  // scalar loads
            const bool ci = c != f;
            const u32 orv = ci ? 0x00200020u : 0u;
            const u32 cmpv = splat16(ci ? (c | 0x20) : c);
            const u32 cv = splat16(c);
            // previous chunk's parked vectors for this row (zero for the first chunk)
            // Parked form (the slab traffic is what bounds this kernel): the gap-open charges are 0 or gop' per lane = ONE bit per
            // lane, one dword for the whole top half; row values of the u8 score class fit a byte, four lanes per dword.
            u32 arow[HT], ag[HT];
            u32* srow = scratch + (size_t)(r * NW) * sstride + sidx;
            if (ch) {
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) {
                        const u32 pk = srow[(size_t)t * sstride];
                        arow[2 * t] = __builtin_amdgcn_perm(0u, pk, 0x0c010c00u);
                        arow[2 * t + 1] = __builtin_amdgcn_perm(0u, pk, 0x0c030c02u);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) arow[t] = srow[(size_t)t * sstride];
                }
                const u32 bits = srow[(size_t)HT * sstride];
#pragma unroll
                for (int t = 0; t < HT; t++) ag[t] = p_mul(((bits >> (2 * t)) & 1u) | (((bits >> (2 * t + 1)) & 1u) << 16), gopmv);
            } else {
#pragma unroll
                for (int t = 0; t < HT; t++) arow[t] = 0u, ag[t] = 0u;
            }
            u32 row[NW], g[NW];
#pragma unroll
            for (int d = 0; d < NW; d++) {
                const u32 mm = p_subs(ONE, (hw[d] | orv) ^ cmpv);
                const u32 ex = ci ? p_subs(ONE, hw[d] ^ cv) : mm;
                const u32 sh = __builtin_amdgcn_alignbit(prev[d], d ? prev[d - 1] : carry, 16);
                u32 t = p_add(p_mul(mm, bonus[d]), sh);
                t = p_subs(t, Xv);
                const u32 diag = p_add(p_mul(ex, casev), t);
                const u32 up = p_subs(p_subs(prev[d], gexv), gprev[d]);
                row[d] = p_max(diag, up);
                g[d] = p_mul(mm, gopmv);
            }
            carry = arow[HT - 1];
            // propagate over the concatenation [parked top half of the previous chunk | this chunk]
            if (BIAS) {
                u32 b[NW], ab[HT];
#pragma unroll
                for (int d = 0; d < NW; d++) b[d] = p_add(row[d], (u32)nd.gex * (u32)((2 * d + SWL) + ((2 * d + 1 + SWL) << 16)));
#pragma unroll
                for (int t = 0; t < HT; t++) ab[t] = p_add(arow[t], (u32)nd.gex * (u32)((2 * (HT + t)) + ((2 * (HT + t) + 1) << 16)));
                {
                    u32 nb[NW];
#pragma unroll
                    for (int d = 0; d < NW; d++) {
                        const u32 sb = __builtin_amdgcn_alignbit(b[d], d ? b[d - 1] : ab[HT - 1], 16);
                        const u32 sg = __builtin_amdgcn_alignbit(g[d], d ? g[d - 1] : ag[HT - 1], 16);
                        nb[d] = p_max(b[d], p_subs(sb, sg));
                    }
#pragma unroll
                    for (int d = 0; d < NW; d++) b[d] = nb[d];
                }
#pragma unroll
                for (int off = 1; off < NW; off *= 2) {
                    u32 nb[NW];
#pragma unroll
                    for (int d = 0; d < NW; d++) {
                        const u32 sb = d >= off ? b[d - off] : ab[HT + d - off];
                        const u32 sg = d >= off ? g[d - off] : ag[HT + d - off];
                        nb[d] = p_max(b[d], p_subs(sb, sg));
                    }
#pragma unroll
                    for (int d = 0; d < NW; d++) b[d] = nb[d];
                }
#pragma unroll
                for (int d = 0; d < NW; d++) row[d] = p_sub(b[d], (u32)nd.gex * (u32)((2 * d + SWL) + ((2 * d + 1 + SWL) << 16)));
            } else {
                u32 kg = gexv;
                {
                    u32 nb[NW];
#pragma unroll
                    for (int d = 0; d < NW; d++) {
                        const u32 sb = __builtin_amdgcn_alignbit(row[d], d ? row[d - 1] : arow[HT - 1], 16);
                        const u32 sg = __builtin_amdgcn_alignbit(g[d], d ? g[d - 1] : ag[HT - 1], 16);
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
                    for (int d = 0; d < NW; d++) {
                        const u32 sb = d >= off ? row[d - off] : arow[HT + d - off];
                        const u32 sg = d >= off ? g[d - off] : ag[HT + d - off];
                        nb[d] = p_max(row[d], p_subs(sb, p_add(kg, sg)));
                    }
#pragma unroll
                    for (int d = 0; d < NW; d++) row[d] = nb[d];
                }
            }
            // park this chunk's top half for the next chunk
            if (ch + 1 < nchunks) {
                if (u8class) {
#pragma unroll
                    for (int t = 0; t < HT / 2; t++) srow[(size_t)t * sstride] = __builtin_amdgcn_perm(row[HT + 2 * t + 1], row[HT + 2 * t], 0x06040200u);
                } else {
#pragma unroll
                    for (int t = 0; t < HT; t++) srow[(size_t)t * sstride] = row[HT + t];
                }
                u32 bits = 0;
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    const u32 m01 = p_min(g[HT + t], ONE);  // 1 where the lane is charged
                    bits |= ((m01 & 1u) | ((m01 >> 15) & 2u)) << (2 * t);
                }
                srow[(size_t)HT * sstride] = bits;
            }
#pragma unroll
            for (int d = 0; d < NW; d++) prev[d] = row[d], gprev[d] = g[d];
        }
#pragma unroll
        for (int d = 0; d < NW; d++) maxs[d] = p_max(maxs[d], prev[d]);
    }
    u32 mx = maxs[0];
#pragma unroll
    for (int d = 1; d < NW; d++) mx = p_max(mx, maxs[d]);
    return max(mx & 0xFFFF, mx >> 16);
}
