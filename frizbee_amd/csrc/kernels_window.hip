// gfx950 kernels, stage 2a: the LANE-EXACT prefilter.  One thread per survivor of the streaming filter
// re-runs the reference's chunked multi-path algorithm with the same chunk width (PFL = 16/32/64 "lanes")
// the emulated CPU backend uses, because both the accept decision (with typos) and the returned window
// depend on it.  Chunk occurrence masks are 64-bit integers exactly like AVX-512's __mmask64
// (src/prefilter/backend/avx512.rs:9-57); movemask/tzcnt/lzcnt become SWAR compares + ffs/clz.
//
//   ASCII  1 / 2 / N typos : src/prefilter/algo/ascii_typos.rs:15-110, 113-251, 254-360, end scan :375-397
//   unicode 0 typos        : src/prefilter/algo/unicode.rs:119-219, back scan :222-276
//   unicode 1 / 2 / N typos: src/prefilter/algo/unicode_typos.rs:15-141, 144-330, 333-466, end scan :479-508
//
// Output per survivor: win[2j] = start, win[2j+1] = end (start = 0xFFFFFFFF if rejected) and a keep-bit
// (wave ballot) + per-tile keep counts for the second-level compaction.
#include <algorithm>
#include <type_traits>

#include "kernels_common.h"
#include "knobs.h"

template <int PFL>
struct Chunk {
    static constexpr int NW = PFL / 4;
    u32 w[NW];
};

template <int PFL>
__device__ __forceinline__ u64 m_all() { return PFL == 64 ? ~(u64)0 : (((u64)1 << PFL) - 1); }
template <int PFL>
__device__ __forceinline__ u64 m_first_n(u32 n) { return n >= (u32)PFL ? m_all<PFL>() : (((u64)1 << n) - 1); }
template <int PFL>
__device__ __forceinline__ u32 m_lz(u64 m) { return (u32)__builtin_clzll(m) - (64 - PFL); }  // m != 0
__device__ __forceinline__ u32 m_tz(u64 m) { return (u32)__builtin_ctzll(m); }                 // m != 0
template <int PFL>
__device__ __forceinline__ u64 m_clear_through_lowest(u64 self, u64 hit) { return self & ~(hit ^ (hit - 1)) & m_all<PFL>(); }

// PFL bytes of the haystack starting at byte `pos` (haystack-relative), zero beyond `len`.
template <int PFL>
__device__ __forceinline__ void load_chunk(Chunk<PFL>& c, const u8* __restrict__ hay, u32 pos, u32 len) {
    if ((pos & 15) == 0) {
        // `hay` is 16-byte aligned (padded-16 layout): whole-vector loads, zero beyond len (>= 80 readable bytes follow the corpus)
#pragma unroll
        for (int v = 0; v < Chunk<PFL>::NW / 4; v++) {
            const u32 p = pos + 16 * v;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (p < len) q = *(const uint4*)(hay + p);
            const u32 w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 pk = p + 4 * k;
                u32 x = w4[k];
                if (pk >= len) x = 0;
                else if (len - pk < 4) x &= (1u << (8 * (len - pk))) - 1;
                c.w[4 * v + k] = x;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < Chunk<PFL>::NW; k++) {
        const u32 p = pos + 4 * k;
        u32 v = 0;
        if (p < len) {
            v = load_u32_unaligned(hay, p);
            const u32 rem = len - p;
            if (rem < 4) v &= (1u << (8 * rem)) - 1;
        }
        c.w[k] = v;
    }
}

// 4-bit mask of the bytes of x that are zero (exact, no borrow artefacts)
__device__ __forceinline__ u32 zero_bytes4(u32 x) {
    u32 y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;  // bit7 set in every non-zero byte
    y = ~y & 0x80808080u;
    return ((y >> 7) * 0x00204081u >> 21) & 0xF;
}
template <int PFL>
__device__ __forceinline__ u64 eq_mask(const Chunk<PFL>& c, u32 v) {
    const u32 sp = v * 0x01010101u;
    u64 m = 0;
#pragma unroll
    for (int k = 0; k < Chunk<PFL>::NW; k++) m |= (u64)zero_bytes4(c.w[k] ^ sp) << (4 * k);
    return m;
}
template <int PFL>
__device__ __forceinline__ u64 occ_mask(const Chunk<PFL>& c, u32 a, u32 b) {
    if (b == a) return eq_mask<PFL>(c, a);
    // a / b are the two cases of one ASCII letter (they differ only in bit 5): (h | 0x20) == (a | 0x20)  <=>  h in {a, b}
    const u32 sp = (a | 0x20u) * 0x01010101u;
    u64 m = 0;
#pragma unroll
    for (int k = 0; k < Chunk<PFL>::NW; k++) m |= (u64)zero_bytes4((c.w[k] | 0x20202020u) ^ sp) << (4 * k);
    return m;
}

// ------------------------------------------------------------------------------------------------
// Occurrence sources: what `B::occ(chunk, needle[idx])` / `unicode_char_mask(start, .., needle[idx])` return
// ------------------------------------------------------------------------------------------------
template <int PFL, typename ND = NeedleDev>
struct AsciiSrc {
    static constexpr bool kUnicode = false;
    const ND& nd;
    const u8* hay;
    u32 len;
    Chunk<PFL> chunk;
    u64 valid;  // chunk_mask from load_window
    // The typo algorithms ask for the occurrence mask of a needle row again every time a path advances (~20 times for a
    // 6-row needle), each one a 16-dword SWAR compare plus two per-thread byte loads of the needle.  With a cache (LDS,
    // [row][thread of the workgroup], this thread's column only - no barriers) every row's mask is computed once per chunk, rows in a
    // wave-uniform loop (scalar needle loads), and a request is one ds_read_b64.
    u64* cache;
    u32 loaded;  // chunk start the cache / chunk belong to
    __device__ AsciiSrc(const ND& n, const u8* h, u32 l, u64* cache_ = nullptr) : nd(n), hay(h), len(l), cache(cache_), loaded(0xFFFFFFFFu) {}
    __device__ __forceinline__ void load(u32 start) {
        if (start == loaded) return;  // (the end scan of a one-chunk haystack asks for the chunk that is already here)
        loaded = start;
        load_chunk<PFL>(chunk, hay, start, len);
        valid = m_first_n<PFL>(len - start);
        if (cache)
            for (int r = 0; r < nd.rows; r++) cache[r * blockDim.x + threadIdx.x] = occ_mask<PFL>(chunk, nd.c[r], nd.f[r]);
    }
    __device__ __forceinline__ u64 mask(u32 idx) const { return cache ? cache[idx * blockDim.x + threadIdx.x] : occ_mask<PFL>(chunk, nd.c[idx], nd.f[idx]); }
    __device__ __forceinline__ u64 init_mask() const { return valid; }  // ASCII path masks start as chunk_mask
    __device__ __forceinline__ u32 rows() const { return (u32)nd.rows; }
};

template <int PFL, typename ND = NeedleDev>
struct UnicodeSrc {
    static constexpr bool kUnicode = true;
    const ND& nd;
    const u8* hay;
    u32 len;
    u32 start;
    Chunk<PFL> w0;  // window at start+0
    u32 guard;      // the 4 bytes after it: the windows at start+1 .. start+3 are byte shifts of (w0, guard), formed on demand
                    // (keeping four chunks and indexing them with a per-thread scalar length put the whole set into scratch memory)
    u64* cache;     // as AsciiSrc: [row][thread of the workgroup] in LDS, every row's mask computed once per chunk (a mask is 2 variants x up to
                    // 4 byte positions x 16 dwords of SWAR compares, and the typo algorithms ask for a row again every time a path advances)
    __device__ UnicodeSrc(const ND& n, const u8* h, u32 l, u64* cache_ = nullptr) : nd(n), hay(h), len(l), start(0), guard(0), cache(cache_) {}
    __device__ __forceinline__ void load(u32 s) {
        start = s;
        load_chunk<PFL>(w0, hay, s, len);
        guard = 0;
        const u32 p = s + PFL;
        if (p < len) {
            guard = load_u32_unaligned(hay, p);
            if (len - p < 4) guard &= (1u << (8 * (len - p))) - 1;
        }
        if (cache)
            for (int r = 0; r < nd.rows; r++) cache[r * blockDim.x + threadIdx.x] = compute_mask((u32)r);
    }
    // window at start + o, o in 0..3 (v_alignbyte_b32 takes the shift from a register)
    __device__ __forceinline__ Chunk<PFL> at(u32 o) const {
        Chunk<PFL> r;
#pragma unroll
        for (int k = 0; k < Chunk<PFL>::NW; k++) r.w[k] = __builtin_amdgcn_alignbyte(k + 1 < Chunk<PFL>::NW ? w0.w[k + 1] : guard, w0.w[k], o);
        return r;
    }
    // match_unicode_char_prefix + char_variant_mask (unicode.rs:9-72) for one case variant
    __device__ __forceinline__ u64 variant(const u8 chars[4], u32 cl, u64 chunk_mask) const {
        u64 m = eq_mask<PFL>(at(cl - 1), chars[cl - 1]) & chunk_mask;
        if (m != 0 && cl > 1) {
            m &= eq_mask<PFL>(at(0), chars[0]);
            if (cl > 2) m &= eq_mask<PFL>(at(1), chars[1]);
            if (cl > 3) m &= eq_mask<PFL>(at(2), chars[2]);
        }
        return m;
    }
    __device__ __forceinline__ u64 mask(u32 idx) const { return cache ? cache[idx * blockDim.x + threadIdx.x] : compute_mask(idx); }
    // unicode_char_mask (unicode.rs:74-117)
    __device__ __forceinline__ u64 compute_mask(u32 idx) const {
        const u32 cl = nd.ulen[idx];
        if (start + cl > len) return 0;
        const u64 valid = m_first_n<PFL>(len - (start + cl - 1));
        // (a scalar without a second case - every Arabic letter, digits, punctuation - has uf == uc: one variant, wave-uniform test)
        bool same = true;
        for (u32 k = 0; k < cl; k++) same = same && nd.uc[idx][k] == nd.uf[idx][k];
        const u64 m = variant(nd.uc[idx], cl, valid);
        return same ? m : (m | variant(nd.uf[idx], cl, valid));
    }
    __device__ __forceinline__ u64 init_mask() const { return m_all<PFL>(); }  // unicode path masks start as all()
    __device__ __forceinline__ u32 rows() const { return (u32)nd.rows; }
};

struct Win {
    bool matched;
    u32 start, end;
};

// ---- end scans --------------------------------------------------------------------------------
// find_end_pos_with_typos (ascii_typos.rs:375-397)
template <int PFL, typename S>
__device__ __forceinline__ u32 ascii_end_pos(S& src, u32 max_typos) {
    const u32 n = src.rows(), first = n - 1 - max_typos, len = src.len;
    u32 start = (len - 1) / PFL * PFL;
    for (;;) {
        src.load(start);
        u64 mask = 0;
        for (u32 i = first; i < n; i++) mask |= src.mask(i);
        mask &= src.valid;
        if (mask != 0) return start + PFL - m_lz<PFL>(mask);
        if (start == 0) break;
        start -= PFL;
    }
    return len;
}
// find_end_pos_with_unicode_typos (unicode_typos.rs:479-508)
template <int PFL, typename S>
__device__ __forceinline__ u32 unicode_end_pos(S& src, u32 max_typos) {
    const u32 n = src.rows(), first = n - 1 - max_typos, len = src.len;
    u32 start = len >= (u32)PFL ? len - PFL : 0;
    for (;;) {
        src.load(start);
        u32 end_pos = 0;
        for (u32 i = first; i < n; i++) {
            const u64 mask = src.mask(i);
            if (mask != 0) end_pos = max(end_pos, start + PFL - m_lz<PFL>(mask) + src.nd.ulen[i] - 1);
        }
        if (end_pos != 0) return end_pos;
        if (start == 0) break;
        start = start >= (u32)PFL ? start - PFL : 0;
    }
    return len;
}
template <int PFL, typename S>
__device__ __forceinline__ u32 end_pos(S& s, u32 k) {
    if constexpr (S::kUnicode) return unicode_end_pos<PFL>(s, k);
    else return ascii_end_pos<PFL>(s, k);
}

// ------------------------------------------------------------------------------------------------
// Occurrence masks computed AHEAD, by the whole workgroup (k2a_window's PRE form).  The typo algorithms walk a haystack chunk by chunk at
// the fixed starts 0, PFL, 2 PFL, ... and what costs is a chunk's occurrence masks (a unicode scalar: two case variants x up to four byte
// positions x 16 dwords of SWAR compares), not the walk.  One thread per haystack makes the longest haystack of a tile the tile's duration
// (Arabic-shaped list, 1 typo: ten chunks one after the other = 89 us of a 177 us step, while the median haystack has one).  In the PRE form
// the tile's (haystack, chunk) pairs are dealt out evenly to the workgroup's threads, which leave every needle row's mask in LDS
// ([pair][row]); the walk then only reads.  `Real` computes what was not laid out ahead: the end scan's unaligned windows (unicode) and
// the haystacks of a tile whose pairs did not fit the buffer.
// ------------------------------------------------------------------------------------------------
template <int PFL, typename Real>
struct PreSrc {
    static constexpr bool kUnicode = Real::kUnicode;
    Real real;
    const decltype(real.nd)& nd;
    const u8* hay;
    u32 len;
    const u64* buf;  // this haystack's masks, [chunk][row]
    u32 nch;         // chunks laid out ahead (0: none - every request goes to `real`)
    u32 cur;
    bool direct;
    u64 valid;
    template <typename ND>
    __device__ PreSrc(const ND& n, const u8* h, u32 l, const u64* b, u32 nch_) : real(n, h, l, nullptr), nd(real.nd), hay(h), len(l), buf(b), nch(nch_), cur(0), direct(true), valid(0) {}
    __device__ __forceinline__ void load(u32 start) {
        const u32 c = start / PFL;
        if (start % PFL == 0 && c < nch) {
            cur = c;
            direct = false;
            valid = m_first_n<PFL>(len - start);
        } else {
            real.load(start);
            direct = true;
            if constexpr (!kUnicode) valid = real.valid;
        }
    }
    __device__ __forceinline__ u64 mask(u32 idx) const { return direct ? real.mask(idx) : buf[(size_t)cur * (u32)nd.rows + idx]; }
    __device__ __forceinline__ u64 init_mask() const {
        if constexpr (kUnicode) return m_all<PFL>();
        else return valid;
    }
    __device__ __forceinline__ u32 rows() const { return (u32)nd.rows; }
};

// ---- 1 typo (ascii_typos.rs:15-110 / unicode_typos.rs:15-141) -----------------------------------
template <int PFL, typename Src>
__device__ __forceinline__ Win prefilter_1_typo(Src& src) {
    const u32 n = src.rows(), len = src.len;
    if (n <= 1) return {true, 0, len};
    if (len == 0) return {false, 0, 0};
    u32 i1 = 0, i2 = 1, msp = 0xFFFFFFFFu;
    for (u32 start = 0; start < len; start += PFL) {
        src.load(start);
        u64 m1 = src.mask(i1), m2 = src.mask(i2);
        u64 c1 = src.init_mask(), c2 = c1;
        for (;;) {
            bool advanced = false;
            const u32 cand = i1 + 1;
            if (cand > i2) {
                if (cand == n) return {true, msp, end_pos<PFL>(src, 1)};
                i2 = cand; c2 = c1; m2 = src.mask(i2);
            } else if (cand == i2 && c1 > c2) {
                c2 = c1;
            }
            const u64 h1 = m1 & c1;
            if (h1 != 0) {
                msp = min(msp, start + m_tz(h1));
                i1 += 1;
                c1 = m_clear_through_lowest<PFL>(c1, h1);
                m1 = src.mask(i1);
                advanced = true;
            }
            const u64 h2 = m2 & c2;
            if (h2 != 0) {
                msp = min(msp, start + m_tz(h2));
                i2 += 1;
                if (i2 >= n) return {true, msp, end_pos<PFL>(src, 1)};
                c2 = m_clear_through_lowest<PFL>(c2, h2);
                m2 = src.mask(i2);
                advanced = true;
            }
            if (!advanced) break;
        }
    }
    return {false, 0, len};
}

// ---- 2 typos (ascii_typos.rs:113-251 / unicode_typos.rs:144-330) --------------------------------
template <int PFL, typename Src>
__device__ __forceinline__ Win prefilter_2_typos(Src& src) {
    const u32 n = src.rows(), len = src.len;
    if (n <= 2) return {true, 0, len};
    if (len == 0) return {false, 0, 0};
    u32 i1 = 0, i2 = 1, i3 = 2, msp = 0xFFFFFFFFu;
    for (u32 start = 0; start < len; start += PFL) {
        src.load(start);
        u64 m1 = src.mask(i1), m2 = src.mask(i2), m3 = src.mask(i3);
        u64 c1 = src.init_mask(), c2 = c1, c3 = c1;
        for (;;) {
            bool advanced = false;
            const u32 cand2 = i1 + 1;
            if (cand2 > i2) {
                if (cand2 == n) return {true, msp, end_pos<PFL>(src, 2)};
                i2 = cand2; c2 = c1; m2 = src.mask(i2);
            } else if (cand2 == i2 && c1 > c2) {
                c2 = c1;
            }
            const u32 cand3 = i2 + 1;
            if (cand3 > i3) {
                if (cand3 == n) return {true, msp, end_pos<PFL>(src, 2)};
                i3 = cand3; c3 = c2; m3 = src.mask(i3);
            } else if (cand3 == i3 && c2 > c3) {
                c3 = c2;
            }
            const u64 h1 = m1 & c1;
            if (h1 != 0) {
                msp = min(msp, start + m_tz(h1));
                i1 += 1;
                c1 = m_clear_through_lowest<PFL>(c1, h1);
                m1 = src.mask(i1);
                advanced = true;
            }
            const u64 h2 = m2 & c2;
            if (h2 != 0) {
                msp = min(msp, start + m_tz(h2));
                i2 += 1;
                if (i2 >= n) return {true, msp, end_pos<PFL>(src, 2)};
                c2 = m_clear_through_lowest<PFL>(c2, h2);
                m2 = src.mask(i2);
                advanced = true;
            }
            const u64 h3 = m3 & c3;
            if (h3 != 0) {
                msp = min(msp, start + m_tz(h3));
                i3 += 1;
                if (i3 >= n) return {true, msp, end_pos<PFL>(src, 2)};
                c3 = m_clear_through_lowest<PFL>(c3, h3);
                m3 = src.mask(i3);
                advanced = true;
            }
            if (!advanced) break;
        }
    }
    return {false, 0, len};
}

// ---- N typos (ascii_typos.rs:254-360 / unicode_typos.rs:333-466) --------------------------------
// Per-path state (needle index + the occurrence mask of that needle row in the current chunk), max_typos + 1 paths.  Needles that fit
// NeedleDev (<= 63 rows) keep it in the thread's own arrays; long needles (NeedleLongDev, up to ~11 k paths) in a global slab laid out
// [path][thread].
struct LocalPaths {
    u8 idx_[FZB_MAX_ROWS + 1];
    u64 nmask_[FZB_MAX_ROWS + 1];
    __device__ __forceinline__ u32 idx(u32 p) const { return idx_[p]; }
    __device__ __forceinline__ void set_idx(u32 p, u32 v) { idx_[p] = (u8)v; }
    __device__ __forceinline__ u64 nmask(u32 p) const { return nmask_[p]; }
    __device__ __forceinline__ void set_nmask(u32 p, u64 v) { nmask_[p] = v; }
};
struct GlobalPaths {
    u32* idx_;     // [path][stride]
    u64* nmask_;   // [path][stride]
    u32 stride, slot;
    __device__ __forceinline__ u32 idx(u32 p) const { return idx_[(size_t)p * stride + slot]; }
    __device__ __forceinline__ void set_idx(u32 p, u32 v) { idx_[(size_t)p * stride + slot] = v; }
    __device__ __forceinline__ u64 nmask(u32 p) const { return nmask_[(size_t)p * stride + slot]; }
    __device__ __forceinline__ void set_nmask(u32 p, u64 v) { nmask_[(size_t)p * stride + slot] = v; }
};

template <int PFL, typename Src, typename Paths>
__device__ __forceinline__ Win prefilter_many_typos(Src& src, u32 max_typos, Paths& paths) {
    const u32 n = src.rows(), len = src.len;
    if (n <= max_typos) return {true, 0, len};
    if (len == 0) return {false, 0, 0};
    const u32 path_count = max_typos + 1;  // <= rows
    for (u32 p = 0; p < path_count; p++) paths.set_idx(p, 0);
    u32 msp = 0xFFFFFFFFu;
    for (u32 start = 0; start < len; start += PFL) {
        src.load(start);
        u64 chunk_mask = src.init_mask();
        for (u32 p = 0; p < path_count; p++) paths.set_nmask(p, src.mask(paths.idx(p)));
        for (;;) {
            for (u32 p = 1; p < path_count; p++) {
                const u32 cand = paths.idx(p - 1) + 1;
                if (cand > paths.idx(p)) {
                    if (cand == n) return {true, msp, end_pos<PFL>(src, max_typos)};
                    paths.set_idx(p, cand);
                    paths.set_nmask(p, src.mask(cand));
                }
            }
            u64 match_mask = 0;
            for (u32 p = 0; p < path_count; p++) match_mask |= paths.nmask(p);
            const u64 matches = match_mask & chunk_mask;
            if (matches == 0) break;
            const u32 hit_pos = m_tz(matches);
            const u64 hit = matches & m_first_n<PFL>(hit_pos + 1);
            msp = min(msp, start + hit_pos);
            for (u32 p = 0; p < path_count; p++) {
                if ((paths.nmask(p) & hit) == 0) continue;
                const u32 ni = paths.idx(p) + 1;
                paths.set_idx(p, ni);
                if (ni == n) return {true, msp, end_pos<PFL>(src, max_typos)};
                paths.set_nmask(p, src.mask(ni));
            }
            chunk_mask = m_clear_through_lowest<PFL>(chunk_mask, hit);
        }
    }
    return {false, 0, len};
}

// ---- ASCII 0 typos (src/prefilter/algo/ascii.rs:6-72), only launched for long needles: short ones are decided by the streaming DFA
// filter and their window comes from the scorer.  Chunk by chunk like the reference; its result is lane-free (accept <=> ordered
// subsequence; start = first occurrence of needle[0]; end = one past the last occurrence of needle[n-1]) -------------------------
template <int PFL, typename ND>
__device__ __forceinline__ Win prefilter_ascii_0(AsciiSrc<PFL, ND>& src) {
    const u32 n = src.rows(), len = src.len;
    if (len == 0) return {false, 0, 0};
    bool can_skip = true;
    u32 msp = 0, row = 0;
    for (u32 start = 0; start < len; start += PFL) {
        src.load(start);
        u64 chunk_mask = src.valid;
        for (;;) {
            const u64 mask = src.mask(row) & chunk_mask;
            if (mask == 0) break;
            chunk_mask = m_clear_through_lowest<PFL>(chunk_mask, mask);
            if (can_skip) { msp = start + m_tz(mask); can_skip = false; }
            if (row + 1 < n) { row++; continue; }
            if (start + PFL >= len) return {true, msp, start + PFL - m_lz<PFL>(mask)};
            // find_last_char_pos on haystack[start..] (ascii.rs:58-72)
            const u32 sub_len = len - start;
            u32 s2 = sub_len >= (u32)PFL ? sub_len - PFL : 0;
            for (;;) {
                // a window at an arbitrary offset of the sub-slice: load_chunk handles unaligned positions
                Chunk<PFL> ch;
                load_chunk<PFL>(ch, src.hay, start + s2, len);
                const u64 m2 = occ_mask<PFL>(ch, src.nd.c[n - 1], src.nd.f[n - 1]) & m_first_n<PFL>(sub_len - s2);
                if (m2 != 0) return {true, msp, start + s2 + PFL - m_lz<PFL>(m2)};
                s2 = s2 >= (u32)PFL ? s2 - PFL : 0;  // (always terminates: the last needle byte occurs in this sub-slice)
            }
        }
    }
    return {false, msp, len};
}

// ---- unicode 0 typos (unicode.rs:119-219) + back scan (unicode.rs:222-276) -----------------------
template <int PFL, typename ND>
__device__ __forceinline__ u32 find_last_unicode_char_pos(UnicodeSrc<PFL, ND>& src, u32 row, u32 sub_start) {
    // operates on haystack[sub_start..]; positions returned are relative to sub_start
    const ND& nd = src.nd;
    const u32 cl = nd.ulen[row];
    const u32 len = src.len - sub_start;
    const u32 back = PFL + cl - 1;
    u32 start = len >= back ? len - back : 0;
    UnicodeSrc<PFL, ND> sub(nd, src.hay + sub_start, len);
    for (;;) {
        sub.load(start);
        // load_window(start + cl - 1): valid lanes are those whose last byte lies inside the haystack
        const u64 valid = m_first_n<PFL>(len - (start + cl - 1));
        u64 mask = (eq_mask<PFL>(sub.at(cl - 1), nd.uc[row][cl - 1]) | eq_mask<PFL>(sub.at(cl - 1), nd.uf[row][cl - 1])) & valid;
        if (mask != 0 && cl > 1) {
            u64 pa = eq_mask<PFL>(sub.at(0), nd.uc[row][0]), pb = eq_mask<PFL>(sub.at(0), nd.uf[row][0]);
            if (cl > 2) { pa &= eq_mask<PFL>(sub.at(1), nd.uc[row][1]); pb &= eq_mask<PFL>(sub.at(1), nd.uf[row][1]); }
            if (cl > 3) { pa &= eq_mask<PFL>(sub.at(2), nd.uc[row][2]); pb &= eq_mask<PFL>(sub.at(2), nd.uf[row][2]); }
            mask &= (pa | pb);
        }
        if (mask != 0) return start + PFL - m_lz<PFL>(mask) + cl - 1;
        if (start == 0) break;
        start = start >= (u32)PFL ? start - PFL : 0;
    }
    return len;
}

template <int PFL, typename ND>
__device__ __forceinline__ Win prefilter_unicode_0(UnicodeSrc<PFL, ND>& src) {
    const ND& nd = src.nd;
    const u32 len = src.len, n = src.rows();
    if (len == 0) return {false, 0, 0};
    bool can_skip = true;
    u32 msp = 0;
    u32 row = 0;  // current needle scalar
    u32 start = 0;
    while (start + nd.ulen[row] <= len) {
        u32 char_len = nd.ulen[row];
        src.load(start);
        u64 valid = m_first_n<PFL>(len - (start + char_len - 1));
        u64 available = m_all<PFL>();
        for (;;) {
            const u64 chunk_mask = available & valid;
            // NOTE: the window (`chunk`) is the one loaded for `char_len`; the prefix compares use needle_char.len
            const u32 ncl = nd.ulen[row];
            u64 mask;
            {
                // char_variant_mask with chunk = window at start+char_len-1, prefixes for a scalar of ncl bytes
                u64 a = eq_mask<PFL>(src.at(char_len - 1), nd.uc[row][ncl - 1]) & chunk_mask;
                if (a != 0 && ncl > 1) {
                    a &= eq_mask<PFL>(src.at(0), nd.uc[row][0]);
                    if (ncl > 2) a &= eq_mask<PFL>(src.at(1), nd.uc[row][1]);
                    if (ncl > 3) a &= eq_mask<PFL>(src.at(2), nd.uc[row][2]);
                }
                u64 b = eq_mask<PFL>(src.at(char_len - 1), nd.uf[row][ncl - 1]) & chunk_mask;
                if (b != 0 && ncl > 1) {
                    b &= eq_mask<PFL>(src.at(0), nd.uf[row][0]);
                    if (ncl > 2) b &= eq_mask<PFL>(src.at(1), nd.uf[row][1]);
                    if (ncl > 3) b &= eq_mask<PFL>(src.at(2), nd.uf[row][2]);
                }
                mask = a | b;
            }
            if (mask == 0) break;
            available = m_clear_through_lowest<PFL>(available, mask);
            if (can_skip) { msp = start + m_tz(mask); can_skip = false; }
            if (row + 1 < n) {
                row += 1;
                if (nd.ulen[row] != char_len) {
                    if (start + nd.ulen[row] > len) break;
                    char_len = nd.ulen[row];
                    valid = m_first_n<PFL>(len - (start + char_len - 1));
                }
            } else if (start + nd.ulen[row] - 1 + PFL >= len) {
                return {true, msp, start + PFL - m_lz<PFL>(mask) + nd.ulen[row] - 1};
            } else {
                return {true, msp, start + find_last_unicode_char_pos<PFL>(src, row, start)};
            }
        }
        start += PFL;
    }
    return {false, 0, len};
}

// ------------------------------------------------------------------------------------------------
// K2a kernel.  One instantiation per (chunk width, algorithm) keeps each kernel small: the all-in-one
// version (7 algorithms inlined, 134 SGPR spills) was miscompiled by hipcc -O3 (ROCm 7.2) into an endless
// loop for PFL=64 / unicode / 2 typos while -O1 and the same source built for the host ran correctly.
// ------------------------------------------------------------------------------------------------
enum { ALG_ASCII_1 = 0, ALG_ASCII_2 = 1, ALG_ASCII_N = 2, ALG_UNI_0 = 3, ALG_UNI_1 = 4, ALG_UNI_2 = 5, ALG_UNI_N = 6, ALG_ASCII_0 = 7 };

// global path slab of the N-typo algorithms for long needles: (max_typos + 1) x threads entries each (nullptr for NeedleDev)
struct ManyScratch {
    u32* idx;
    u64* nmask;
};

// DECIDE form (typo configurations on the short-haystack path): the listed haystacks are the filter's MARGINAL survivors; nothing is
// written for the accepted ones (their window has a lane-free form that the scorer computes itself), a rejected one sets its bit in
// `rej.bits`, bumps its tile's count and the total.
// min_len: haystacks shorter than this are rejected first (`original_len >= self.min_haystack_len`, src/matcher/algo.rs:88) - the
// streaming filter does it on the usual path; for a long needle this kernel is the first stage.
// 256-thread workgroups, four passes per 1024-haystack tile (item lists, the decide form, long needles, the unicode 0-typo algorithm when its
// automaton does not fit; contiguous ranges of a typo query take k2a_window_pre below).
template <int PFL, int ALG, bool DECIDE = false, typename ND = NeedleDev>
__global__ __launch_bounds__(256) void k2a_window(const u8* __restrict__ bytes, const void* __restrict__ ends_v, int ends_u64, u64 first,
                                                  const u32* __restrict__ surv_idx, const u32* __restrict__ n_surv_ptr, const ND nd,
                                                  u32* __restrict__ win, u64* __restrict__ bitmap2, u32* __restrict__ tile_counts2, int use_cache, RejectOut rej, u32 min_len,
                                                  ManyScratch many) {
    extern __shared__ __attribute__((aligned(16))) u64 mask_cache[];  // rows x 256 occurrence masks (ASCII algorithms, use_cache)
    __shared__ u32 s_cnt;
    const u32 M = *n_surv_ptr;
    const u32 ntiles = (M + FZB_TILE - 1) / FZB_TILE;
    const int tid = threadIdx.x;
    constexpr int TPB = 256;
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        u32 cnt = 0;
#pragma unroll 1
        for (int p = 0; p < FZB_TILE / TPB; p++) {
            const u32 j = tile * FZB_TILE + p * TPB + tid;
            bool keep = false;
            if (j < M) {
                const u32 li = surv_idx ? surv_idx[j] : j;
                u64 s;
                u32 L;
                if (ends_u64) haystack_span((const u64*)ends_v, first + li, s, L);
                else haystack_span((const u32*)ends_v, first + li, s, L);
                const u8* hay = bytes + s;
                Win w = {false, 0, 0};
                if (L < min_len) {
                    // too short for this needle and typo budget
                } else if (ALG >= ALG_UNI_0 && ALG != ALG_ASCII_0) {
                    UnicodeSrc<PFL, ND> src(nd, hay, L, use_cache ? mask_cache : nullptr);
                    if (ALG == ALG_UNI_0) w = prefilter_unicode_0<PFL>(src);
                    else if (ALG == ALG_UNI_1) w = prefilter_1_typo<PFL>(src);
                    else if (ALG == ALG_UNI_2) w = prefilter_2_typos<PFL>(src);
                    else if (ND::kLong) {
                        GlobalPaths paths{many.idx, many.nmask, gridDim.x * (u32)TPB, blockIdx.x * (u32)TPB + (u32)tid};
                        w = prefilter_many_typos<PFL>(src, (u32)nd.max_typos, paths);
                    } else {
                        LocalPaths paths;
                        w = prefilter_many_typos<PFL>(src, (u32)nd.max_typos, paths);
                    }
                } else {
                    AsciiSrc<PFL, ND> src(nd, hay, L, use_cache ? mask_cache : nullptr);
                    if (ALG == ALG_ASCII_0) w = prefilter_ascii_0<PFL>(src);
                    else if (ALG == ALG_ASCII_1) w = prefilter_1_typo<PFL>(src);
                    else if (ALG == ALG_ASCII_2) w = prefilter_2_typos<PFL>(src);
                    else if (ND::kLong) {
                        GlobalPaths paths{many.idx, many.nmask, gridDim.x * (u32)TPB, blockIdx.x * (u32)TPB + (u32)tid};
                        w = prefilter_many_typos<PFL>(src, (u32)nd.max_typos, paths);
                    } else {
                        LocalPaths paths;
                        w = prefilter_many_typos<PFL>(src, (u32)nd.max_typos, paths);
                    }
                }
                keep = w.matched;
                if (DECIDE) {
                    if (!keep) {
                        atomicOr((unsigned long long*)&rej.bits[li >> 6], 1ull << (li & 63));
                        atomicAdd(&rej.tile_rejects[li / FZB_TILE], 1u);
                        atomicAdd(rej.count, 1u);
                    }
                } else {
                    win[2 * j] = keep ? w.start : 0xFFFFFFFFu;
                    win[2 * j + 1] = w.end;
                }
            }
            if (!DECIDE) {
                const u64 b = __ballot(keep);
                if (lane_id() == 0) {
                    bitmap2[(tile * FZB_TILE + p * TPB) / 64 + (tid >> 6)] = b;
                    cnt += __popcll(b);
                }
            }
        }
        if (DECIDE) continue;
        if (lane_id() == 0 && cnt) atomicAdd(&s_cnt, cnt);
        __syncthreads();
        if (tid == 0) tile_counts2[tile] = s_cnt;
        __syncthreads();
    }
}

// The PRE form of the kernel above for the typo algorithms over short needles (NeedleDev), a QUARTER of a 1024-haystack tile per 256-thread
// workgroup: (1) every thread notes its haystack's span and number of PFL-byte chunks, a workgroup scan gives each haystack its place in
// the mask buffer; (2) the tile's (haystack, chunk) pairs are dealt out round-robin - a thread finds a pair's owner by bisection over
// the scan - and every needle row's occurrence mask goes to LDS; (3) thread per haystack, the reference's walk over PreSrc.  Pairs beyond
// the buffer (cap_pairs) are not laid out: their haystacks compute on demand, as in the plain form.  Same results bit for bit (the masks
// are the same function of the same bytes).
template <int PFL, int ALG>
__global__ __launch_bounds__(256) void k2a_window_pre(const u8* __restrict__ bytes, const void* __restrict__ ends_v, int ends_u64, u64 first,
                                                      const u32* __restrict__ surv_idx, const u32* __restrict__ n_surv_ptr, const NeedleDev nd,
                                                      u32* __restrict__ win, u64* __restrict__ bitmap2, u32* __restrict__ tile_counts2, u32 cap_pairs) {
    // A workgroup owns a PART of a 1024-haystack tile (NQ = 4 parts per tile, one unit of work each) and ADDS its count to the tile's (the
    // launcher zeroes the counts).  Round 5 measured the one-tile-per-1024-thread-workgroup form on the Arabic-shaped 1-typo query (105 tiles:
    // 96 us for the stage): the masks laid out ahead are 10 us of it, the WALK 66 - sixteen waves of divergent per-haystack loops sharing four
    // SIMDs on 105 of the 256 CUs.  As quarter tiles the same walks run on every CU, one or two waves per SIMD (67 us).
    constexpr int TPB = 256;
    constexpr int NQ = FZB_TILE / TPB;
    constexpr bool UNI = ALG >= ALG_UNI_0 && ALG != ALG_ASCII_0;
    using Real = typename std::conditional<UNI, UnicodeSrc<PFL, NeedleDev>, AsciiSrc<PFL, NeedleDev>>::type;
    extern __shared__ __attribute__((aligned(16))) u64 mask_buf[];  // [pair][row]
    __shared__ u32 s_s16[TPB], s_len[TPB], s_base[TPB + 1], s_wsum[TPB / 64], s_cnt;
    const u32 M = *n_surv_ptr;
    const u32 nunits = ((M + FZB_TILE - 1) / FZB_TILE) * NQ;
    const int tid = threadIdx.x;
    const u32 rows = (u32)nd.rows;
    for (u32 unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const u32 tile = unit / NQ, part = unit % NQ;
        if (tid == 0) s_cnt = 0;
        const u32 j = tile * FZB_TILE + part * TPB + tid;
        u64 s = 0;
        u32 L = 0;
        if (j < M) {
            const u32 li = surv_idx ? surv_idx[j] : j;
            if (ends_u64) haystack_span((const u64*)ends_v, first + li, s, L);
            else haystack_span((const u32*)ends_v, first + li, s, L);
        }
        const u32 nch = (j < M) ? (L + PFL - 1) / PFL : 0;
        s_s16[tid] = (u32)(s >> 4);  // (padded-16 layout: every haystack starts on a 16-byte boundary; the launcher checks the corpus is below 64 GiB)
        s_len[tid] = L;
        // exclusive scan of nch over the workgroup: inclusive scan inside the wave, then the wave totals
        u32 inc = nch;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 up = __shfl_up(inc, d, 64);
            if (lane_id() >= d) inc += up;
        }
        if (lane_id() == 63) s_wsum[tid >> 6] = inc;
        __syncthreads();
        u32 wbase = 0;
        for (int w = 0; w < (tid >> 6); w++) wbase += s_wsum[w];
        const u32 base = wbase + inc - nch;
        s_base[tid] = base;
        if (tid == TPB - 1) s_base[TPB] = base + nch;
        __syncthreads();
        const u32 T = min(s_base[TPB], cap_pairs);
        for (u32 q = tid; q < T; q += TPB) {
            u32 lo = 0, hi = TPB - 1;  // the last t with s_base[t] <= q (haystacks without chunks share their successor's base and are skipped)
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (s_base[mid] <= q) lo = mid;
                else hi = mid - 1;
            }
            const u32 c = q - s_base[lo];
            Real r(nd, bytes + ((u64)s_s16[lo] << 4), s_len[lo], nullptr);
            r.load(c * PFL);
            for (u32 row = 0; row < rows; row++) mask_buf[(size_t)q * rows + row] = r.mask(row);
        }
        __syncthreads();
        bool keep = false;
        if (j < M) {
            const u32 ahead = base + nch <= T ? nch : 0;  // a haystack is laid out whole or not at all
            PreSrc<PFL, Real> src(nd, bytes + s, L, mask_buf + (size_t)base * rows, ahead);
            Win w;
            if (ALG == ALG_UNI_1 || ALG == ALG_ASCII_1) w = prefilter_1_typo<PFL>(src);
            else if (ALG == ALG_UNI_2 || ALG == ALG_ASCII_2) w = prefilter_2_typos<PFL>(src);
            else {
                LocalPaths paths;
                w = prefilter_many_typos<PFL>(src, (u32)nd.max_typos, paths);
            }
            keep = w.matched;
            win[2 * j] = keep ? w.start : 0xFFFFFFFFu;
            win[2 * j + 1] = w.end;
        }
        const u64 b = __ballot(keep);
        if (lane_id() == 0) {
            bitmap2[(tile * FZB_TILE + part * TPB) / 64 + (tid >> 6)] = b;
            if (b) atomicAdd(&s_cnt, (u32)__popcll(b));
        }
        __syncthreads();
        if (tid == 0 && s_cnt) atomicAdd(&tile_counts2[tile], s_cnt);
        __syncthreads();
    }
}

template <int PFL>
static void launch_window_pfl(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleDev& nd, u32* win, u64* bitmap2,
                              u32* tile_counts2, int grid, hipStream_t st, const RejectOut* decide, u32 max_items) {
    const int k = nd.max_typos;
    const int alg = nd.unicode ? (k == 0 ? ALG_UNI_0 : k == 1 ? ALG_UNI_1 : k == 2 ? ALG_UNI_2 : ALG_UNI_N) : (k == 1 ? ALG_ASCII_1 : k == 2 ? ALG_ASCII_2 : ALG_ASCII_N);
    // occurrence-mask cache in LDS of the plain form (ASCII and unicode algorithms): rows x 2 KB per 256-thread workgroup, up to 16 rows
    const int use_cache = nd.rows <= 16;
    const size_t lds = use_cache ? (size_t)nd.rows * 256 * 8 : 0;
    const ManyScratch none{nullptr, nullptr};
    if (decide) {  // ASCII typo algorithms only (the unicode path keeps the full form)
#define FZB_K2A_D(ALG) hipLaunchKernelGGL((k2a_window<PFL, ALG, true>), dim3(grid), dim3(256), lds, st, c.bytes, c.ends, c.ends_u64, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, use_cache, *decide, 0u, none)
        if (alg == ALG_ASCII_1) FZB_K2A_D(ALG_ASCII_1);
        else if (alg == ALG_ASCII_2) FZB_K2A_D(ALG_ASCII_2);
        else FZB_K2A_D(ALG_ASCII_N);
#undef FZB_K2A_D
        return;
    }
    // the PRE form (masks laid out ahead by the whole workgroup, quarter tiles, a grid-stride loop over the units: lists of any size): the typo
    // algorithms over a range whose size the host knows (the tile counts it adds to are zeroed here) in a corpus below 64 GiB
    if (max_items != 0 && alg != ALG_UNI_0 && c.total_bytes < ((u64)1 << 36)) {
        const u32 ntiles_max = (max_items + FZB_TILE - 1) / FZB_TILE;
        (void)hipMemsetAsync(tile_counts2, 0, (size_t)ntiles_max * 4, st);
        const size_t dyn = (size_t)40 * 1024;
#define FZB_K2A_P(ALG) hipLaunchKernelGGL((k2a_window_pre<PFL, ALG>), dim3(std::max<u32>(1u, std::min<u32>(ntiles_max * 4u, (u32)grid * 8u))), dim3(256), dyn, st, c.bytes, c.ends, c.ends_u64, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, (u32)(dyn / 8 / (size_t)std::max(nd.rows, 1)))
        switch (alg) {
            case ALG_ASCII_1: FZB_K2A_P(ALG_ASCII_1); break;
            case ALG_ASCII_2: FZB_K2A_P(ALG_ASCII_2); break;
            case ALG_ASCII_N: FZB_K2A_P(ALG_ASCII_N); break;
            case ALG_UNI_1: FZB_K2A_P(ALG_UNI_1); break;
            case ALG_UNI_2: FZB_K2A_P(ALG_UNI_2); break;
            default: FZB_K2A_P(ALG_UNI_N); break;
        }
#undef FZB_K2A_P
        return;
    }
#define FZB_K2A(ALG) hipLaunchKernelGGL((k2a_window<PFL, ALG>), dim3(grid), dim3(256), lds, st, c.bytes, c.ends, c.ends_u64, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, use_cache, RejectOut{}, 0u, none)
    switch (alg) {
        case ALG_ASCII_1: FZB_K2A(ALG_ASCII_1); break;
        case ALG_ASCII_2: FZB_K2A(ALG_ASCII_2); break;
        case ALG_ASCII_N: FZB_K2A(ALG_ASCII_N); break;
        case ALG_UNI_0: FZB_K2A(ALG_UNI_0); break;
        case ALG_UNI_1: FZB_K2A(ALG_UNI_1); break;
        case ALG_UNI_2: FZB_K2A(ALG_UNI_2); break;
        default: FZB_K2A(ALG_UNI_N); break;
    }
#undef FZB_K2A
}

void fzb_launch_window(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleDev& nd, int pf_lanes,
                       u32* win, u64* bitmap2, u32* tile_counts2, u32* counters, int grid, hipStream_t st, const RejectOut* decide, u32 max_items) {
    // max_items: an upper bound of *n_surv_ptr known on the host (the range's size; 0 = unknown: item lists)
    if (pf_lanes == 64) launch_window_pfl<64>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, grid, st, decide, max_items);
    else if (pf_lanes == 32) launch_window_pfl<32>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, grid, st, decide, max_items);
    else launch_window_pfl<16>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, grid, st, decide, max_items);
}

// ---- long needles: this kernel is the FIRST stage (length test + the reference's prefilter at the exact lane width, every typo
// budget, ASCII and unicode), over the whole range or an item list; needle arrays and N-typo path state in global memory ----------
size_t fzb_window_long_scratch_bytes(const NeedleLongDev& nd, int grid) {
    const int k = nd.max_typos;
    if (k < 3) return 0;
    return (size_t)(k + 1) * (size_t)grid * 256 * (sizeof(u32) + sizeof(u64));
}

template <int PFL>
static void launch_window_long_pfl(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleLongDev& nd, u32* win, u64* bitmap2, u32* tile_counts2,
                                   void* scratch, int grid, hipStream_t st) {
    const int k = nd.max_typos;
    const int alg = nd.unicode ? (k == 0 ? ALG_UNI_0 : k == 1 ? ALG_UNI_1 : k == 2 ? ALG_UNI_2 : ALG_UNI_N) : (k == 0 ? ALG_ASCII_0 : k == 1 ? ALG_ASCII_1 : k == 2 ? ALG_ASCII_2 : ALG_ASCII_N);
    ManyScratch many{nullptr, nullptr};
    if (k >= 3) {
        many.nmask = (u64*)scratch;  // 8-byte entries first (alignment), then the 4-byte ones
        many.idx = (u32*)((u8*)scratch + (size_t)(k + 1) * (size_t)grid * 256 * sizeof(u64));
    }
#define FZB_K2A_L(ALG) hipLaunchKernelGGL((k2a_window<PFL, ALG, false, NeedleLongDev>), dim3(grid), dim3(256), 0, st, c.bytes, c.ends, c.ends_u64, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, 0, RejectOut{}, (u32)nd.min_haystack_len, many)
    switch (alg) {
        case ALG_ASCII_0: FZB_K2A_L(ALG_ASCII_0); break;
        case ALG_ASCII_1: FZB_K2A_L(ALG_ASCII_1); break;
        case ALG_ASCII_2: FZB_K2A_L(ALG_ASCII_2); break;
        case ALG_ASCII_N: FZB_K2A_L(ALG_ASCII_N); break;
        case ALG_UNI_0: FZB_K2A_L(ALG_UNI_0); break;
        case ALG_UNI_1: FZB_K2A_L(ALG_UNI_1); break;
        case ALG_UNI_2: FZB_K2A_L(ALG_UNI_2); break;
        default: FZB_K2A_L(ALG_UNI_N); break;
    }
#undef FZB_K2A_L
}

void fzb_launch_window_long(const CorpusDev& c, u64 first, const u32* surv_idx, const u32* n_surv_ptr, const NeedleLongDev& nd, int pf_lanes, u32* win, u64* bitmap2,
                            u32* tile_counts2, void* scratch, int grid, hipStream_t st) {
    if (pf_lanes == 64) launch_window_long_pfl<64>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, scratch, grid, st);
    else if (pf_lanes == 32) launch_window_long_pfl<32>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, scratch, grid, st);
    else launch_window_long_pfl<16>(c, first, surv_idx, n_surv_ptr, nd, win, bitmap2, tile_counts2, scratch, grid, st);
}
