// ORACLE - TEST INFRASTRUCTURE ONLY. Not part of the shipped product.
//
// A plain-CPU restatement of saghen/frizbee v0.12.0's `Matcher::match_list` hot path
// (prefilter -> trim -> Smith-Waterman -> Match -> radix sort), written to be read
// side by side with the Rust sources it follows.  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may include/link/execute anything in oracle/.
//
// The reference is a Rust crate and there is no Rust toolchain in the build image, so it
// cannot be compiled into oracle/_ref.  The restatement is instead pinned by the
// reference's own known-answer tests (tests/golden/*.json, harvested from
// src/smith_waterman/mod.rs, src/prefilter/mod.rs, src/matcher/*.rs, tests/api_properties.rs)
// and by its property tests re-run here on its own input generators
// (tests/test_oracle_reference_properties.py: prefilter == LCS criterion with identical
// windows at every lane width, cross-width score / position identities, the public-API contract).
// Parity with the real AVX-512 binary on large random inputs is NOT pinned.
//
// Every function cites the reference file:line it restates.  The SIMD-generic Rust code is
// `impl<B: Backend>`; here `LANES` (and the score lane type `T` = uint8_t / uint16_t) are
// template parameters, exactly like the reference's const-generic scalar backends
// (src/smith_waterman/backend/scalar.rs:10-16, 359-497; src/prefilter/backend/scalar.rs).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#if defined(FZO_NATIVE_SIMD) && defined(__AVX512BW__) && defined(__AVX512VL__) && defined(__AVX512VBMI__)
#define FZO_AVX512 1
#include <immintrin.h>
#endif

namespace fzo {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

// ---------------------------------------------------------------------------------------
// Public types: src/lib.rs:141-153 (Match), :236-258 (Config), :311-326 (SortStrategy),
// :357-377 (CaseMatching), :379-401 (UnicodeMatching), :439-478 (Scoring), src/const.rs:1-10
// ---------------------------------------------------------------------------------------
struct Scoring {
    u16 match_score = 12, mismatch_penalty = 6, gap_open_penalty = 5, gap_extend_penalty = 1;
    u16 prefix_bonus = 12, capitalization_bonus = 4, matching_case_bonus = 4, exact_match_bonus = 8,
        delimiter_bonus = 4;
};
enum CaseMatching { CASE_IGNORE = 0, CASE_SMART = 1, CASE_RESPECT = 2 };
enum UnicodeMatching { UNI_IGNORE = 0, UNI_SMART = 1, UNI_ALWAYS = 2 };
enum SortStrategy { SORT_SCORE_THEN_INDEX_ASC = 0, SORT_SCORE_THEN_INDEX_DESC = 1, SORT_INDEX_ASC = 2, SORT_INDEX_DESC = 3 };

enum Matching { MATCH_FUZZY = 0, MATCH_EXACT = 1, MATCH_PREFIX = 2, MATCH_SUFFIX = 3, MATCH_SUBSTRING = 4 };  // lib.rs:414-427
struct Config {
    int max_typos = 0;  // -1 == None
    int casing = CASE_SMART;
    int unicode = UNI_SMART;
    int sort = SORT_SCORE_THEN_INDEX_ASC;
    Scoring scoring;
    int matching = MATCH_FUZZY;
};

struct Match {
    u32 index;
    u16 score;
    u8 exact;
    u8 _pad;
};

inline bool sort_is_reversed(int s) { return s == SORT_INDEX_DESC || s == SORT_SCORE_THEN_INDEX_DESC; }  // lib.rs:343-348
inline bool sort_is_by_score(int s) { return s == SORT_SCORE_THEN_INDEX_ASC || s == SORT_SCORE_THEN_INDEX_DESC; }  // lib.rs:351-356

// ---------------------------------------------------------------------------------------
// UTF-8 / Unicode case helpers (stand-ins for Rust's char::{is_uppercase,to_lowercase,...})
// ---------------------------------------------------------------------------------------
#include "unicode_case_table.inc"

inline int utf8_len_of_cp(u32 cp) { return cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4; }
inline int utf8_encode(u32 cp, u8 out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (cp < 0x80) { out[0] = (u8)cp; return 1; }
    if (cp < 0x800) { out[0] = (u8)(0xC0 | (cp >> 6)); out[1] = (u8)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) { out[0] = (u8)(0xE0 | (cp >> 12)); out[1] = (u8)(0x80 | ((cp >> 6) & 0x3F)); out[2] = (u8)(0x80 | (cp & 0x3F)); return 3; }
    out[0] = (u8)(0xF0 | (cp >> 18)); out[1] = (u8)(0x80 | ((cp >> 12) & 0x3F)); out[2] = (u8)(0x80 | ((cp >> 6) & 0x3F)); out[3] = (u8)(0x80 | (cp & 0x3F));
    return 4;
}
// Decode valid UTF-8 (a Rust &str is always valid) into scalars.
inline std::vector<u32> utf8_decode(const u8* s, size_t n) {
    std::vector<u32> out;
    size_t i = 0;
    while (i < n) {
        u8 b = s[i];
        u32 cp; int len;
        if (b < 0x80) { cp = b; len = 1; }
        else if ((b & 0xE0) == 0xC0) { cp = b & 0x1F; len = 2; }
        else if ((b & 0xF0) == 0xE0) { cp = b & 0x0F; len = 3; }
        else if ((b & 0xF8) == 0xF0) { cp = b & 0x07; len = 4; }
        else throw std::invalid_argument("needle is not valid UTF-8");
        if (i + len > n) throw std::invalid_argument("needle is not valid UTF-8");
        for (int k = 1; k < len; k++) {
            if ((s[i + k] & 0xC0) != 0x80) throw std::invalid_argument("needle is not valid UTF-8");
            cp = (cp << 6) | (s[i + k] & 0x3F);
        }
        out.push_back(cp);
        i += len;
    }
    return out;
}
inline bool cp_is_uppercase(u32 cp) {
    if (cp < 0x80) return cp >= 'A' && cp <= 'Z';
    size_t lo = 0, hi = FZB_UPPER_RANGES_LEN;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (cp < FZB_UPPER_RANGES[mid][0]) hi = mid;
        else if (cp > FZB_UPPER_RANGES[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
// Single-scalar same-width case flip, or cp itself (prefilter/mod.rs:71-96)
inline u32 cp_flip_same_width(u32 cp) {
    if (cp < 0x80) {
        if (cp >= 'A' && cp <= 'Z') return cp + 32;
        if (cp >= 'a' && cp <= 'z') return cp - 32;
        return cp;
    }
    size_t lo = 0, hi = FZB_CASE_FLIP_LEN;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (FZB_CASE_FLIP[mid][0] < cp) lo = mid + 1; else hi = mid;
    }
    if (lo < FZB_CASE_FLIP_LEN && FZB_CASE_FLIP[lo][0] == cp) return FZB_CASE_FLIP[lo][1];
    return cp;
}
inline bool is_ascii(const std::string& s) { for (unsigned char c : s) if (c >= 0x80) return false; return true; }

// lib.rs:370-376
inline bool respects_case_for(int casing, const std::string& needle) {
    switch (casing) {
        case CASE_IGNORE: return false;
        case CASE_RESPECT: return true;
        default: {
            for (u32 cp : utf8_decode((const u8*)needle.data(), needle.size())) if (cp_is_uppercase(cp)) return true;
            return false;
        }
    }
}
// lib.rs:394-400
inline bool respects_unicode_for(int unicode, const std::string& needle) {
    switch (unicode) {
        case UNI_IGNORE: return false;
        case UNI_ALWAYS: return true;
        default: return !is_ascii(needle);
    }
}

// prefilter/mod.rs:21-47
struct UnicodeChar { u8 chars[4]; u8 flipped[4]; int len; };

// prefilter/mod.rs:49-65
inline std::vector<std::pair<u8, u8>> case_needle(const std::string& needle, bool case_sensitive) {
    std::vector<std::pair<u8, u8>> out;
    for (unsigned char c : needle) {
        u8 f;
        if (case_sensitive) f = c;
        else if (c >= 'a' && c <= 'z') f = (u8)(c - 32);
        else if (c >= 'A' && c <= 'Z') f = (u8)(c + 32);
        else f = c;
        out.push_back({c, f});
    }
    return out;
}
// prefilter/mod.rs:71-96
inline std::vector<UnicodeChar> case_needle_unicode(const std::string& needle, bool case_sensitive) {
    std::vector<UnicodeChar> out;
    for (u32 cp : utf8_decode((const u8*)needle.data(), needle.size())) {
        UnicodeChar uc;
        uc.len = utf8_encode(cp, uc.chars);
        u32 f = case_sensitive ? cp : cp_flip_same_width(cp);
        utf8_encode(f, uc.flipped);
        out.push_back(uc);
    }
    return out;
}

// ---------------------------------------------------------------------------------------
// Scoring guards: lib.rs:480-538, smith_waterman/mod.rs:92-116
// ---------------------------------------------------------------------------------------
inline u16 sat_add16(u16 a, u16 b) { u32 s = (u32)a + b; return s > 0xFFFF ? 0xFFFF : (u16)s; }
inline u16 sat_sub16(u16 a, u16 b) { return a > b ? (u16)(a - b) : 0; }
inline u16 max_per_char_bonus(const Scoring& s) {  // lib.rs:488-494
    u16 bonus = std::max(s.delimiter_bonus, s.capitalization_bonus);
    u16 amortized = std::max<u16>((u16)((bonus + 1) / 2), sat_sub16(bonus, s.gap_open_penalty));
    return sat_add16(amortized, s.matching_case_bonus);
}
inline u16 max_one_time_bonus(const Scoring& s) {  // lib.rs:497-503
    u16 bonus = std::max(s.delimiter_bonus, s.capitalization_bonus);
    u16 amortized = std::max<u16>((u16)((bonus + 1) / 2), sat_sub16(bonus, s.gap_open_penalty));
    return (u16)(bonus - amortized);
}
// Returns "" if OK, else the panic message (lib.rs:506-537)
inline std::string guard_against_score_overflow(const Scoring& s, size_t needle_len, int max_bonus_per_char_in = -1, int max_one_time_in = -1) {
    // fuzzy callers pass the amortised bonuses (matcher/algo.rs:311-325), the literal matcher its own (literal/algo.rs:314-322)
    const u16 bonus_pc = max_bonus_per_char_in < 0 ? max_per_char_bonus(s) : (u16)max_bonus_per_char_in;
    const u16 one_time = max_one_time_in < 0 ? max_one_time_bonus(s) : (u16)max_one_time_in;
    u16 max_per_char = sat_add16(s.match_score, bonus_pc);
    if (max_per_char == 0) return "";
    u16 headroom = sat_sub16(sat_sub16(sat_sub16(sat_sub16(0xFFFF, s.prefix_bonus), s.exact_match_bonus), s.mismatch_penalty), one_time);
    u16 max_needle_len = (u16)(headroom / max_per_char);
    if (needle_len > (size_t)max_needle_len)
        return "needle too long and could overflow the u16 score: " + std::to_string(needle_len) + " > " + std::to_string(max_needle_len);
    size_t max_gap_penalty = 32 * (size_t)s.gap_extend_penalty + (size_t)s.gap_open_penalty;
    if (max_gap_penalty > 0xFFFF)
        return "gap penalties too large and could overflow the u16 score: " + std::to_string(max_gap_penalty) + " > 65535";
    return "";
}
inline bool score_fits_in_u8(size_t needle_len, const Scoring& s) {  // smith_waterman/mod.rs:92-116
    size_t max_constant = std::max<size_t>({(size_t)s.match_score + (size_t)s.mismatch_penalty, s.gap_open_penalty, s.gap_extend_penalty,
                                            s.matching_case_bonus, s.capitalization_bonus, s.delimiter_bonus, s.prefix_bonus});
    if (max_constant > 255) return false;
    size_t max_gap_penalty = 64 * (size_t)s.gap_extend_penalty + (size_t)s.gap_open_penalty;
    if (max_gap_penalty > 255) return false;
    size_t max_per_char = (size_t)s.match_score + (size_t)max_per_char_bonus(s);
    size_t max_matrix_score = max_per_char * needle_len + (size_t)max_one_time_bonus(s) + (size_t)s.prefix_bonus;
    return max_matrix_score + (size_t)s.mismatch_penalty <= 255;
}

// =======================================================================================
// PREFILTER  (src/prefilter/algo/*.rs over the generic Backend, src/prefilter/backend/mod.rs)
// =======================================================================================
struct Window { bool matched; size_t start; size_t end; };

// One LANES-byte load of the prefilter with its compare-to-bitmask ops (the `Vector` side of
// prefilter/backend/mod.rs:116-172).  Lanes at or past `len` read as zero.
template <int LANES>
struct PfChunk {
    u8 b[LANES];
    static PfChunk load(const u8* h, size_t start, size_t len) {
        PfChunk c;
        for (int i = 0; i < LANES; i++) c.b[i] = (start + i < len) ? h[start + i] : 0;
        return c;
    }
    u64 eq(u8 v) const { u64 m = 0; for (int i = 0; i < LANES; i++) if (b[i] == v) m |= (u64)1 << i; return m; }
    u64 occ(std::pair<u8, u8> n) const { u64 m = 0; for (int i = 0; i < LANES; i++) if (b[i] == n.first || b[i] == n.second) m |= (u64)1 << i; return m; }
};
#if defined(FZO_AVX512)
// The same three ops as the reference's AVX-512 backend computes them (prefilter/backend/avx512.rs): one masked
// 64-byte load, byte compares straight into a 64-bit mask register.  Only compiled for the CPU-baseline build of the
// oracle (oracle/Makefile NATIVE=1 on a CPU with AVX-512 BW+VBMI); tests/test_oracle_avx512.py checks it lane for
// lane against the portable definitions above.
template <>
struct PfChunk<64> {
    __m512i x;
    static PfChunk load(const u8* h, size_t start, size_t len) {
        PfChunk c;
        const size_t take = len > start ? std::min<size_t>(len - start, 64) : 0;
        const __mmask64 k = take >= 64 ? ~(__mmask64)0 : (((__mmask64)1 << take) - 1);
        c.x = _mm512_maskz_loadu_epi8(k, h + start);
        return c;
    }
    u64 eq(u8 v) const { return (u64)_mm512_cmpeq_epi8_mask(x, _mm512_set1_epi8((char)v)); }
    u64 occ(std::pair<u8, u8> n) const {
        return (u64)(_mm512_cmpeq_epi8_mask(x, _mm512_set1_epi8((char)n.first)) | _mm512_cmpeq_epi8_mask(x, _mm512_set1_epi8((char)n.second)));
    }
};
#endif

template <int LANES>
struct Prefilter {
    static_assert(LANES == 16 || LANES == 32 || LANES == 64, "prefilter LANES");
    typedef u64 Mask;  // low LANES bits meaningful (u16/u32/u64 in the reference)
    typedef PfChunk<LANES> Chunk;

    std::vector<std::pair<u8, u8>> needle_ascii;
    std::vector<UnicodeChar> needle_unicode;

    Prefilter(const std::string& needle, bool case_sensitive)  // prefilter/algo/mod.rs:30-42
        : needle_ascii(case_needle(needle, case_sensitive)), needle_unicode(case_needle_unicode(needle, case_sensitive)) {}

    // ---- BitMaskOps (prefilter/backend/mod.rs:45-114) on a LANES-bit integer ----
    static Mask m_all() { return LANES == 64 ? ~(u64)0 : (((u64)1 << LANES) - 1); }
    static Mask m_first_n(size_t n) { return n >= (size_t)LANES ? m_all() : (((u64)1 << n) - 1); }
    static size_t m_tz(Mask m) { return m == 0 ? (size_t)LANES : (size_t)__builtin_ctzll(m); }
    static size_t m_lz(Mask m) { return m == 0 ? (size_t)LANES : (size_t)(__builtin_clzll(m) - (64 - LANES)); }
    static Mask m_clear_through_lowest(Mask self, Mask hit) { return self & ~(hit ^ ((hit - 1) & m_all())) & m_all(); }

    // ---- loads (prefilter/algo/load.rs:4-49).  Over-read bytes are never observable
    // (every use is ANDed with a validity mask), so lanes past `len` are zero here. ----
    static Chunk load_raw(const u8* h, size_t start, size_t len) { return Chunk::load(h, start, len); }
    static Chunk load_window(const u8* h, size_t start, size_t len, Mask& mask) {
        size_t remaining = len - start;
        mask = remaining >= (size_t)LANES ? m_all() : m_first_n(remaining);
        return load_raw(h, start, len);
    }
    static Mask eq(const Chunk& c, u8 v) { return c.eq(v); }
    static Mask occ(const Chunk& c, std::pair<u8, u8> n) { return c.occ(n); }

    // ---- 0 typos, ASCII: prefilter/algo/ascii.rs:6-54 ----
    Window match_haystack(const u8* h, size_t len) const {
        if (len == 0) return {false, 0, 0};
        bool can_skip_chunks = true;
        size_t match_start_pos = 0;
        size_t needle_i = 0;
        std::pair<u8, u8> needle_char = needle_ascii[needle_i++];
        size_t start = 0;
        while (start < len) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            for (;;) {
                Mask mask = occ(chunk, needle_char) & chunk_mask;
                if (mask == 0) break;
                chunk_mask = m_clear_through_lowest(chunk_mask, mask);
                if (can_skip_chunks) { match_start_pos = start + m_tz(mask); can_skip_chunks = false; }
                if (needle_i < needle_ascii.size()) {
                    needle_char = needle_ascii[needle_i++];
                } else if (start + LANES >= len) {
                    return {true, match_start_pos, start + LANES - m_lz(mask)};
                } else {
                    size_t end_pos = start + find_last_char_pos(needle_ascii.back(), h + start, len - start);
                    return {true, match_start_pos, end_pos};
                }
            }
            start += LANES;
        }
        return {false, match_start_pos, len};
    }
    // prefilter/algo/ascii.rs:58-72
    static size_t find_last_char_pos(std::pair<u8, u8> needle, const u8* h, size_t len) {
        size_t start = len >= (size_t)LANES ? len - LANES : 0;
        for (;;) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            Mask mask = occ(chunk, needle) & chunk_mask;
            if (mask != 0) return start + LANES - m_lz(mask);
            start = start >= (size_t)LANES ? start - LANES : 0;
        }
    }

    // ---- 1 typo, ASCII: prefilter/algo/ascii_typos.rs:15-110 ----
    Window match_haystack_1_typo(const u8* h, size_t len) const {
        size_t needle_len = needle_ascii.size();
        if (needle_len <= 1) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t first_idx = 0, second_idx = 1;
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            Mask first_mask = occ(chunk, needle_ascii[first_idx]);
            Mask second_mask = occ(chunk, needle_ascii[second_idx]);
            Mask first_cm = chunk_mask, second_cm = chunk_mask;
            for (;;) {
                bool advanced = false;
                size_t cand = first_idx + 1;
                if (cand > second_idx) {
                    if (cand == needle_len) return found_with_typos(h, len, match_start_pos, 1);
                    second_idx = cand;
                    second_cm = first_cm;
                    second_mask = occ(chunk, needle_ascii[second_idx]);
                } else if (cand == second_idx && first_cm > second_cm) {
                    second_cm = first_cm;
                }
                Mask fm = first_mask & first_cm;
                if (fm != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(fm));
                    first_idx += 1;
                    first_cm = m_clear_through_lowest(first_cm, fm);
                    first_mask = occ(chunk, needle_ascii[first_idx]);  // first_idx < needle_len: see note in found paths
                    advanced = true;
                }
                Mask sm = second_mask & second_cm;
                if (sm != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(sm));
                    second_idx += 1;
                    if (second_idx >= needle_len) return found_with_typos(h, len, match_start_pos, 1);
                    second_cm = m_clear_through_lowest(second_cm, sm);
                    second_mask = occ(chunk, needle_ascii[second_idx]);
                    advanced = true;
                }
                if (!advanced) break;
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }

    // ---- 2 typos, ASCII: prefilter/algo/ascii_typos.rs:113-251 ----
    Window match_haystack_2_typos(const u8* h, size_t len) const {
        size_t needle_len = needle_ascii.size();
        if (needle_len <= 2) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t i1 = 0, i2 = 1, i3 = 2;
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            Mask m1 = occ(chunk, needle_ascii[i1]), m2 = occ(chunk, needle_ascii[i2]), m3 = occ(chunk, needle_ascii[i3]);
            Mask c1 = chunk_mask, c2 = chunk_mask, c3 = chunk_mask;
            for (;;) {
                bool advanced = false;
                size_t cand2 = i1 + 1;
                if (cand2 > i2) {
                    if (cand2 == needle_len) return found_with_typos(h, len, match_start_pos, 2);
                    i2 = cand2; c2 = c1; m2 = occ(chunk, needle_ascii[i2]);
                } else if (cand2 == i2 && c1 > c2) {
                    c2 = c1;
                }
                size_t cand3 = i2 + 1;
                if (cand3 > i3) {
                    if (cand3 == needle_len) return found_with_typos(h, len, match_start_pos, 2);
                    i3 = cand3; c3 = c2; m3 = occ(chunk, needle_ascii[i3]);
                } else if (cand3 == i3 && c2 > c3) {
                    c3 = c2;
                }
                Mask h1 = m1 & c1;
                if (h1 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h1));
                    i1 += 1;
                    c1 = m_clear_through_lowest(c1, h1);
                    m1 = occ(chunk, needle_ascii[i1]);
                    advanced = true;
                }
                Mask h2 = m2 & c2;
                if (h2 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h2));
                    i2 += 1;
                    if (i2 >= needle_len) return found_with_typos(h, len, match_start_pos, 2);
                    c2 = m_clear_through_lowest(c2, h2);
                    m2 = occ(chunk, needle_ascii[i2]);
                    advanced = true;
                }
                Mask h3 = m3 & c3;
                if (h3 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h3));
                    i3 += 1;
                    if (i3 >= needle_len) return found_with_typos(h, len, match_start_pos, 2);
                    c3 = m_clear_through_lowest(c3, h3);
                    m3 = occ(chunk, needle_ascii[i3]);
                    advanced = true;
                }
                if (!advanced) break;
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }

    // ---- N typos, ASCII: prefilter/algo/ascii_typos.rs:254-360 ----
    Window match_haystack_many_typos(const u8* h, size_t len, size_t max_typos) const {
        size_t needle_len = needle_ascii.size();
        if (needle_len <= max_typos) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t path_count = max_typos + 1;
        std::vector<size_t> idx(path_count, 0);
        std::vector<Mask> nmask(path_count, 0);
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            for (size_t p = 0; p < path_count; p++) nmask[p] = occ(chunk, needle_ascii[idx[p]]);
            for (;;) {
                for (size_t p = 1; p < path_count; p++) {
                    size_t cand = idx[p - 1] + 1;
                    if (cand > idx[p]) {
                        if (cand == needle_len) return found_with_typos(h, len, match_start_pos, max_typos);
                        idx[p] = cand;
                        nmask[p] = occ(chunk, needle_ascii[cand]);
                    }
                }
                Mask match_mask = 0;
                for (size_t p = 0; p < path_count; p++) match_mask |= nmask[p];
                Mask matches = match_mask & chunk_mask;
                if (matches == 0) break;
                size_t hit_pos = m_tz(matches);
                Mask hit = matches & m_first_n(hit_pos + 1);
                match_start_pos = std::min(match_start_pos, start + hit_pos);
                for (size_t p = 0; p < path_count; p++) {
                    if ((nmask[p] & hit) == 0) continue;
                    idx[p] += 1;
                    if (idx[p] == needle_len) return found_with_typos(h, len, match_start_pos, max_typos);
                    nmask[p] = occ(chunk, needle_ascii[idx[p]]);
                }
                chunk_mask = m_clear_through_lowest(chunk_mask, hit);
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }

    // prefilter/algo/ascii_typos.rs:363-397
    Window found_with_typos(const u8* h, size_t len, size_t match_start_pos, size_t max_typos) const {
        return {true, match_start_pos, find_end_pos_with_typos(h, len, max_typos)};
    }
    size_t find_end_pos_with_typos(const u8* h, size_t len, size_t max_typos) const {
        size_t needle_len = needle_ascii.size();
        size_t first = needle_len - 1 - max_typos;
        size_t start = (len - 1) / LANES * LANES;
        for (;;) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start, len, chunk_mask);
            Mask mask = 0;
            for (size_t i = first; i < needle_len; i++) mask |= occ(chunk, needle_ascii[i]);
            mask &= chunk_mask;
            if (mask != 0) return start + LANES - m_lz(mask);
            if (start == 0) break;
            start -= LANES;
        }
        return len;
    }

    // =================== Unicode prefilter: prefilter/algo/unicode.rs ===================
    // unicode.rs:9-52
    static Mask match_unicode_char_prefix(size_t start, size_t len, const u8* h, int char_len, const u8 chars[4]) {
        switch (char_len) {
            case 2: return eq(load_raw(h, start, len), chars[0]);
            case 3: return eq(load_raw(h, start + 1, len), chars[1]) & eq(load_raw(h, start, len), chars[0]);
            default: return eq(load_raw(h, start + 2, len), chars[2]) & eq(load_raw(h, start + 1, len), chars[1]) & eq(load_raw(h, start, len), chars[0]);
        }
    }
    // unicode.rs:56-72
    static Mask char_variant_mask(const Chunk& chunk, Mask chunk_mask, u8 last_byte, size_t start, size_t len, const u8* h, int char_len, const u8 chars[4]) {
        Mask mask = eq(chunk, last_byte) & chunk_mask;
        if (mask != 0 && char_len > 1) mask &= match_unicode_char_prefix(start, len, h, char_len, chars);
        return mask;
    }
    // unicode.rs:74-117
    static Mask unicode_char_mask(size_t start, size_t len, const u8* h, const UnicodeChar& nc) {
        int char_len = nc.len;
        if (start + char_len > len) return 0;
        Mask chunk_mask;
        Chunk chunk = load_window(h, start + char_len - 1, len, chunk_mask);
        Mask mask = char_variant_mask(chunk, chunk_mask, nc.chars[char_len - 1], start, len, h, char_len, nc.chars);
        mask |= char_variant_mask(chunk, chunk_mask, nc.flipped[char_len - 1], start, len, h, char_len, nc.flipped);
        return mask;
    }
    // unicode.rs:119-219
    Window match_haystack_unicode(const u8* h, size_t len) const {
        if (len == 0) return {false, 0, 0};
        bool can_skip_chunks = true;
        size_t match_start_pos = 0;
        size_t needle_i = 0;
        const UnicodeChar* needle_char = &needle_unicode[needle_i++];
        u8 last0 = needle_char->chars[needle_char->len - 1], last1 = needle_char->flipped[needle_char->len - 1];
        size_t start = 0;
        while (start + needle_char->len <= len) {
            int char_len = needle_char->len;
            Mask valid;
            Chunk chunk = load_window(h, start + char_len - 1, len, valid);
            Mask available = m_all();
            for (;;) {
                Mask chunk_mask = available & valid;
                Mask mask = char_variant_mask(chunk, chunk_mask, last0, start, len, h, needle_char->len, needle_char->chars);
                mask |= char_variant_mask(chunk, chunk_mask, last1, start, len, h, needle_char->len, needle_char->flipped);
                if (mask == 0) break;
                available = m_clear_through_lowest(available, mask);
                if (can_skip_chunks) { match_start_pos = start + m_tz(mask); can_skip_chunks = false; }
                if (needle_i < needle_unicode.size()) {
                    needle_char = &needle_unicode[needle_i++];
                    last0 = needle_char->chars[needle_char->len - 1];
                    last1 = needle_char->flipped[needle_char->len - 1];
                    if (needle_char->len != char_len) {
                        if (start + needle_char->len > len) break;
                        char_len = needle_char->len;
                        chunk = load_window(h, start + char_len - 1, len, valid);
                    }
                } else if (start + needle_char->len - 1 + LANES >= len) {
                    return {true, match_start_pos, start + LANES - m_lz(mask) + needle_char->len - 1};
                } else {
                    size_t end_pos = start + find_last_unicode_char_pos(*needle_char, h + start, len - start);
                    return {true, match_start_pos, end_pos};
                }
            }
            start += LANES;
        }
        return {false, match_start_pos, len};
    }
    // unicode.rs:222-276
    static size_t find_last_unicode_char_pos(const UnicodeChar& nc, const u8* h, size_t len) {
        int char_len = nc.len;
        u8 l0 = nc.chars[char_len - 1], l1 = nc.flipped[char_len - 1];
        size_t back = (size_t)LANES + char_len - 1;
        size_t start = len >= back ? len - back : 0;
        for (;;) {
            Mask chunk_mask;
            Chunk chunk = load_window(h, start + char_len - 1, len, chunk_mask);
            Mask mask = (eq(chunk, l0) | eq(chunk, l1)) & chunk_mask;
            if (mask != 0 && char_len > 1)
                mask &= (match_unicode_char_prefix(start, len, h, char_len, nc.chars) | match_unicode_char_prefix(start, len, h, char_len, nc.flipped));
            if (mask != 0) return start + LANES - m_lz(mask) + char_len - 1;
            if (start == 0) break;
            start = start >= (size_t)LANES ? start - LANES : 0;
        }
        return len;
    }

    // ---- unicode typos: prefilter/algo/unicode_typos.rs ----
    // unicode_typos.rs:15-141
    Window match_haystack_unicode_1_typo(const u8* h, size_t len) const {
        size_t needle_len = needle_unicode.size();
        if (needle_len <= 1) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t first_idx = 0, second_idx = 1;
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask first_mask = unicode_char_mask(start, len, h, needle_unicode[first_idx]);
            Mask second_mask = unicode_char_mask(start, len, h, needle_unicode[second_idx]);
            Mask first_cm = m_all(), second_cm = m_all();
            for (;;) {
                bool advanced = false;
                size_t cand = first_idx + 1;
                if (cand > second_idx) {
                    if (cand == needle_len) return found_with_unicode_typos(h, len, match_start_pos, 1);
                    second_idx = cand;
                    second_cm = first_cm;
                    second_mask = unicode_char_mask(start, len, h, needle_unicode[second_idx]);
                } else if (cand == second_idx && first_cm > second_cm) {
                    second_cm = first_cm;
                }
                Mask fm = first_mask & first_cm;
                if (fm != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(fm));
                    first_idx += 1;
                    first_cm = m_clear_through_lowest(first_cm, fm);
                    first_mask = unicode_char_mask(start, len, h, needle_unicode[first_idx]);
                    advanced = true;
                }
                Mask sm = second_mask & second_cm;
                if (sm != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(sm));
                    second_idx += 1;
                    if (second_idx >= needle_len) return found_with_unicode_typos(h, len, match_start_pos, 1);
                    second_cm = m_clear_through_lowest(second_cm, sm);
                    second_mask = unicode_char_mask(start, len, h, needle_unicode[second_idx]);
                    advanced = true;
                }
                if (!advanced) break;
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }
    // unicode_typos.rs:144-330
    Window match_haystack_unicode_2_typos(const u8* h, size_t len) const {
        size_t needle_len = needle_unicode.size();
        if (needle_len <= 2) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t i1 = 0, i2 = 1, i3 = 2;
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask m1 = unicode_char_mask(start, len, h, needle_unicode[i1]);
            Mask m2 = unicode_char_mask(start, len, h, needle_unicode[i2]);
            Mask m3 = unicode_char_mask(start, len, h, needle_unicode[i3]);
            Mask c1 = m_all(), c2 = m_all(), c3 = m_all();
            for (;;) {
                bool advanced = false;
                size_t cand2 = i1 + 1;
                if (cand2 > i2) {
                    if (cand2 == needle_len) return found_with_unicode_typos(h, len, match_start_pos, 2);
                    i2 = cand2; c2 = c1; m2 = unicode_char_mask(start, len, h, needle_unicode[i2]);
                } else if (cand2 == i2 && c1 > c2) {
                    c2 = c1;
                }
                size_t cand3 = i2 + 1;
                if (cand3 > i3) {
                    if (cand3 == needle_len) return found_with_unicode_typos(h, len, match_start_pos, 2);
                    i3 = cand3; c3 = c2; m3 = unicode_char_mask(start, len, h, needle_unicode[i3]);
                } else if (cand3 == i3 && c2 > c3) {
                    c3 = c2;
                }
                Mask h1 = m1 & c1;
                if (h1 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h1));
                    i1 += 1;
                    c1 = m_clear_through_lowest(c1, h1);
                    m1 = unicode_char_mask(start, len, h, needle_unicode[i1]);
                    advanced = true;
                }
                Mask h2 = m2 & c2;
                if (h2 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h2));
                    i2 += 1;
                    if (i2 >= needle_len) return found_with_unicode_typos(h, len, match_start_pos, 2);
                    c2 = m_clear_through_lowest(c2, h2);
                    m2 = unicode_char_mask(start, len, h, needle_unicode[i2]);
                    advanced = true;
                }
                Mask h3 = m3 & c3;
                if (h3 != 0) {
                    match_start_pos = std::min(match_start_pos, start + m_tz(h3));
                    i3 += 1;
                    if (i3 >= needle_len) return found_with_unicode_typos(h, len, match_start_pos, 2);
                    c3 = m_clear_through_lowest(c3, h3);
                    m3 = unicode_char_mask(start, len, h, needle_unicode[i3]);
                    advanced = true;
                }
                if (!advanced) break;
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }
    // unicode_typos.rs:333-466
    Window match_haystack_unicode_many_typos(const u8* h, size_t len, size_t max_typos) const {
        size_t needle_len = needle_unicode.size();
        if (needle_len <= max_typos) return {true, 0, len};
        if (len == 0) return {false, 0, 0};
        size_t path_count = max_typos + 1;
        std::vector<size_t> idx(path_count, 0);
        std::vector<Mask> nmask(path_count, 0);
        size_t match_start_pos = SIZE_MAX;
        for (size_t start = 0; start < len; start += LANES) {
            Mask chunk_mask = m_all();
            for (size_t p = 0; p < path_count; p++) nmask[p] = unicode_char_mask(start, len, h, needle_unicode[idx[p]]);
            for (;;) {
                for (size_t p = 1; p < path_count; p++) {
                    size_t cand = idx[p - 1] + 1;
                    if (cand > idx[p]) {
                        if (cand == needle_len) return found_with_unicode_typos(h, len, match_start_pos, max_typos);
                        idx[p] = cand;
                        nmask[p] = unicode_char_mask(start, len, h, needle_unicode[cand]);
                    }
                }
                Mask match_mask = 0;
                for (size_t p = 0; p < path_count; p++) match_mask |= nmask[p];
                Mask matches = match_mask & chunk_mask;
                if (matches == 0) break;
                size_t hit_pos = m_tz(matches);
                Mask hit = matches & m_first_n(hit_pos + 1);
                match_start_pos = std::min(match_start_pos, start + hit_pos);
                for (size_t p = 0; p < path_count; p++) {
                    if ((nmask[p] & hit) == 0) continue;
                    idx[p] += 1;
                    if (idx[p] == needle_len) return found_with_unicode_typos(h, len, match_start_pos, max_typos);
                    nmask[p] = unicode_char_mask(start, len, h, needle_unicode[idx[p]]);
                }
                chunk_mask = m_clear_through_lowest(chunk_mask, hit);
            }
        }
        return {false, match_start_pos == SIZE_MAX ? 0 : match_start_pos, len};
    }
    // unicode_typos.rs:469-508
    Window found_with_unicode_typos(const u8* h, size_t len, size_t match_start_pos, size_t max_typos) const {
        return {true, match_start_pos, find_end_pos_with_unicode_typos(h, len, max_typos)};
    }
    size_t find_end_pos_with_unicode_typos(const u8* h, size_t len, size_t max_typos) const {
        size_t needle_len = needle_unicode.size();
        size_t first = needle_len - 1 - max_typos;
        size_t start = len >= (size_t)LANES ? len - LANES : 0;
        for (;;) {
            size_t end_pos = 0;
            for (size_t i = first; i < needle_len; i++) {
                const UnicodeChar& nc = needle_unicode[i];
                Mask mask = unicode_char_mask(start, len, h, nc);
                if (mask != 0) end_pos = std::max(end_pos, start + LANES - m_lz(mask) + nc.len - 1);
            }
            if (end_pos != 0) return end_pos;
            if (start == 0) break;
            start = start >= (size_t)LANES ? start - LANES : 0;
        }
        return len;
    }
};

// =======================================================================================
// SMITH-WATERMAN  (src/smith_waterman/algo/{ascii,ascii_gap,unicode,unicode_gap}.rs)
// =======================================================================================
static const size_t MAX_HAYSTACK_LEN = 1024;  // smith_waterman/algo/mod.rs:18

// smith_waterman/greedy.rs:7-91.  Returns false for None.
inline bool match_greedy(const std::string& needle_s, const u8* haystack, size_t hlen, const Scoring& scoring, bool case_sensitive, bool include_prefix, u16& score_out,
                         std::vector<u32>* indices = nullptr) {
    auto needle = case_needle(needle_s, case_sensitive);
    if (needle.size() > hlen) return false;
    u16 score = 0;
    size_t haystack_idx = 0;
    bool delimiter_bonus_enabled = false, prev_is_lower = false, prev_is_delim = false;
    for (size_t needle_idx = 0; needle_idx < needle.size(); needle_idx++) {
        u8 nc = needle[needle_idx].first, fc = needle[needle_idx].second;
        size_t haystack_start_idx = haystack_idx;
        bool found = false;
        while (haystack_idx <= (hlen - needle.size() + needle_idx)) {
            u8 hc = haystack[haystack_idx];
            bool is_digit = hc >= '0' && hc <= '9';
            bool is_upper = hc >= 'A' && hc <= 'Z';
            bool is_lower = hc >= 'a' && hc <= 'z';
            bool is_delim = hc < 128 && !(is_lower || is_upper || is_digit);
            if (!is_delim) delimiter_bonus_enabled = true;
            if (nc != hc && fc != hc) {
                prev_is_delim = delimiter_bonus_enabled && is_delim;
                prev_is_lower = is_lower;
                haystack_idx += 1;
                continue;
            }
            score = sat_add16(score, scoring.match_score);
            if (haystack_idx != haystack_start_idx && needle_idx != 0) {
                size_t gl = haystack_idx - haystack_start_idx;
                gl = gl > 0 ? gl - 1 : 0;
                u16 gap_len = (u16)std::min<size_t>(gl, 0xFFFF);
                u32 mul = (u32)scoring.gap_extend_penalty * gap_len;
                u16 ext = mul > 0xFFFF ? 0xFFFF : (u16)mul;
                score = sat_sub16(score, sat_add16(scoring.gap_open_penalty, ext));
            }
            if (nc == hc) score = sat_add16(score, scoring.matching_case_bonus);
            if (is_upper && prev_is_lower) score = sat_add16(score, scoring.capitalization_bonus);
            if (include_prefix && haystack_idx == 0) score = sat_add16(score, scoring.prefix_bonus);
            if (prev_is_delim && !is_delim) score = sat_add16(score, scoring.delimiter_bonus);
            prev_is_delim = delimiter_bonus_enabled && is_delim;
            prev_is_lower = is_lower;
            if (indices) indices->push_back((u32)haystack_idx);
            haystack_idx += 1;
            found = true;
            break;
        }
        if (!found) return false;
    }
    score_out = score;
    return true;
}

// A score vector of LANES lanes of T (ScalarScoreU16<LANES> / ScalarScoreU8<LANES>,
// smith_waterman/backend/scalar.rs:13-16, 147-356).  Masks are 0/all-ones lanes.
template <int LANES, typename T>
struct SV {
    T v[LANES];
    static SV zero() { SV r; for (int i = 0; i < LANES; i++) r.v[i] = 0; return r; }
    static SV splat(u16 x) { SV r; for (int i = 0; i < LANES; i++) r.v[i] = (T)x; return r; }          // `value as u8` truncation for u8
    static SV first_lane(u16 x) { SV r = zero(); r.v[0] = (T)x; return r; }
    SV max(const SV& o) const { SV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] > o.v[i] ? v[i] : o.v[i]; return r; }
    u16 horizontal_max() const { T m = 0; for (int i = 0; i < LANES; i++) if (v[i] > m) m = v[i]; return (u16)m; }
    SV add(const SV& o) const { SV r; for (int i = 0; i < LANES; i++) r.v[i] = (T)(v[i] + o.v[i]); return r; }              // wrapping
    SV subs(const SV& o) const { SV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] > o.v[i] ? (T)(v[i] - o.v[i]) : (T)0; return r; }  // saturating at 0
    SV and_(const SV& o) const { SV r; for (int i = 0; i < LANES; i++) r.v[i] = (T)(v[i] & o.v[i]); return r; }
    // shift_right_padded::<L>: low L lanes come from the top L lanes of prev (scalar.rs:222-232)
    SV shift_right_padded(int L, const SV& prev) const {
        SV r;
        for (int i = 0; i < L; i++) r.v[i] = prev.v[LANES - L + i];
        for (int i = L; i < LANES; i++) r.v[i] = v[i - L];
        return r;
    }
    T lane(int i) const { return v[i]; }
    // mask lanes (0xFF / 0x00 bytes) widened to score lanes (all-ones / zero): scalar.rs:18-42
    template <typename BYTES>
    static SV from_mask(const BYTES& m) { SV r; for (int i = 0; i < LANES; i++) r.v[i] = m.lane(i) ? (T)~(T)0 : (T)0; return r; }
};
// Byte/mask vector (ScalarBytes<LANES>, scalar.rs:8-145): mask lane = 0xFF / 0x00
template <int LANES>
struct BV {
    u8 v[LANES];
    static BV zero() { BV r; memset(r.v, 0, LANES); return r; }
    static BV load_partial(const u8* data, size_t start, size_t len) {  // scalar.rs:77-85
        BV r = zero();
        size_t take = len > start ? std::min<size_t>(len - start, LANES) : 0;
        for (size_t i = 0; i < take; i++) r.v[i] = data[start + i];
        return r;
    }
    BV eq(u8 c) const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] == c ? 0xFF : 0; return r; }
    BV gt(u8 c) const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] > c ? 0xFF : 0; return r; }
    BV lt(u8 c) const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] < c ? 0xFF : 0; return r; }
    BV and_(const BV& o) const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] & o.v[i]; return r; }
    BV or_(const BV& o) const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = v[i] | o.v[i]; return r; }
    BV not_() const { BV r; for (int i = 0; i < LANES; i++) r.v[i] = (u8)~v[i]; return r; }
    bool is_zero() const { for (int i = 0; i < LANES; i++) if (v[i]) return false; return true; }
    BV shift_right_padded_1(const BV& prev) const {  // scalar.rs:127-132
        BV r; r.v[0] = prev.v[LANES - 1];
        for (int i = 1; i < LANES; i++) r.v[i] = v[i - 1];
        return r;
    }
    static BV first_n(size_t n) { BV r = zero(); for (size_t i = 0; i < n && i < (size_t)LANES; i++) r.v[i] = 0xFF; return r; }
    u8 lane(int i) const { return v[i]; }
};

#if defined(FZO_AVX512)
// ---------------------------------------------------------------------------------------
// AVX-512 forms of the two lane-vector types at the widths the reference's AVX-512 backend uses
// (smith_waterman/backend/avx512.rs: 64 x u8 and 32 x u16 in one zmm).  Same operations, same
// names, so SmithWaterman<64,u8> / <32,u16> below run the identical algorithm text on real
// SIMD registers.  CPU-baseline build only; checked against the portable types by
// tests/test_oracle_avx512.py.
// ---------------------------------------------------------------------------------------
template <>
struct BV<64> {
    __m512i x;
    static BV zero() { BV r; r.x = _mm512_setzero_si512(); return r; }
    static BV load_partial(const u8* data, size_t start, size_t len) {
        const size_t take = len > start ? std::min<size_t>(len - start, 64) : 0;
        const __mmask64 k = take >= 64 ? ~(__mmask64)0 : (((__mmask64)1 << take) - 1);
        BV r; r.x = _mm512_maskz_loadu_epi8(k, data + start); return r;
    }
    static BV from_k(__mmask64 k) { BV r; r.x = _mm512_movm_epi8(k); return r; }
    BV eq(u8 c) const { return from_k(_mm512_cmpeq_epi8_mask(x, _mm512_set1_epi8((char)c))); }
    BV gt(u8 c) const { return from_k(_mm512_cmpgt_epu8_mask(x, _mm512_set1_epi8((char)c))); }
    BV lt(u8 c) const { return from_k(_mm512_cmplt_epu8_mask(x, _mm512_set1_epi8((char)c))); }
    BV and_(const BV& o) const { BV r; r.x = _mm512_and_si512(x, o.x); return r; }
    BV or_(const BV& o) const { BV r; r.x = _mm512_or_si512(x, o.x); return r; }
    BV not_() const { BV r; r.x = _mm512_xor_si512(x, _mm512_set1_epi8((char)0xFF)); return r; }
    bool is_zero() const { return _mm512_test_epi8_mask(x, x) == 0; }
    BV shift_right_padded_1(const BV& prev) const {
        alignas(64) static const u8 idx[64] = {63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,  74,  75,  76,  77,  78,  79,  80,  81,  82,  83,  84,
                                               85,  86,  87,  88,  89,  90,  91,  92,  93,  94,  95,  96,  97,  98,  99,  100, 101, 102, 103, 104, 105, 106,
                                               107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 126};
        BV r; r.x = _mm512_permutex2var_epi8(prev.x, _mm512_load_si512((const void*)idx), x); return r;  // index < 64: prev, >= 64: this
    }
    static BV first_n(size_t n) { return from_k(n >= 64 ? ~(__mmask64)0 : (((__mmask64)1 << n) - 1)); }
    u8 lane(int i) const { alignas(64) u8 t[64]; _mm512_store_si512((void*)t, x); return t[i]; }  // (only the mixed-width pairs, e.g. 64 x u16, go through this)
};
template <>
struct SV<64, u8> {
    __m512i x;
    static SV zero() { SV r; r.x = _mm512_setzero_si512(); return r; }
    static SV splat(u16 v) { SV r; r.x = _mm512_set1_epi8((char)(u8)v); return r; }  // `value as u8`
    static SV first_lane(u16 v) { SV r; r.x = _mm512_zextsi128_si512(_mm_cvtsi32_si128((int)(u8)v)); return r; }
    SV max(const SV& o) const { SV r; r.x = _mm512_max_epu8(x, o.x); return r; }
    u16 horizontal_max() const {
        __m256i a = _mm256_max_epu8(_mm512_castsi512_si256(x), _mm512_extracti64x4_epi64(x, 1));
        __m128i b = _mm_max_epu8(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1));
        b = _mm_max_epu8(b, _mm_srli_si128(b, 8));
        b = _mm_max_epu8(b, _mm_srli_si128(b, 4));
        b = _mm_max_epu8(b, _mm_srli_si128(b, 2));
        b = _mm_max_epu8(b, _mm_srli_si128(b, 1));
        return (u16)(u8)_mm_cvtsi128_si32(b);
    }
    SV add(const SV& o) const { SV r; r.x = _mm512_add_epi8(x, o.x); return r; }    // wrapping
    SV subs(const SV& o) const { SV r; r.x = _mm512_subs_epu8(x, o.x); return r; }  // saturating at 0
    SV and_(const SV& o) const { SV r; r.x = _mm512_and_si512(x, o.x); return r; }
    SV shift_right_padded(int L, const SV& prev) const {
        alignas(64) static const u8 iota[64] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
                                                22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43,
                                                44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63};
        const __m512i idx = _mm512_add_epi8(_mm512_load_si512((const void*)iota), _mm512_set1_epi8((char)(64 - L)));  // lane i <- concat(prev, this)[64 - L + i]
        SV r; r.x = _mm512_permutex2var_epi8(prev.x, idx, x); return r;
    }
    static SV from_mask(const BV<64>& m) { SV r; r.x = m.x; return r; }
    u8 lane(int i) const { alignas(64) u8 t[64]; _mm512_store_si512((void*)t, x); return t[i]; }
};
template <>
struct BV<32> {
    __m256i x;
    static BV zero() { BV r; r.x = _mm256_setzero_si256(); return r; }
    static BV load_partial(const u8* data, size_t start, size_t len) {
        const size_t take = len > start ? std::min<size_t>(len - start, 32) : 0;
        const __mmask32 k = take >= 32 ? ~(__mmask32)0 : (((__mmask32)1 << take) - 1);
        BV r; r.x = _mm256_maskz_loadu_epi8(k, data + start); return r;
    }
    static BV from_k(__mmask32 k) { BV r; r.x = _mm256_movm_epi8(k); return r; }
    BV eq(u8 c) const { return from_k(_mm256_cmpeq_epi8_mask(x, _mm256_set1_epi8((char)c))); }
    BV gt(u8 c) const { return from_k(_mm256_cmpgt_epu8_mask(x, _mm256_set1_epi8((char)c))); }
    BV lt(u8 c) const { return from_k(_mm256_cmplt_epu8_mask(x, _mm256_set1_epi8((char)c))); }
    BV and_(const BV& o) const { BV r; r.x = _mm256_and_si256(x, o.x); return r; }
    BV or_(const BV& o) const { BV r; r.x = _mm256_or_si256(x, o.x); return r; }
    BV not_() const { BV r; r.x = _mm256_xor_si256(x, _mm256_set1_epi8((char)0xFF)); return r; }
    bool is_zero() const { return _mm256_testz_si256(x, x) != 0; }
    BV shift_right_padded_1(const BV& prev) const {
        alignas(32) static const u8 idx[32] = {31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62};
        BV r; r.x = _mm256_permutex2var_epi8(prev.x, _mm256_load_si256((const __m256i*)idx), x); return r;
    }
    static BV first_n(size_t n) { return from_k(n >= 32 ? ~(__mmask32)0 : (((__mmask32)1 << n) - 1)); }
    u8 lane(int i) const { alignas(32) u8 t[32]; _mm256_store_si256((__m256i*)t, x); return t[i]; }
};
template <>
struct SV<32, u16> {
    __m512i x;
    static SV zero() { SV r; r.x = _mm512_setzero_si512(); return r; }
    static SV splat(u16 v) { SV r; r.x = _mm512_set1_epi16((short)v); return r; }
    static SV first_lane(u16 v) { SV r; r.x = _mm512_zextsi128_si512(_mm_cvtsi32_si128((int)v)); return r; }
    SV max(const SV& o) const { SV r; r.x = _mm512_max_epu16(x, o.x); return r; }
    u16 horizontal_max() const {
        __m256i a = _mm256_max_epu16(_mm512_castsi512_si256(x), _mm512_extracti64x4_epi64(x, 1));
        __m128i b = _mm_max_epu16(_mm256_castsi256_si128(a), _mm256_extracti128_si256(a, 1));
        b = _mm_max_epu16(b, _mm_srli_si128(b, 8));
        b = _mm_max_epu16(b, _mm_srli_si128(b, 4));
        b = _mm_max_epu16(b, _mm_srli_si128(b, 2));
        return (u16)_mm_cvtsi128_si32(b);
    }
    SV add(const SV& o) const { SV r; r.x = _mm512_add_epi16(x, o.x); return r; }
    SV subs(const SV& o) const { SV r; r.x = _mm512_subs_epu16(x, o.x); return r; }
    SV and_(const SV& o) const { SV r; r.x = _mm512_and_si512(x, o.x); return r; }
    SV shift_right_padded(int L, const SV& prev) const {
        alignas(64) static const u16 iota[32] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31};
        const __m512i idx = _mm512_add_epi16(_mm512_load_si512((const void*)iota), _mm512_set1_epi16((short)(32 - L)));  // lane i <- concat(prev, this)[32 - L + i]
        SV r; r.x = _mm512_permutex2var_epi16(prev.x, idx, x); return r;
    }
    static SV from_mask(const BV<32>& m) { SV r; r.x = _mm512_cvtepi8_epi16(m.x); return r; }  // 0xFF -> 0xFFFF (sign extension)
    u16 lane(int i) const { alignas(64) u16 t[32]; _mm512_store_si512((void*)t, x); return t[i]; }
};
#endif

template <int LANES, typename T>
struct SmithWaterman {
    typedef SV<LANES, T> S;
    typedef BV<LANES> B;

    std::string needle;
    std::vector<std::pair<u8, u8>> needle_simd;
    std::vector<UnicodeChar> needle_unicode;
    bool case_sensitive;
    Scoring scoring;
    // Matrix<B>: (needle_len+1) x (MAX_HAYSTACK_LEN/LANES + 1) (smith_waterman/matrix.rs:11-22)
    size_t stride;
    size_t chunks_used = 0;  // `haystack_chunks` of the most recent scoring call (smith_waterman/mod.rs:131-134)
    std::vector<S> score_matrix, match_masks, unicode_pending;

    SmithWaterman(const std::string& n, const Scoring& sc, bool cs)  // smith_waterman/algo/mod.rs:21-42
        : needle(n), needle_simd(case_needle(n, cs)), needle_unicode(case_needle_unicode(n, cs)), case_sensitive(cs), scoring(sc) {
        stride = (MAX_HAYSTACK_LEN + LANES - 1) / LANES + 1;
        score_matrix.assign((n.size() + 1) * stride, S::zero());
        match_masks.assign((n.size() + 1) * stride, S::zero());
        unicode_pending.assign(needle_unicode.size() + 1, S::zero());
    }
    S& sm(size_t r, size_t c) { return score_matrix[r * stride + c]; }
    S& mmx(size_t r, size_t c) { return match_masks[r * stride + c]; }

    static S widen(const B& m) { return S::from_mask(m); }  // scalar.rs:18-42

    // smith_waterman/algo/ascii_gap.rs:11-105 (gap_step! + propagate_{8,16,32,64}_lane)
    static S propagate_horizontal_gaps(S row, const S& adj, const S& mm, const S& amm, const S& gop, S gex) {
        for (int shift = 1; shift < LANES; shift *= 2) {
            S shifted_row = row.shift_right_padded(shift, adj);
            S shifted_mm = mm.shift_right_padded(shift, amm);
            S gap_penalty = gex.add(gop.and_(shifted_mm));
            S decayed = shifted_row.subs(gap_penalty);
            row = row.max(decayed);
            gex = gex.add(gex);
        }
        return row;
    }

    // smith_waterman/algo/ascii.rs:10-158
    u16 score_haystack(const u8* haystack, size_t hlen, bool include_prefix) {
        if (hlen > MAX_HAYSTACK_LEN) {
            u16 s;
            return match_greedy(needle, haystack, hlen, scoring, case_sensitive, include_prefix, s) ? s : 0;
        }
        size_t haystack_chunks = (hlen + LANES - 1) / LANES + 1;
        chunks_used = haystack_chunks;
        S gap_extend_penalty = S::splat(scoring.gap_extend_penalty);
        S gap_open_penalty = S::splat(sat_sub16(scoring.gap_open_penalty, scoring.gap_extend_penalty));
        S match_score = S::splat(sat_add16(scoring.match_score, scoring.mismatch_penalty));
        S mismatch_penalty = S::splat(scoring.mismatch_penalty);
        S matching_case_bonus = S::splat(scoring.matching_case_bonus);
        S capitalization_bonus = S::splat(scoring.capitalization_bonus);
        S delimiter_bonus = S::splat(scoring.delimiter_bonus);

        S prefix_bonus_masked = include_prefix ? S::first_lane(scoring.prefix_bonus) : S::zero();
        B prev_chunk_char_is_delimiter_mask = B::zero();
        B prev_chunk_is_lower_mask = B::zero();
        S max_scores = S::zero();

        for (size_t col_idx = 1; col_idx < haystack_chunks; col_idx++) {
            B haystack_chunk = B::load_partial(haystack, (col_idx - 1) * LANES, hlen);
            B is_upper_mask = haystack_chunk.lt('Z' + 1).and_(haystack_chunk.gt('A' - 1));
            B is_lower_mask = haystack_chunk.lt('z' + 1).and_(haystack_chunk.gt('a' - 1));
            B is_letter_mask = is_upper_mask.or_(is_lower_mask);
            S capitalization_mask = widen(is_upper_mask.and_(is_lower_mask.shift_right_padded_1(prev_chunk_is_lower_mask)));
            S capitalization_bonus_masked = capitalization_mask.and_(capitalization_bonus);
            prev_chunk_is_lower_mask = is_lower_mask;

            B is_digit_mask = haystack_chunk.gt('0' - 1).and_(haystack_chunk.lt('9' + 1));
            B char_is_delimiter_mask = is_letter_mask.or_(is_digit_mask).or_(haystack_chunk.gt(127)).not_();
            B prev_char_is_delimiter_mask = char_is_delimiter_mask.shift_right_padded_1(prev_chunk_char_is_delimiter_mask);
            S delimiter_mask = widen(prev_char_is_delimiter_mask.and_(char_is_delimiter_mask.not_()));
            S delimiter_bonus_masked = delimiter_mask.and_(delimiter_bonus);
            prev_chunk_char_is_delimiter_mask = char_is_delimiter_mask;

            S match_and_masked_bonuses = delimiter_bonus_masked.add(capitalization_bonus_masked).add(prefix_bonus_masked).add(match_score);

            S up_gap_mask = S::zero();
            S prev_row_scores = S::zero();
            S row_scores = S::zero();

            for (size_t row_idx = 1; row_idx <= needle_simd.size(); row_idx++) {
                u8 needle_char = needle_simd[row_idx - 1].first, flipped = needle_simd[row_idx - 1].second;
                B exact_case_match_mask_b = haystack_chunk.eq(needle_char);
                B flipped_case_match_mask = haystack_chunk.eq(flipped);
                S match_mask = widen(exact_case_match_mask_b.or_(flipped_case_match_mask));
                S exact_case_match_mask = widen(exact_case_match_mask_b);

                S diag = prev_row_scores.shift_right_padded(1, sm(row_idx - 1, col_idx - 1));
                diag = diag.add(match_mask.and_(match_and_masked_bonuses));
                diag = diag.subs(mismatch_penalty);
                S diag_scores = diag.add(exact_case_match_mask.and_(matching_case_bonus));

                S after_extend = prev_row_scores.subs(gap_extend_penalty);
                S up_scores = after_extend.subs(up_gap_mask.and_(gap_open_penalty));

                row_scores = propagate_horizontal_gaps(diag_scores.max(up_scores), sm(row_idx, col_idx - 1), match_mask, mmx(row_idx, col_idx - 1), gap_open_penalty, gap_extend_penalty);

                sm(row_idx, col_idx) = row_scores;
                mmx(row_idx, col_idx) = match_mask;
                prev_row_scores = row_scores;
                up_gap_mask = match_mask;
            }
            max_scores = max_scores.max(row_scores);
            prefix_bonus_masked = S::zero();
        }
        return max_scores.horizontal_max();
    }

    // smith_waterman/algo/unicode_gap.rs:110-236
    static void unicode_gap_step(int SHIFT, S& row, S& pending, const S& adj_row, const S& adj_pending, const S& cont_gex, const S& scalar_end_mask, const S& total_gex, const S& gop) {
        S shifted_row = row.shift_right_padded(SHIFT, adj_row);
        S shifted_pending = pending.shift_right_padded(SHIFT, adj_pending);
        S scalar_gap_extend_penalty = total_gex.subs(cont_gex);
        S pending_crossed = shifted_pending.and_(scalar_end_mask);
        S gap_penalty = scalar_gap_extend_penalty.add(gop.and_(pending_crossed));
        S candidate_row = shifted_row.subs(gap_penalty);
        row = row.max(candidate_row);
        S candidate_pending = shifted_pending.subs(scalar_end_mask);
        pending = pending.max(candidate_pending);
    }
    static void prepare_next_unicode_gap_step(int SHIFT, S& cont_gex, S& adj_cont_gex, S& scalar_end_mask, S& adj_scalar_end_mask, S& total_gex) {
        S zero = S::zero();
        S shifted_cont = cont_gex.shift_right_padded(SHIFT, adj_cont_gex);
        cont_gex = cont_gex.add(shifted_cont);
        adj_cont_gex = adj_cont_gex.add(adj_cont_gex.shift_right_padded(SHIFT, zero));
        S shifted_end = scalar_end_mask.shift_right_padded(SHIFT, adj_scalar_end_mask);
        scalar_end_mask = scalar_end_mask.max(shifted_end);
        adj_scalar_end_mask = adj_scalar_end_mask.max(adj_scalar_end_mask.shift_right_padded(SHIFT, zero));
        total_gex = total_gex.add(total_gex);
    }
    static void propagate_horizontal_unicode_gaps(S& row, const S& adj_row, S& pending, const S& adj_pending, S cont_gex, S adj_cont_gex, S scalar_end_mask, S adj_scalar_end_mask, const S& gop, const S& gex) {
        S total_gex = gex;
        int shift = 1;
        for (; shift < LANES / 2; shift *= 2) {
            unicode_gap_step(shift, row, pending, adj_row, adj_pending, cont_gex, scalar_end_mask, total_gex, gop);
            prepare_next_unicode_gap_step(shift, cont_gex, adj_cont_gex, scalar_end_mask, adj_scalar_end_mask, total_gex);
        }
        unicode_gap_step(shift, row, pending, adj_row, adj_pending, cont_gex, scalar_end_mask, total_gex, gop);
    }

    // smith_waterman/algo/unicode.rs:219-273
    static B valid_haystack_lanes(size_t hlen, size_t start) {
        size_t valid = hlen > start ? std::min<size_t>(hlen - start, LANES) : 0;
        return B::first_n(valid);
    }
    static B unicode_char_match_mask(const B chunks[4], const B& scalar_start_mask, int char_len, const u8 chars[4]) {
        B mask = chunks[4 - char_len].eq(chars[char_len - 1]).and_(scalar_start_mask);
        if (char_len > 1 && !mask.is_zero())
            for (int byte_idx = 0; byte_idx < char_len - 1; byte_idx++) mask = mask.and_(chunks[3 - byte_idx].eq(chars[byte_idx]));
        return mask;
    }

    // smith_waterman/algo/unicode.rs:10-217
    u16 score_haystack_unicode(const u8* haystack, size_t hlen, bool include_prefix) {
        if (hlen > MAX_HAYSTACK_LEN) {
            u16 s;
            return match_greedy(needle, haystack, hlen, scoring, case_sensitive, include_prefix, s) ? s : 0;
        }
        if (needle_unicode.empty()) return 0;
        size_t haystack_chunks = (hlen + LANES - 1) / LANES + 1;
        chunks_used = haystack_chunks;
        S gap_extend_penalty = S::splat(scoring.gap_extend_penalty);
        S gap_open_penalty = S::splat(sat_sub16(scoring.gap_open_penalty, scoring.gap_extend_penalty));
        S match_score = S::splat(sat_add16(scoring.match_score, scoring.mismatch_penalty));
        S mismatch_penalty = S::splat(scoring.mismatch_penalty);
        S matching_case_bonus = S::splat(scoring.matching_case_bonus);
        S capitalization_bonus = S::splat(scoring.capitalization_bonus);
        S delimiter_bonus = S::splat(scoring.delimiter_bonus);

        size_t final_row_idx = needle_unicode.size();
        S max_scores = S::zero();
        for (size_t i = 0; i <= final_row_idx; i++) unicode_pending[i] = S::zero();

        S prefix_bonus_masked = include_prefix ? S::first_lane(scoring.prefix_bonus) : S::zero();
        B prev_chunk_char_is_delimiter_mask = B::zero();
        B prev_chunk_is_lower_mask = B::zero();
        S prev_chunk_cont_gex = S::zero();
        S prev_chunk_scalar_start_mask = S::zero();

        for (size_t col_idx = 1; col_idx < haystack_chunks; col_idx++) {
            size_t chunk_start = (col_idx - 1) * LANES;
            B chunks[4] = {B::load_partial(haystack, chunk_start + 3, hlen), B::load_partial(haystack, chunk_start + 2, hlen),
                           B::load_partial(haystack, chunk_start + 1, hlen), B::load_partial(haystack, chunk_start, hlen)};
            B haystack_chunk = chunks[3];
            B valid_mask = valid_haystack_lanes(hlen, chunk_start);
            B continuation_mask = haystack_chunk.gt(0x7f).and_(haystack_chunk.lt(0xc0)).and_(valid_mask);
            B scalar_start_mask = continuation_mask.not_().and_(valid_mask);
            S scalar_start_score_mask = widen(scalar_start_mask);
            S continuation_gap_extend_penalty = widen(continuation_mask).and_(gap_extend_penalty);

            B is_upper_mask = haystack_chunk.lt('Z' + 1).and_(haystack_chunk.gt('A' - 1));
            B is_lower_mask = haystack_chunk.lt('z' + 1).and_(haystack_chunk.gt('a' - 1));
            B is_letter_mask = is_upper_mask.or_(is_lower_mask);
            S capitalization_mask = widen(is_upper_mask.and_(is_lower_mask.shift_right_padded_1(prev_chunk_is_lower_mask)));
            S capitalization_bonus_masked = capitalization_mask.and_(capitalization_bonus);
            prev_chunk_is_lower_mask = is_lower_mask;

            B is_digit_mask = haystack_chunk.gt('0' - 1).and_(haystack_chunk.lt('9' + 1));
            B char_is_delimiter_mask = is_letter_mask.or_(is_digit_mask).or_(haystack_chunk.gt(127)).not_();
            B prev_char_is_delimiter_mask = char_is_delimiter_mask.shift_right_padded_1(prev_chunk_char_is_delimiter_mask);
            S delimiter_mask = widen(prev_char_is_delimiter_mask.and_(char_is_delimiter_mask.not_()));
            S delimiter_bonus_masked = delimiter_mask.and_(delimiter_bonus);
            prev_chunk_char_is_delimiter_mask = char_is_delimiter_mask;

            S match_and_masked_bonuses = delimiter_bonus_masked.add(capitalization_bonus_masked).add(prefix_bonus_masked).add(match_score);
            prefix_bonus_masked = S::zero();

            S up_gap_mask = S::zero();
            S prev_row_scores = S::zero();
            S row_scores = S::zero();

            for (size_t row_idx = 1; row_idx <= needle_unicode.size(); row_idx++) {
                const UnicodeChar& nc = needle_unicode[row_idx - 1];
                B exact_b = unicode_char_match_mask(chunks, scalar_start_mask, nc.len, nc.chars);
                B flipped_b = unicode_char_match_mask(chunks, scalar_start_mask, nc.len, nc.flipped);
                S match_mask = widen(exact_b.or_(flipped_b));
                S exact_case_match_mask = widen(exact_b);

                S diag = prev_row_scores.shift_right_padded(1, sm(row_idx - 1, col_idx - 1));
                diag = diag.add(match_mask.and_(match_and_masked_bonuses));
                diag = diag.subs(mismatch_penalty);
                diag = diag.add(exact_case_match_mask.and_(matching_case_bonus));
                S diag_scores = diag.and_(scalar_start_score_mask);

                S after_extend = prev_row_scores.subs(gap_extend_penalty);
                S up = after_extend.subs(up_gap_mask.and_(gap_open_penalty));
                S up_scores = up.and_(scalar_start_score_mask);

                S next_row = diag_scores.max(up_scores);
                S pending = match_mask;
                propagate_horizontal_unicode_gaps(next_row, sm(row_idx, col_idx - 1), pending, unicode_pending[row_idx], continuation_gap_extend_penalty,
                                                  prev_chunk_cont_gex, scalar_start_score_mask, prev_chunk_scalar_start_mask, gap_open_penalty, gap_extend_penalty);

                sm(row_idx, col_idx) = next_row;
                mmx(row_idx, col_idx) = match_mask;
                unicode_pending[row_idx] = pending;
                prev_row_scores = next_row;
                row_scores = next_row;
                up_gap_mask = match_mask;
            }
            max_scores = max_scores.max(row_scores);
            prev_chunk_cont_gex = continuation_gap_extend_penalty;
            prev_chunk_scalar_start_mask = scalar_start_score_mask;
        }
        return max_scores.horizontal_max();
    }
    // ---- traceback (smith_waterman/alignment_iter.rs:35-181): walks the stored matrices from the first lane of the last row
    // that holds `score`; pushes the haystack byte position of every Match step (reverse order).  Returns false when the typo
    // budget was exceeded (the iterator yields None).  Columns are global lane numbers, column = chunk * LANES + lane, chunk 0 =
    // the zero column.
    u16 cell(size_t row, size_t col) { return (u16)sm(row, col / LANES).lane((int)(col % LANES)); }
    bool cell_is_match(size_t row, size_t col) { return mmx(row, col / LANES).lane((int)(col % LANES)) != 0; }
    template <typename F>
    bool walk_alignment(size_t needle_len, size_t haystack_start_pos, const u8* unicode_haystack, size_t unicode_len, u16 score, int max_typos, F on_match) {
        size_t col = SIZE_MAX;
        for (size_t c = 1; c < chunks_used && col == SIZE_MAX; c++)
            for (int i = 0; i < LANES; i++)
                if ((u16)sm(needle_len, c).lane(i) == score) { col = c * LANES + i; break; }
        if (col == SIZE_MAX) throw std::runtime_error("could not find max score in score matrix final row");
        size_t row = needle_len;
        u32 typos = 0;
        for (;;) {
            if (row == 0) return true;
            if (max_typos >= 0 && typos > (u32)max_typos) return false;
            if (col < (size_t)LANES || score == 0) {  // must be moving up only (at left edge), or lost alignment
                if (max_typos >= 0 && typos + (u32)row > (u32)max_typos) return false;
                return true;
            }
            size_t haystack_idx = col - LANES;
            if (unicode_haystack && haystack_idx < unicode_len && (unicode_haystack[haystack_idx] & 0xC0) == 0x80) {  // continuation byte: walk left
                col -= 1;
                score = cell(row, col);
                continue;
            }
            if (cell_is_match(row, col)) {
                on_match(row - 1, haystack_idx + haystack_start_pos);
                row -= 1;
                col -= 1;
                score = cell(row, col);
                continue;
            }
            u16 diag = cell(row - 1, col - 1), left = cell(row, col - 1), up = cell(row - 1, col);
            if (diag >= left && diag >= up) { row -= 1; col -= 1; typos += 1; score = diag; }
            else if (left >= up) { col -= 1; score = left; }
            else { typos += 1; row -= 1; score = up; }
        }
    }
    // has_alignment_path (smith_waterman/alignment.rs:24-35)
    bool has_alignment_path(u16 score, int max_typos) {
        return walk_alignment(needle.size(), 0, nullptr, 0, score, max_typos, [](size_t, size_t) {});
    }
    // smith_waterman/algo/mod.rs:49-94
    u16 score_haystack_indices(const u8* haystack, size_t hlen, size_t haystack_start_pos, int max_typos, std::vector<u32>& indices) {
        indices.clear();
        if (hlen > MAX_HAYSTACK_LEN) return greedy_indices(haystack, hlen, haystack_start_pos, indices);
        u16 score = score_haystack(haystack, hlen, haystack_start_pos == 0);
        if (score == 0) return 0;
        walk_alignment(needle.size(), haystack_start_pos, nullptr, 0, score, max_typos, [&](size_t, size_t pos) { indices.push_back((u32)pos); });
        return score;
    }
    // smith_waterman/algo/mod.rs:97-152
    u16 score_haystack_unicode_indices(const u8* haystack, size_t hlen, size_t haystack_start_pos, int max_typos, std::vector<u32>& indices) {
        indices.clear();
        if (hlen > MAX_HAYSTACK_LEN) return greedy_indices(haystack, hlen, haystack_start_pos, indices);
        u16 score = score_haystack_unicode(haystack, hlen, haystack_start_pos == 0);
        if (score == 0) return 0;
        size_t prev = SIZE_MAX;
        walk_alignment(needle_unicode.size(), haystack_start_pos, haystack, hlen, score, max_typos, [&](size_t needle_idx, size_t pos) {
            if (prev != pos) {
                int len = needle_unicode[needle_idx].len;
                for (int off = len - 1; off >= 0; off--) indices.push_back((u32)(pos + off));
                prev = pos;
            }
        });
        return score;
    }
    u16 greedy_indices(const u8* haystack, size_t hlen, size_t haystack_start_pos, std::vector<u32>& indices) {
        u16 s;
        std::vector<u32> fwd;
        if (!match_greedy(needle, haystack, hlen, scoring, case_sensitive, haystack_start_pos == 0, s, &fwd)) return 0;
        for (size_t i = fwd.size(); i-- > 0;) indices.push_back((u32)(fwd[i] + haystack_start_pos));
        return s;
    }
};


// =======================================================================================
// ORDERING: src/sort.rs:6-40, src/k_merge.rs:90-170
// =======================================================================================
inline void radix_sort_matches(std::vector<Match>& matches) {
    size_t n = matches.size();
    u32 histogram[256] = {0};
    for (auto& m : matches) histogram[m.score & 0xFF]++;
    u32 offsets[256] = {0};
    for (int idx = 255; idx >= 1; idx--) offsets[idx - 1] = offsets[idx] + histogram[idx];
    std::vector<Match> b(n);
    for (auto& m : matches) b[offsets[m.score & 0xFF]++] = m;
    memset(histogram, 0, sizeof(histogram));
    for (auto& m : b) histogram[(m.score >> 8) & 0xFF]++;
    offsets[255] = 0;
    for (int idx = 255; idx >= 1; idx--) offsets[idx - 1] = offsets[idx] + histogram[idx];
    for (auto& m : b) matches[offsets[(m.score >> 8) & 0xFF]++] = m;
}

inline bool merge_less(int order, const Match& l, const Match& r) {  // k_merge.rs:14-53
    switch (order) {
        case SORT_SCORE_THEN_INDEX_ASC: return l.score > r.score || (l.score == r.score && l.index < r.index);
        case SORT_SCORE_THEN_INDEX_DESC: return l.score > r.score || (l.score == r.score && l.index > r.index);
        case SORT_INDEX_ASC: return l.index < r.index;
        default: return l.index > r.index;
    }
}
// k_merge.rs:90-170 (binary heap of run cursors)
inline std::vector<Match> k_merge_matches_by(int order, const std::vector<std::vector<Match>>& runs) {
    struct Cursor { size_t run_idx, match_idx; Match head; };
    size_t total = 0;
    for (auto& r : runs) total += r.size();
    std::vector<Match> merged;
    merged.reserve(total);
    std::vector<Cursor> heap;
    for (size_t i = 0; i < runs.size(); i++) if (!runs[i].empty()) heap.push_back({i, 0, runs[i][0]});
    auto sift_down = [&](size_t index) {
        size_t pos = index, child = 2 * pos + 1;
        while (child + 1 < heap.size()) {
            child += merge_less(order, heap[child + 1].head, heap[child].head) ? 1 : 0;
            if (!merge_less(order, heap[child].head, heap[pos].head)) return;
            std::swap(heap[pos], heap[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child < heap.size() && merge_less(order, heap[child].head, heap[pos].head)) std::swap(heap[pos], heap[child]);
    };
    for (size_t i = heap.size() / 2; i-- > 0;) sift_down(i);
    while (heap.size() > 1) {
        size_t run_idx = heap[0].run_idx, next = heap[0].match_idx + 1;
        merged.push_back(heap[0].head);
        if (next < runs[run_idx].size()) { heap[0].match_idx = next; heap[0].head = runs[run_idx][next]; }
        else { heap[0] = heap.back(); heap.pop_back(); }
        sift_down(0);
    }
    if (!heap.empty()) {
        auto& c = heap.back();
        merged.insert(merged.end(), runs[c.run_idx].begin() + c.match_idx, runs[c.run_idx].end());
    }
    return merged;
}

// =======================================================================================
// MATCHER: src/matcher/mod.rs:90-222, 373-411, 448-498; src/matcher/algo.rs:57-103, 172-193,
// 230-263, 302-338; src/matcher/parallel.rs:18-89
// =======================================================================================
struct HaystackList {  // packed bytes + exclusive end offsets (the boundary's corpus format)
    const u8* bytes;
    const u64* ends;
    size_t n;
    // optional indirection: item i is haystack gather[i] of the packed list (the `gathered: Vec<&str>` of matcher/multi.rs:107-114)
    const u32* gather = nullptr;
    size_t at(size_t i) const { return gather ? (size_t)gather[i] : i; }
    const u8* ptr(size_t i) const { size_t k = at(i); return bytes + (k ? ends[k - 1] : 0); }
    size_t len(size_t i) const { size_t k = at(i); return (size_t)(ends[k] - (k ? ends[k - 1] : 0)); }
};

struct MatcherBase {
    virtual ~MatcherBase() {}
    virtual void match_list_into(const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) = 0;
    // match_list_indices_impl (matcher/algo.rs:196-227; literal/algo.rs:129-155): matches + the matched byte positions, reverse order
    virtual void match_list_indices(const HaystackList& hs, std::vector<Match>& out, std::vector<std::vector<u32>>& indices) = 0;
    virtual MatcherBase* clone() const = 0;
};

// MatcherImpl<P,S> (matcher/algo.rs:47-103), parameterised like the backend enum
// (matcher/backend.rs:26-79): PF_LANES = prefilter lanes, SW_LANES x T = score vector.
template <int PF_LANES, int SW_LANES, typename T>
struct MatcherImpl : MatcherBase {
    std::string needle;
    Config config;
    size_t min_haystack_len;
    bool needs_unicode;
    Prefilter<PF_LANES> prefilter;
    SmithWaterman<SW_LANES, T> sw;

    MatcherImpl(const std::string& n, const Config& c, bool case_sensitive, bool unicode)
        : needle(n), config(c), needs_unicode(unicode), prefilter(n, case_sensitive), sw(n, c.scoring, case_sensitive) {
        size_t nchars = utf8_decode((const u8*)n.data(), n.size()).size();
        min_haystack_len = c.max_typos < 0 ? 0 : (nchars > (size_t)c.max_typos ? nchars - (size_t)c.max_typos : 0);  // algo.rs:62-65
    }
    MatcherBase* clone() const override { return new MatcherImpl(*this); }

    Window prefilter_haystack(const u8* h, size_t len) const {  // algo.rs:172-193 + dispatch_typos! mod.rs:58-73
        int t = config.max_typos;
        if (t < 0) return {true, 0, len};
        if (needs_unicode) {
            if (t == 0) return prefilter.match_haystack_unicode(h, len);
            if (t == 1) return prefilter.match_haystack_unicode_1_typo(h, len);
            if (t == 2) return prefilter.match_haystack_unicode_2_typos(h, len);
            return prefilter.match_haystack_unicode_many_typos(h, len, (size_t)t);
        }
        if (t == 0) return prefilter.match_haystack(h, len);
        if (t == 1) return prefilter.match_haystack_1_typo(h, len);
        if (t == 2) return prefilter.match_haystack_2_typos(h, len);
        return prefilter.match_haystack_many_typos(h, len, (size_t)t);
    }

    void match_list_into(const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) override {  // algo.rs:78-103
        for (size_t i = lo; i < hi; i++) {
            const u8* h = hs.ptr(i);
            size_t original_len = hs.len(i);
            if (original_len < min_haystack_len) continue;
            Window w = prefilter_haystack(h, original_len);
            if (!w.matched) continue;
            // trim_haystack (algo.rs:332-338)
            size_t start_pos = w.start > 0 ? w.start - 1 : 0;
            bool include_exact = start_pos == 0 && w.end == original_len;
            const u8* trimmed = h + start_pos;
            size_t tlen = w.end - start_pos;
            // smith_waterman_one (algo.rs:230-263); include_prefix = start_pos == 0 (algo/mod.rs:156-163)
            u16 score = needs_unicode ? sw.score_haystack_unicode(trimmed, tlen, start_pos == 0) : sw.score_haystack(trimmed, tlen, start_pos == 0);
            bool exact = include_exact && needle.size() == tlen && memcmp(needle.data(), trimmed, tlen) == 0;
            if (exact) score = (u16)(score + config.scoring.exact_match_bonus);
            out.push_back(Match{(u32)(index_offset + (i - lo)), score, (u8)(exact ? 1 : 0), 0});
        }
    }

    void match_list_indices(const HaystackList& hs, std::vector<Match>& out, std::vector<std::vector<u32>>& indices) override {  // algo.rs:196-227, 264-292
        const int max_typos_opt = config.max_typos < 0 ? -1 : config.max_typos;  // NO_PREFILTER -> None
        for (size_t i = 0; i < hs.n; i++) {
            const u8* h = hs.ptr(i);
            size_t original_len = hs.len(i);
            if (original_len < min_haystack_len) continue;
            Window w = prefilter_haystack(h, original_len);
            if (!w.matched) continue;
            size_t start_pos = w.start > 0 ? w.start - 1 : 0;
            bool include_exact = start_pos == 0 && w.end == original_len;
            const u8* trimmed = h + start_pos;
            size_t tlen = w.end - start_pos;
            std::vector<u32> idx;
            u16 score = needs_unicode ? sw.score_haystack_unicode_indices(trimmed, tlen, start_pos, max_typos_opt, idx) : sw.score_haystack_indices(trimmed, tlen, start_pos, max_typos_opt, idx);
            bool exact = include_exact && needle.size() == tlen && memcmp(needle.data(), trimmed, tlen) == 0;
            if (exact) score = (u16)(score + config.scoring.exact_match_bonus);
            out.push_back(Match{(u32)i, score, (u8)(exact ? 1 : 0), 0});
            indices.push_back(idx);
        }
    }
};

// =======================================================================================
// LITERAL MATCHING: exact / prefix / suffix / substring, src/literal/algo.rs.
// The needle must occur as a contiguous run; scoring is the Smith-Waterman bonuses of a gap-free
// alignment.  Independent of the SIMD width (the reference asserts all its backends agree,
// src/literal/backend.rs:103-200), so the two-seed-byte scan is restated as a plain scan.
// =======================================================================================
inline bool literal_is_delimiter(u8 b) { return b <= 127 && !((b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z') || (b >= '0' && b <= '9')); }  // algo.rs:326-329

struct LiteralMatcher : MatcherBase {
    int mode;
    Scoring scoring;
    std::string needle;
    bool unicode;
    std::vector<std::pair<u8, u8>> needle_ascii;
    std::vector<UnicodeChar> needle_unicode;

    LiteralMatcher(const std::string& n, const Config& c, bool case_sensitive, bool unicode_)
        : mode(c.matching), scoring(c.scoring), needle(n), unicode(unicode_), needle_ascii(case_needle(n, case_sensitive)), needle_unicode(case_needle_unicode(n, case_sensitive)) {}
    MatcherBase* clone() const override { return new LiteralMatcher(*this); }

    bool matches_at(const u8* h, size_t pos) const {  // algo.rs:157-176
        if (unicode) {
            size_t k = pos;
            for (const UnicodeChar& c : needle_unicode) {
                if (memcmp(h + k, c.chars, c.len) != 0 && memcmp(h + k, c.flipped, c.len) != 0) return false;
                k += c.len;
            }
        } else {
            for (size_t k = 0; k < needle_ascii.size(); k++) {
                u8 b = h[pos + k];
                if (b != needle_ascii[k].first && b != needle_ascii[k].second) return false;
            }
        }
        return true;
    }
    u16 score_scalar(const u8* h, size_t start, bool matched_exact_case) const {  // algo.rs:180-200
        u16 score = scoring.match_score;
        if (matched_exact_case) score = (u16)(score + scoring.matching_case_bonus);
        if (start == 0) {
            score = (u16)(score + scoring.prefix_bonus);
        } else {
            u8 b = h[start], prev = h[start - 1];
            if (b >= 'A' && b <= 'Z' && prev >= 'a' && prev <= 'z') score = (u16)(score + scoring.capitalization_bonus);
            if (literal_is_delimiter(prev) && !literal_is_delimiter(b)) score = (u16)(score + scoring.delimiter_bonus);
        }
        return score;
    }
    u16 score_at(const u8* h, size_t hlen, size_t pos) const {  // algo.rs:204-225
        u16 score = 0;
        if (unicode) {
            size_t start = pos;
            for (const UnicodeChar& c : needle_unicode) {
                score = (u16)(score + score_scalar(h, start, memcmp(h + start, c.chars, c.len) == 0));
                start += c.len;
            }
        } else {
            for (size_t k = 0; k < needle_ascii.size(); k++) score = (u16)(score + score_scalar(h, pos + k, h[pos + k] == needle_ascii[k].first));
        }
        if (pos == 0 && needle.size() == hlen) score = (u16)(score + scoring.exact_match_bonus);
        return score;
    }
    bool find(const u8* h, size_t hlen, size_t& pos_out, u16& score_out) const {  // algo.rs:232-312
        size_t nl = needle.size();
        if (hlen < nl) return false;
        auto hit = [&](size_t pos) { pos_out = pos; score_out = score_at(h, hlen, pos); return true; };
        switch (mode) {
            case MATCH_EXACT: return hlen == nl && matches_at(h, 0) && hit(0);
            case MATCH_PREFIX: return matches_at(h, 0) && hit(0);
            case MATCH_SUFFIX: return matches_at(h, hlen - nl) && hit(hlen - nl);
            default: {  // substring: best score, earliest position on ties
                bool found = false;
                for (size_t pos = 0; pos + nl <= hlen; pos++) {
                    if (!matches_at(h, pos)) continue;
                    u16 sc = score_at(h, hlen, pos);
                    if (!found || sc > score_out) { found = true; pos_out = pos; score_out = sc; }
                }
                return found;
            }
        }
    }
    void match_list_into(const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) override {  // algo.rs:95-127
        for (size_t i = lo; i < hi; i++) {
            size_t pos = 0;
            u16 score = 0;
            if (!find(hs.ptr(i), hs.len(i), pos, score)) continue;
            bool exact = pos == 0 && needle.size() == hs.len(i);
            out.push_back(Match{(u32)(index_offset + (i - lo)), score, (u8)(exact ? 1 : 0), 0});
        }
    }
    void match_list_indices(const HaystackList& hs, std::vector<Match>& out, std::vector<std::vector<u32>>& indices) override {  // algo.rs:129-155
        for (size_t i = 0; i < hs.n; i++) {
            size_t pos = 0;
            u16 score = 0;
            if (!find(hs.ptr(i), hs.len(i), pos, score)) continue;
            bool exact = pos == 0 && needle.size() == hs.len(i);
            out.push_back(Match{(u32)i, score, (u8)(exact ? 1 : 0), 0});
            std::vector<u32> idx;
            for (size_t k = needle.size(); k-- > 0;) idx.push_back((u32)(pos + k));  // the whole UTF-8 run, reversed
            indices.push_back(idx);
        }
    }
};

struct Matcher {
    Config config;
    std::string needle;
    bool empty;
    int pf_lanes, sw_lanes;  // resolved (never 0)
    bool use_u8;
    MatcherBase* impl = nullptr;

    // Matcher::new (matcher/mod.rs:90-111, 178-204) + get_backend (mod.rs:448-498).
    // pf_lanes/sw_lanes select which ISA's backend pair is emulated:
    //   AVX-512(+VBMI): pf 64, sw 64 (u8) / 32 (u16);  AVX2: 32, 32/16;  SSE/NEON: 16, 16/8;  scalar: 16, 16/8.
    Matcher(const std::string& n, const Config& c, int pf_lanes_, int sw_lanes_u8, int sw_lanes_u16) : config(c), needle(n), empty(n.empty()) {
        utf8_decode((const u8*)n.data(), n.size());  // validates
        pf_lanes = pf_lanes_;
        use_u8 = score_fits_in_u8(n.size(), c.scoring);
        sw_lanes = use_u8 ? sw_lanes_u8 : sw_lanes_u16;
        if (empty) return;
        bool case_sensitive = respects_case_for(c.casing, n);
        bool unicode = respects_unicode_for(c.unicode, n);
        if (c.matching != MATCH_FUZZY) {  // get_literal_backend (mod.rs:449-451); LiteralImpl::new (literal/algo.rs:33-36, 314-322)
            u16 bonus = sat_add16(std::max(c.scoring.capitalization_bonus, c.scoring.delimiter_bonus), c.scoring.matching_case_bonus);
            std::string err = guard_against_score_overflow(c.scoring, n.size(), bonus, 0);
            if (!err.empty()) throw std::runtime_error(err);
            impl = new LiteralMatcher(n, c, case_sensitive, unicode);
            return;
        }
        // guard_against_score_overflow (algo.rs:311-325): rows = chars on the unicode path, bytes otherwise
        size_t rows = unicode ? utf8_decode((const u8*)n.data(), n.size()).size() : n.size();
        std::string err = guard_against_score_overflow(c.scoring, rows);
        if (!err.empty()) throw std::runtime_error(err);
#define FZO_MK(PF, SW, TY) impl = new MatcherImpl<PF, SW, TY>(n, c, case_sensitive, unicode)
        if (use_u8) {
            if (pf_lanes == 64 && sw_lanes == 64) FZO_MK(64, 64, u8);
            else if (pf_lanes == 32 && sw_lanes == 32) FZO_MK(32, 32, u8);
            else if (pf_lanes == 16 && sw_lanes == 16) FZO_MK(16, 16, u8);
            else if (pf_lanes == 64 && sw_lanes == 32) FZO_MK(64, 32, u8);
            else if (pf_lanes == 64 && sw_lanes == 16) FZO_MK(64, 16, u8);
            else if (pf_lanes == 16 && sw_lanes == 64) FZO_MK(16, 64, u8);  // no CPU backend pairs these; the GPU tests do, to reach the
            else if (pf_lanes == 32 && sw_lanes == 64) FZO_MK(32, 64, u8);  // short-haystack scorer with a multi-chunk prefilter
            else throw std::runtime_error("unsupported (pf_lanes, sw_lanes) for u8 class");
        } else {
            if (pf_lanes == 64 && sw_lanes == 32) FZO_MK(64, 32, u16);
            else if (pf_lanes == 32 && sw_lanes == 16) FZO_MK(32, 16, u16);
            else if (pf_lanes == 16 && sw_lanes == 8) FZO_MK(16, 8, u16);
            else if (pf_lanes == 64 && sw_lanes == 16) FZO_MK(64, 16, u16);
            else if (pf_lanes == 64 && sw_lanes == 8) FZO_MK(64, 8, u16);
            else throw std::runtime_error("unsupported (pf_lanes, sw_lanes) for u16 class");
        }
#undef FZO_MK
    }
    ~Matcher() { delete impl; }
    Matcher(const Matcher&) = delete;

    static std::string guard_against_haystack_overflow(size_t n, u32 index_offset) {  // mod.rs:438-446
        if (n + (size_t)index_offset > 0xFFFFFFFFull)
            return "too many items in haystack, will overflow the u32 index: " + std::to_string(n + index_offset) + " > 4294967295 (index offset: " + std::to_string(index_offset) + ")";
        return "";
    }

    void match_list_into(MatcherBase* m, const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) const {  // mod.rs:373-392
        std::string err = guard_against_haystack_overflow(hi - lo, index_offset);
        if (!err.empty()) throw std::runtime_error(err);
        if (empty) {
            for (size_t i = lo; i < hi; i++) out.push_back(Match{(u32)(index_offset + (i - lo)), 0, 0, 0});
            return;
        }
        m->match_list_into(hs, lo, hi, index_offset, out);
    }

    std::vector<Match> match_list(const HaystackList& hs) const {  // mod.rs:212-222
        std::vector<Match> matches;
        match_list_into(impl, hs, 0, hs.n, 0, matches);
        if (sort_is_reversed(config.sort)) std::reverse(matches.begin(), matches.end());
        if (!empty && sort_is_by_score(config.sort)) radix_sort_matches(matches);
        return matches;
    }

    std::vector<Match> match_list_parallel(const HaystackList& hs, size_t threads) const {  // parallel.rs:18-89
        std::string err = guard_against_haystack_overflow(hs.n, 0);
        if (!err.empty()) throw std::runtime_error(err);
        if (threads == 0) throw std::runtime_error("threads must be positive");
        threads = std::max<size_t>(std::min(threads, (hs.n + 1999) / 2000), 1);
        if (hs.n == 0 || empty || threads == 1) return match_list(hs);
        const size_t chunk_size = 2048;
        size_t num_chunks = (hs.n + chunk_size - 1) / chunk_size;
        std::atomic<size_t> next_chunk(0);
        std::vector<std::vector<Match>> runs(threads);
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; t++) {
            pool.emplace_back([&, t]() {
                MatcherBase* local = impl->clone();
                std::vector<Match>& local_matches = runs[t];
                for (;;) {
                    size_t chunk_idx = next_chunk.fetch_add(1, std::memory_order_relaxed);
                    if (chunk_idx >= num_chunks) break;
                    size_t start = chunk_idx * chunk_size, end = std::min(start + chunk_size, hs.n);
                    match_list_into(local, hs, start, end, (u32)start, local_matches);
                }
                if (sort_is_reversed(config.sort)) std::reverse(local_matches.begin(), local_matches.end());
                if (sort_is_by_score(config.sort)) radix_sort_matches(local_matches);
                delete local;
            });
        }
        for (auto& th : pool) th.join();
        return k_merge_matches_by(config.sort, runs);
    }

    // Timing aid for bench.py's cpu_baseline (not part of the reference API): the worker loop of match_list_parallel
    // WITHOUT the per-thread sort and the single-threaded k-way merge, i.e. the same scope as one step of the GPU
    // pipeline (score every haystack, keep the records of the accepted ones).  Returns the number of matches.
    size_t score_parallel_unordered(const HaystackList& hs, size_t threads) const {
        if (threads == 0) throw std::runtime_error("threads must be positive");
        threads = std::max<size_t>(std::min(threads, (hs.n + 1999) / 2000), 1);
        if (hs.n == 0 || empty) return hs.n;
        const size_t chunk_size = 2048;
        size_t num_chunks = (hs.n + chunk_size - 1) / chunk_size;
        std::atomic<size_t> next_chunk(0), total(0);
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; t++) {
            pool.emplace_back([&]() {
                MatcherBase* local = impl->clone();
                std::vector<Match> local_matches;
                for (;;) {
                    size_t chunk_idx = next_chunk.fetch_add(1, std::memory_order_relaxed);
                    if (chunk_idx >= num_chunks) break;
                    size_t start = chunk_idx * chunk_size, end = std::min(start + chunk_size, hs.n);
                    match_list_into(local, hs, start, end, (u32)start, local_matches);
                }
                total.fetch_add(local_matches.size());
                delete local;
            });
        }
        for (auto& th : pool) th.join();
        return total.load();
    }
};

// =======================================================================================
// MULTI-PATTERN COMPOSITION: src/matcher/multi.rs.
// A pattern = needle + negation + per-pattern overrides resolved against the matcher's config
// (PatternConfig::resolve, src/pattern.rs:250-262; Matcher::compile, src/matcher/mod.rs:192-204).
// =======================================================================================
struct PatternSpec {
    std::string needle;
    bool negated = false;
    bool has_max_typos = false;  // Some(k): overrides; None: inherits (even when the config says None)
    int max_typos = 0;
    int casing = -1, unicode = -1;  // -1: inherit
    bool has_scoring = false;
    Scoring scoring;
    int matching = -1;  // -1: inherit Config::matching
};

// Pattern::parse (src/pattern.rs:87-167): one query atom; `^foo` prefix, `foo$` suffix, `^foo$` exact, `'foo` substring, `!foo` negated
// (a bare negated atom matches substrings), backslash escapes.
inline PatternSpec parse_pattern(const std::string& atom) {
    std::vector<u32> cps = utf8_decode((const u8*)atom.data(), atom.size());
    std::vector<std::pair<u32, bool>> tokens;  // (char, escaped)
    for (size_t i = 0; i < cps.size(); i++) {
        if (cps[i] == '\\' && i + 1 < cps.size()) { tokens.push_back({cps[i + 1], true}); i++; }
        else tokens.push_back({cps[i], false});
    }
    size_t lo = 0, hi = tokens.size();
    auto strip_first = [&](u32 op) { if (lo < hi && !tokens[lo].second && tokens[lo].first == op) { lo++; return true; } return false; };
    auto strip_last = [&](u32 op) { if (lo < hi && !tokens[hi - 1].second && tokens[hi - 1].first == op) { hi--; return true; } return false; };
    bool negated = strip_first('!');
    bool prefix = strip_first('^');
    bool substring = !prefix && strip_first('\'');
    bool suffix = strip_last('$');
    auto is_ws = [](u32 c) {  // char::is_whitespace (Unicode White_Space)
        return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
    };
    auto is_special = [&](u32 c) { return c == '!' || c == '^' || c == '\'' || c == '$' || is_ws(c); };
    PatternSpec sp;
    for (size_t i = lo; i < hi; i++) {
        if (tokens[i].second && !is_special(tokens[i].first)) sp.needle.push_back('\\');
        u8 buf[4];
        int n = utf8_encode(tokens[i].first, buf);
        sp.needle.append((const char*)buf, n);
    }
    sp.negated = negated;
    if (prefix && suffix) sp.matching = MATCH_EXACT;
    else if (prefix) sp.matching = MATCH_PREFIX;
    else if (suffix) sp.matching = MATCH_SUFFIX;
    else if (substring) sp.matching = MATCH_SUBSTRING;
    else if (negated) sp.matching = MATCH_SUBSTRING;
    return sp;
}
// Pattern::parse_query (src/pattern.rs:186-222): whitespace separated atoms, backslash keeps the next char in the atom, empty needles dropped
inline std::vector<PatternSpec> parse_query(const std::string& query) {
    std::vector<PatternSpec> out;
    std::vector<u32> cps = utf8_decode((const u8*)query.data(), query.size());
    auto is_ws = [](u32 c) {
        return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
    };
    std::string atom;
    bool in_atom = false, escaped = false;
    auto push = [&]() {
        PatternSpec p = parse_pattern(atom);
        if (!p.needle.empty()) out.push_back(p);
        atom.clear();
        in_atom = false;
    };
    for (u32 c : cps) {
        u8 buf[4];
        int n = utf8_encode(c, buf);
        if (escaped) { escaped = false; atom.append((const char*)buf, n); }
        else if (c == '\\') { in_atom = true; escaped = true; atom.append((const char*)buf, n); }
        else if (is_ws(c)) { if (in_atom) push(); }
        else { in_atom = true; atom.append((const char*)buf, n); }
    }
    if (in_atom) push();
    return out;
}

struct MultiMatcher {
    Config config;
    struct Compiled { bool negated; Matcher* m; };
    std::vector<Compiled> patterns;  // empty needles are dropped (mod.rs:193-195)

    MultiMatcher(const std::vector<PatternSpec>& specs, const Config& c, int pf_lanes, int sw_lanes_u8, int sw_lanes_u16) : config(c) {
        for (const PatternSpec& sp : specs) {
            if (sp.needle.empty()) continue;
            Config rc = c;  // PatternConfig::resolve: sort is always the matcher's
            if (sp.has_max_typos) rc.max_typos = sp.max_typos;
            if (sp.casing >= 0) rc.casing = sp.casing;
            if (sp.unicode >= 0) rc.unicode = sp.unicode;
            if (sp.has_scoring) rc.scoring = sp.scoring;
            if (sp.matching >= 0) rc.matching = sp.matching;
            rc.sort = SORT_INDEX_ASC;
            patterns.push_back(Compiled{sp.negated, new Matcher(sp.needle, rc, pf_lanes, sw_lanes_u8, sw_lanes_u16)});
        }
    }
    ~MultiMatcher() { for (auto& p : patterns) delete p.m; }
    MultiMatcher(const MultiMatcher&) = delete;

    // CompiledPatterns::{Empty, Single, Multi} (mod.rs:178-190): a single NEGATED pattern is Multi
    bool is_empty() const { return patterns.empty(); }
    bool is_single() const { return patterns.size() == 1 && !patterns[0].negated; }

    // match_list_multi_into (multi.rs:84-152)
    void match_list_multi_into(const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) const {
        size_t base = patterns.size();
        for (size_t i = 0; i < patterns.size(); i++) if (!patterns[i].negated) { base = i; break; }
        std::vector<Match> candidates;
        if (base != patterns.size()) patterns[base].m->match_list_into(patterns[base].m->impl, hs, lo, hi, index_offset, candidates);
        else for (size_t i = lo; i < hi; i++) candidates.push_back(Match{(u32)(index_offset + (i - lo)), 0, 0, 0});
        std::vector<u32> gathered;
        std::vector<Match> hits;
        for (size_t pi = 0; pi < patterns.size(); pi++) {
            if (pi == base || candidates.empty()) continue;
            gathered.clear();
            for (const Match& m : candidates) gathered.push_back((u32)(lo + (m.index - index_offset)));
            HaystackList sub{hs.bytes, hs.ends, gathered.size(), gathered.data()};
            hits.clear();
            patterns[pi].m->match_list_into(patterns[pi].m->impl, sub, 0, sub.n, 0, hits);
            if (patterns[pi].negated) {
                std::vector<Match> kept;
                size_t h = 0;
                for (size_t pos = 0; pos < candidates.size(); pos++) {
                    bool matched = h < hits.size() && hits[h].index == pos;
                    if (matched) h++;
                    else kept.push_back(candidates[pos]);
                }
                candidates.swap(kept);
            } else {
                std::vector<Match> next;
                for (Match hit : hits) {
                    const Match& cand = candidates[hit.index];
                    hit.index = cand.index;
                    hit.score = sat_add16(hit.score, cand.score);
                    hit.exact = (u8)(hit.exact | cand.exact);
                    next.push_back(hit);
                }
                candidates.swap(next);
            }
        }
        out.insert(out.end(), candidates.begin(), candidates.end());
    }

    void match_list_into(const HaystackList& hs, size_t lo, size_t hi, u32 index_offset, std::vector<Match>& out) const {  // mod.rs:373-392
        std::string err = Matcher::guard_against_haystack_overflow(hi - lo, index_offset);
        if (!err.empty()) throw std::runtime_error(err);
        if (is_empty()) { for (size_t i = lo; i < hi; i++) out.push_back(Match{(u32)(index_offset + (i - lo)), 0, 0, 0}); return; }
        if (is_single()) { patterns[0].m->match_list_into(patterns[0].m->impl, hs, lo, hi, index_offset, out); return; }
        match_list_multi_into(hs, lo, hi, index_offset, out);
    }

    std::vector<Match> match_list(const HaystackList& hs) const {  // mod.rs:212-222
        std::vector<Match> matches;
        match_list_into(hs, 0, hs.n, 0, matches);
        if (sort_is_reversed(config.sort)) std::reverse(matches.begin(), matches.end());
        if (!is_empty() && sort_is_by_score(config.sort)) radix_sort_matches(matches);
        return matches;
    }

    // Matcher::match_list_indices over CompiledPatterns (mod.rs:234-262), haystack order (the caller applies mod.rs:268-273).
    // Multi: match_one_indices_multi per haystack (multi.rs:56-82): a negated pattern that matches drops the haystack; every other
    // pattern must match, scores add with saturation, exact flags OR, the position lists are concatenated, sorted descending and
    // de-duplicated (patterns may share matched bytes).
    void match_list_indices(const HaystackList& hs, std::vector<Match>& out, std::vector<std::vector<u32>>& indices) const {
        std::string err = Matcher::guard_against_haystack_overflow(hs.n, 0);
        if (!err.empty()) throw std::runtime_error(err);
        if (is_empty()) { for (size_t i = 0; i < hs.n; i++) { out.push_back(Match{(u32)i, 0, 0, 0}); indices.emplace_back(); } return; }
        if (is_single()) { patterns[0].m->impl->match_list_indices(hs, out, indices); return; }
        for (size_t i = 0; i < hs.n; i++) {
            const u32 one = (u32)hs.at(i);
            HaystackList sub{hs.bytes, hs.ends, 1, &one};
            Match combined{(u32)i, 0, 0, 0};
            std::vector<u32> all;
            bool keep = true;
            for (const Compiled& p : patterns) {
                std::vector<Match> r;
                if (p.negated) {
                    p.m->match_list_into(p.m->impl, sub, 0, 1, 0, r);
                    if (!r.empty()) { keep = false; break; }
                } else {
                    std::vector<std::vector<u32>> ix;
                    p.m->impl->match_list_indices(sub, r, ix);
                    if (r.empty()) { keep = false; break; }
                    combined.score = sat_add16(combined.score, r[0].score);
                    combined.exact = (u8)(combined.exact | r[0].exact);
                    all.insert(all.end(), ix[0].begin(), ix[0].end());
                }
            }
            if (!keep) continue;
            std::sort(all.begin(), all.end(), [](u32 a, u32 b) { return a > b; });
            all.erase(std::unique(all.begin(), all.end()), all.end());
            out.push_back(combined);
            indices.push_back(all);
        }
    }

    // The reference's own oracle for this composition (tests/api_properties.rs:316-361): match every pattern on its own,
    // intersect the non-negated ones (scores add with saturation, exact flags OR), subtract the negated ones.  Index order.
    std::vector<Match> reference_composition(const HaystackList& hs) const {
        std::vector<std::vector<Match>> per(patterns.size());
        for (size_t pi = 0; pi < patterns.size(); pi++) patterns[pi].m->match_list_into(patterns[pi].m->impl, hs, 0, hs.n, 0, per[pi]);
        std::vector<size_t> cur(patterns.size(), 0);
        std::vector<Match> out;
        for (size_t i = 0; i < hs.n; i++) {
            Match combined{(u32)i, 0, 0, 0};
            bool keep = true;
            for (size_t pi = 0; pi < patterns.size(); pi++) {
                while (cur[pi] < per[pi].size() && per[pi][cur[pi]].index < i) cur[pi]++;
                const Match* hit = (cur[pi] < per[pi].size() && per[pi][cur[pi]].index == i) ? &per[pi][cur[pi]] : nullptr;
                if (patterns[pi].negated) { if (hit) keep = false; }
                else if (!hit) keep = false;
                else { combined.score = sat_add16(combined.score, hit->score); combined.exact = (u8)(combined.exact | hit->exact); }
            }
            if (keep) out.push_back(combined);
        }
        return out;
    }
};

}  // namespace fzo
