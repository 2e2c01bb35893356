// ORACLE - TEST INFRASTRUCTURE ONLY (see frizbee_oracle.hpp header).
// C ABI over the CPU restatement so pytest (ctypes) and bench.py's cpu_baseline leg can call it.
#include "frizbee_oracle.hpp"

using namespace fzo;

template <int L>
static Window pf_run(const std::string& needle, const u8* h, size_t hlen, int max_typos, bool cs, bool unicode) {
    Prefilter<L> p(needle, cs);
    if (unicode) {
        if (max_typos == 0) return p.match_haystack_unicode(h, hlen);
        if (max_typos == 1) return p.match_haystack_unicode_1_typo(h, hlen);
        if (max_typos == 2) return p.match_haystack_unicode_2_typos(h, hlen);
        return p.match_haystack_unicode_many_typos(h, hlen, (size_t)max_typos);
    }
    if (max_typos == 0) return p.match_haystack(h, hlen);
    if (max_typos == 1) return p.match_haystack_1_typo(h, hlen);
    if (max_typos == 2) return p.match_haystack_2_typos(h, hlen);
    return p.match_haystack_many_typos(h, hlen, (size_t)max_typos);
}

template <int L, typename T>
static u16 sw_run(const std::string& n, const Scoring& sc, bool cs, const u8* h, size_t hlen, bool include_prefix, bool unicode) {
    SmithWaterman<L, T> sw(n, sc, cs);
    return unicode ? sw.score_haystack_unicode(h, hlen, include_prefix) : sw.score_haystack(h, hlen, include_prefix);
}

template <int L, typename T>
static int sw_indices_run(const std::string& n, const Scoring& sc, bool cs, const u8* h, size_t hlen, size_t start_pos, bool unicode, int max_typos, std::vector<u32>& idx) {
    SmithWaterman<L, T> sw(n, sc, cs);
    return unicode ? sw.score_haystack_unicode_indices(h, hlen, start_pos, max_typos, idx) : sw.score_haystack_indices(h, hlen, start_pos, max_typos, idx);
}
template <int L, typename T>
static int sw_path_run(const std::string& n, const Scoring& sc, bool cs, const u8* h, size_t hlen, int max_typos) {
    SmithWaterman<L, T> sw(n, sc, cs);
    u16 score = sw.score_haystack(h, hlen, true);
    return sw.has_alignment_path(score, max_typos) ? (int)score : -1;
}

extern "C" {

// which lane-vector implementation this build of the oracle runs on
const char* fzo_simd_kind(void) {
#if defined(FZO_AVX512)
    return "avx512";
#else
    return "portable";
#endif
}


struct fzo_config {
    int32_t max_typos;  // -1 == None
    int32_t casing, unicode, sort;
    uint16_t scoring[9];  // match, mismatch, gap_open, gap_extend, prefix, capitalization, matching_case, exact_match, delimiter
    int32_t matching;     // Matching: 0 fuzzy, 1 exact, 2 prefix, 3 suffix, 4 substring
};

static thread_local std::string g_err;
const char* fzo_last_error() { return g_err.c_str(); }

static Scoring scoring_from(const uint16_t s[9]) {
    Scoring r;
    r.match_score = s[0]; r.mismatch_penalty = s[1]; r.gap_open_penalty = s[2]; r.gap_extend_penalty = s[3];
    r.prefix_bonus = s[4]; r.capitalization_bonus = s[5]; r.matching_case_bonus = s[6]; r.exact_match_bonus = s[7]; r.delimiter_bonus = s[8];
    return r;
}
static Config config_from(const fzo_config* c) {
    Config r;
    r.max_typos = c->max_typos; r.casing = c->casing; r.unicode = c->unicode; r.sort = c->sort;
    r.scoring = scoring_from(c->scoring);
    r.matching = c->matching;
    return r;
}

// out = {matched, start, end}.  Returns 0 on success.
int fzo_prefilter(const uint8_t* needle, size_t nlen, const uint8_t* hay, size_t hlen, int max_typos, int case_sensitive, int unicode, int lanes, uint64_t out[3]) {
    try {
        std::string n((const char*)needle, nlen);
        Window w;
        switch (lanes) {
            case 16: w = pf_run<16>(n, hay, hlen, max_typos, case_sensitive, unicode); break;
            case 32: w = pf_run<32>(n, hay, hlen, max_typos, case_sensitive, unicode); break;
            case 64: w = pf_run<64>(n, hay, hlen, max_typos, case_sensitive, unicode); break;
            default: g_err = "bad lanes"; return 1;
        }
        out[0] = w.matched; out[1] = w.start; out[2] = w.end;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// Raw `score_haystack[_unicode]` (no prefilter / trim / exact bonus).  Returns score, or -1 on error.
int fzo_sw_score(const uint8_t* needle, size_t nlen, const uint8_t* hay, size_t hlen, const uint16_t scoring[9], int case_sensitive, int include_prefix, int unicode, int lanes, int is_u8) {
    try {
        std::string n((const char*)needle, nlen);
        Scoring sc = scoring_from(scoring);
#define RUN(L) (is_u8 ? sw_run<L, u8>(n, sc, case_sensitive, hay, hlen, include_prefix, unicode) : sw_run<L, u16>(n, sc, case_sensitive, hay, hlen, include_prefix, unicode))
        switch (lanes) {
            case 8: return RUN(8);
            case 16: return RUN(16);
            case 32: return RUN(32);
            case 64: return RUN(64);
            default: g_err = "bad lanes"; return -1;
        }
#undef RUN
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// score_haystack[_unicode]_indices (smith_waterman/algo/mod.rs:49-152): returns the score, *out_n indices (reverse order) in out_idx.
// max_typos < 0 = None.
int fzo_sw_indices(const uint8_t* needle, size_t nlen, const uint8_t* hay, size_t hlen, const uint16_t scoring[9], int case_sensitive, size_t start_pos, int unicode, int lanes,
                   int is_u8, int max_typos, uint32_t* out_idx, size_t cap, size_t* out_n) {
    try {
        std::string n((const char*)needle, nlen);
        Scoring sc = scoring_from(scoring);
        std::vector<u32> idx;
        int score;
#define RUN(L) (is_u8 ? sw_indices_run<L, u8>(n, sc, case_sensitive, hay, hlen, start_pos, unicode, max_typos, idx) : sw_indices_run<L, u16>(n, sc, case_sensitive, hay, hlen, start_pos, unicode, max_typos, idx))
        switch (lanes) {
            case 8: score = RUN(8); break;
            case 16: score = RUN(16); break;
            case 32: score = RUN(32); break;
            case 64: score = RUN(64); break;
            default: g_err = "bad lanes"; return -1;
        }
#undef RUN
        if (idx.size() > cap) { g_err = "fzo_sw_indices: output too small"; return -1; }
        if (!idx.empty()) memcpy(out_idx, idx.data(), idx.size() * 4);
        *out_n = idx.size();
        return score;
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}
// score_haystack + has_alignment_path (smith_waterman/alignment.rs:24-35): the score, or -1 when no path fits the typo budget
int fzo_sw_score_typos(const uint8_t* needle, size_t nlen, const uint8_t* hay, size_t hlen, const uint16_t scoring[9], int case_sensitive, int lanes, int is_u8, int max_typos) {
    try {
        std::string n((const char*)needle, nlen);
        Scoring sc = scoring_from(scoring);
#define RUN(L) (is_u8 ? sw_path_run<L, u8>(n, sc, case_sensitive, hay, hlen, max_typos) : sw_path_run<L, u16>(n, sc, case_sensitive, hay, hlen, max_typos))
        switch (lanes) {
            case 8: return RUN(8);
            case 16: return RUN(16);
            case 32: return RUN(32);
            case 64: return RUN(64);
            default: g_err = "bad lanes"; return -2;
        }
#undef RUN
    } catch (std::exception& e) { g_err = e.what(); return -2; }
}
// Matcher::match_list_indices for one pattern (matcher/mod.rs:234-262, index order): records + per-record index lists, flattened.
// out_offsets has n_records + 1 entries.  Everything malloc'd; free with fzo_free.
static int pack_indices(const std::vector<Match>& recs, const std::vector<std::vector<u32>>& idx, Match** out, size_t* out_len, uint32_t** out_idx, uint64_t** out_offsets) {
    size_t total = 0;
    for (auto& v : idx) total += v.size();
    *out = (Match*)malloc(std::max<size_t>(recs.size(), 1) * sizeof(Match));
    *out_idx = (uint32_t*)malloc(std::max<size_t>(total, 1) * 4);
    *out_offsets = (uint64_t*)malloc((recs.size() + 1) * 8);
    size_t off = 0;
    for (size_t i = 0; i < recs.size(); i++) {
        (*out)[i] = recs[i];
        (*out_offsets)[i] = off;
        if (!idx[i].empty()) memcpy(*out_idx + off, idx[i].data(), idx[i].size() * 4);
        off += idx[i].size();
    }
    (*out_offsets)[recs.size()] = off;
    *out_len = recs.size();
    return 0;
}

// Matcher::match_list_indices before its ordering step (haystack order): records, flat positions, offsets[n + 1]
int fzo_match_list_indices(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, Match** out, size_t* out_len, uint32_t** out_idx, uint64_t** out_offsets) {
    try {
        HaystackList hs{bytes, ends, n};
        Matcher* mm = (Matcher*)m;
        std::vector<Match> recs;
        std::vector<std::vector<u32>> idx;
        if (!mm->empty) mm->impl->match_list_indices(hs, recs, idx);
        else for (size_t i = 0; i < n; i++) { recs.push_back(Match{(u32)i, 0, 0, 0}); idx.emplace_back(); }
        return pack_indices(recs, idx, out, out_len, out_idx, out_offsets);
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// the same over CompiledPatterns::Multi (match_one_indices_multi, multi.rs:56-82)
int fzo_multi_match_list_indices(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, Match** out, size_t* out_len, uint32_t** out_idx, uint64_t** out_offsets) {
    try {
        HaystackList hs{bytes, ends, n};
        std::vector<Match> recs;
        std::vector<std::vector<u32>> idx;
        ((MultiMatcher*)m)->match_list_indices(hs, recs, idx);
        return pack_indices(recs, idx, out, out_len, out_idx, out_offsets);
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// match_greedy: returns score or -1 for None
int fzo_greedy(const uint8_t* needle, size_t nlen, const uint8_t* hay, size_t hlen, const uint16_t scoring[9], int case_sensitive, int include_prefix) {
    u16 s;
    return match_greedy(std::string((const char*)needle, nlen), hay, hlen, scoring_from(scoring), case_sensitive, include_prefix, s) ? (int)s : -1;
}

int fzo_score_fits_in_u8(size_t needle_len, const uint16_t scoring[9]) { return score_fits_in_u8(needle_len, scoring_from(scoring)); }
int fzo_max_needle_len(const uint16_t scoring[9]) {  // lib.rs:482-484
    Scoring s = scoring_from(scoring);
    return (int)(sat_sub16(0xFFFF, max_one_time_bonus(s)) / max_per_char_bonus(s));
}

void* fzo_matcher_create(const fzo_config* cfg, const uint8_t* needle, size_t nlen, int pf_lanes, int sw_lanes_u8, int sw_lanes_u16) {
    try {
        return new Matcher(std::string((const char*)needle, nlen), config_from(cfg), pf_lanes, sw_lanes_u8, sw_lanes_u16);
    } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
void fzo_matcher_free(void* m) { delete (Matcher*)m; }
int fzo_matcher_info(void* m, int out[3]) {
    Matcher* mm = (Matcher*)m;
    out[0] = mm->pf_lanes; out[1] = mm->sw_lanes; out[2] = mm->use_u8;
    return 0;
}

// threads < 0 -> match_list, else match_list_parallel(threads). Result is malloc'd; free with fzo_free.
int fzo_match_list(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, long threads, Match** out, size_t* out_len) {
    try {
        HaystackList hs{bytes, ends, n};
        Matcher* mm = (Matcher*)m;
        std::vector<Match> r = threads < 0 ? mm->match_list(hs) : mm->match_list_parallel(hs, (size_t)threads);
        *out_len = r.size();
        *out = (Match*)malloc(std::max<size_t>(r.size(), 1) * sizeof(Match));
        if (!r.empty()) memcpy(*out, r.data(), r.size() * sizeof(Match));
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}
// Same as fzo_match_list but discards the result (timing leg): returns number of matches via out_len.
int fzo_match_list_count(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, long threads, size_t* out_len) {
    try {
        HaystackList hs{bytes, ends, n};
        Matcher* mm = (Matcher*)m;
        std::vector<Match> r = threads < 0 ? mm->match_list(hs) : mm->match_list_parallel(hs, (size_t)threads);
        *out_len = r.size();
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}
// ---- multi-pattern composition (matcher/multi.rs; fuzzy patterns) ----
struct fzo_pattern {
    const uint8_t* needle;
    size_t needle_len;
    int32_t negated, has_max_typos, max_typos, casing, unicode, has_scoring;  // casing / unicode: -1 = inherit
    uint16_t scoring[9];
    int32_t matching;  // -1 = inherit
};
void* fzo_multi_create(const fzo_config* cfg, const fzo_pattern* pats, size_t npats, int pf_lanes, int sw_lanes_u8, int sw_lanes_u16) {
    try {
        std::vector<PatternSpec> specs;
        for (size_t i = 0; i < npats; i++) {
            PatternSpec sp;
            sp.needle.assign((const char*)pats[i].needle, pats[i].needle_len);
            sp.negated = pats[i].negated != 0;
            sp.has_max_typos = pats[i].has_max_typos != 0;
            sp.max_typos = pats[i].max_typos;
            sp.casing = pats[i].casing;
            sp.unicode = pats[i].unicode;
            sp.has_scoring = pats[i].has_scoring != 0;
            sp.matching = pats[i].matching;
            if (sp.has_scoring) {
                const uint16_t* v = pats[i].scoring;
                sp.scoring = Scoring{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]};
            }
            specs.push_back(sp);
        }
        return new MultiMatcher(specs, config_from(cfg), pf_lanes, sw_lanes_u8, sw_lanes_u16);
    } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}
void fzo_multi_free(void* m) { delete (MultiMatcher*)m; }
// mode 0: Matcher::match_list over CompiledPatterns (sequential narrowing, multi.rs:84-152, then the ordering step);
// mode 1: the reference's test oracle for it (per-pattern lists composed per haystack, index order).
int fzo_multi_match_list(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, int mode, Match** out, size_t* out_len) {
    try {
        HaystackList hs{bytes, ends, n};
        MultiMatcher* mm = (MultiMatcher*)m;
        std::vector<Match> r = mode == 0 ? mm->match_list(hs) : mm->reference_composition(hs);
        *out_len = r.size();
        *out = (Match*)malloc(std::max<size_t>(r.size(), 1) * sizeof(Match));
        if (!r.empty()) memcpy(*out, r.data(), r.size() * sizeof(Match));
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}

// Pattern::parse_query -> flat text, one line per pattern: "<negated 0|1> <matching -1..4> <needle bytes as hex>\n"
int fzo_parse_query(const uint8_t* query, size_t qlen, char* out, size_t out_cap) {
    try {
        std::vector<PatternSpec> ps = parse_query(std::string((const char*)query, qlen));
        std::string r;
        for (const PatternSpec& p : ps) {
            r += std::to_string((int)p.negated) + " " + std::to_string(p.matching) + " ";
            static const char* hex = "0123456789abcdef";
            for (unsigned char c : p.needle) { r.push_back(hex[c >> 4]); r.push_back(hex[c & 15]); }
            r.push_back('\n');
        }
        if (r.size() + 1 > out_cap) { g_err = "fzo_parse_query: output buffer too small"; return -1; }
        memcpy(out, r.c_str(), r.size() + 1);
        return (int)ps.size();
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// Timing leg: score every haystack on `threads` workers, no ordering step (Matcher::score_parallel_unordered).
int fzo_score_count_unordered(void* m, const uint8_t* bytes, const uint64_t* ends, size_t n, long threads, size_t* out_len) {
    try {
        HaystackList hs{bytes, ends, n};
        *out_len = ((Matcher*)m)->score_parallel_unordered(hs, (size_t)threads);
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return 1; }
}
void fzo_free(void* p) { free(p); }

void fzo_radix_sort(Match* matches, size_t n) {
    std::vector<Match> v(matches, matches + n);
    radix_sort_matches(v);
    if (n) memcpy(matches, v.data(), n * sizeof(Match));
}
// runs: concatenated, run_lens[k] entries each; out must hold the total.
void fzo_k_merge(int order, const Match* runs, const size_t* run_lens, size_t nruns, Match* out) {
    std::vector<std::vector<Match>> rs;
    size_t off = 0;
    for (size_t k = 0; k < nruns; k++) { rs.emplace_back(runs + off, runs + off + run_lens[k]); off += run_lens[k]; }
    auto merged = k_merge_matches_by(order, rs);
    if (!merged.empty()) memcpy(out, merged.data(), merged.size() * sizeof(Match));
}

// Needle preparation probes (for host-logic tests)
int fzo_respects_case_for(int casing, const uint8_t* needle, size_t nlen) {
    try { return respects_case_for(casing, std::string((const char*)needle, nlen)); } catch (std::exception& e) { g_err = e.what(); return -1; }
}
// out: per scalar 9 bytes: chars[4], flipped[4], len. Returns scalar count or -1.
int fzo_case_needle_unicode(const uint8_t* needle, size_t nlen, int case_sensitive, uint8_t* out, size_t cap) {
    try {
        auto v = case_needle_unicode(std::string((const char*)needle, nlen), case_sensitive);
        for (size_t i = 0; i < v.size() && i < cap; i++) {
            memcpy(out + 9 * i, v[i].chars, 4);
            memcpy(out + 9 * i + 4, v[i].flipped, 4);
            out[9 * i + 8] = (uint8_t)v[i].len;
        }
        return (int)v.size();
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
