// Host side of the MI355X frizbee backend: the C ABI of include/frizbee_hip.h.
//
// Mirrors, in C++ (the reference is compiled Rust and no Rust toolchain exists in the build image):
//   Matcher::new / compile / get_backend      src/matcher/mod.rs:90-111, 178-204, 448-498
//   MatcherImpl::new                          src/matcher/algo.rs:57-71, 311-325
//   case_needle / case_needle_unicode         src/prefilter/mod.rs:49-96
//   score_fits_in_u8 / Scoring guards         src/smith_waterman/mod.rs:92-116, src/lib.rs:480-538
//   match_list / match_list_parallel post-steps  src/matcher/mod.rs:212-222, src/matcher/parallel.rs:18-89
//   radix_sort_matches / k_merge              src/sort.rs:6-40, src/k_merge.rs:56-170
// All scoring work happens in the gfx950 kernels; there is NO CPU fallback - without a HIP device every
// scoring entry point fails with FZB_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>

#include "host_internal.h"

namespace {
#include "unicode_case_table.inc"

thread_local std::string g_err;
inline int fail(int code, const std::string& msg) { return fzb_fail(code, msg); }

// ---- UTF-8 / case helpers ---------------------------------------------------------------------------
bool decode_utf8(const u8* s, size_t n, std::vector<u32>& out) {
    size_t i = 0;
    while (i < n) {
        u8 b = s[i];
        u32 cp;
        int len;
        if (b < 0x80) { cp = b; len = 1; }
        else if ((b & 0xE0) == 0xC0) { cp = b & 0x1F; len = 2; }
        else if ((b & 0xF0) == 0xE0) { cp = b & 0x0F; len = 3; }
        else if ((b & 0xF8) == 0xF0) { cp = b & 0x07; len = 4; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; k++) {
            if ((s[i + k] & 0xC0) != 0x80) return false;
            cp = (cp << 6) | (s[i + k] & 0x3F);
        }
        // what a Rust &str can never hold: overlong forms, UTF-16 surrogates, scalars above U+10FFFF
        static const u32 min_cp[5] = {0, 0, 0x80, 0x800, 0x10000};
        if (cp < min_cp[len] || (cp >= 0xD800 && cp <= 0xDFFF) || cp > 0x10FFFF) return false;
        out.push_back(cp);
        i += len;
    }
    return true;
}
int encode_utf8(u32 cp, u8 out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (cp < 0x80) { out[0] = (u8)cp; return 1; }
    if (cp < 0x800) { out[0] = (u8)(0xC0 | (cp >> 6)); out[1] = (u8)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) { out[0] = (u8)(0xE0 | (cp >> 12)); out[1] = (u8)(0x80 | ((cp >> 6) & 0x3F)); out[2] = (u8)(0x80 | (cp & 0x3F)); return 3; }
    out[0] = (u8)(0xF0 | (cp >> 18)); out[1] = (u8)(0x80 | ((cp >> 12) & 0x3F)); out[2] = (u8)(0x80 | ((cp >> 6) & 0x3F)); out[3] = (u8)(0x80 | (cp & 0x3F));
    return 4;
}
bool is_uppercase(u32 cp) {  // char::is_uppercase
    if (cp < 0x80) return cp >= 'A' && cp <= 'Z';
    const auto* end = FZB_UPPER_RANGES + FZB_UPPER_RANGES_LEN;
    const auto* it = std::upper_bound(FZB_UPPER_RANGES, end, cp, [](u32 v, const unsigned int (&r)[2]) { return v < r[0]; });
    if (it == FZB_UPPER_RANGES) return false;
    --it;
    return cp >= (*it)[0] && cp <= (*it)[1];
}
u32 flip_same_width(u32 cp) {  // the flip `case_needle_unicode` keeps (single scalar, same UTF-8 width), else cp
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : (cp >= 'a' && cp <= 'z') ? cp - 32 : cp;
    const auto* end = FZB_CASE_FLIP + FZB_CASE_FLIP_LEN;
    const auto* it = std::lower_bound(FZB_CASE_FLIP, end, cp, [](const unsigned int (&r)[2], u32 v) { return r[0] < v; });
    return (it != end && (*it)[0] == cp) ? (*it)[1] : cp;
}

// ---- Scoring guards (src/lib.rs:480-538, src/smith_waterman/mod.rs:92-116) --------------------------------
u16 sadd16(u32 a, u32 b) { return (u16)std::min<u32>(a + b, 0xFFFF); }
u16 ssub16(u32 a, u32 b) { return (u16)(a > b ? a - b : 0); }
u16 max_per_char_bonus(const fzb_scoring& s) {
    u16 bonus = std::max(s.delimiter_bonus, s.capitalization_bonus);
    u16 amortized = std::max<u16>((u16)((bonus + 1) / 2), ssub16(bonus, s.gap_open_penalty));
    return sadd16(amortized, s.matching_case_bonus);
}
u16 max_one_time_bonus(const fzb_scoring& s) {
    u16 bonus = std::max(s.delimiter_bonus, s.capitalization_bonus);
    u16 amortized = std::max<u16>((u16)((bonus + 1) / 2), ssub16(bonus, s.gap_open_penalty));
    return (u16)(bonus - amortized);
}
// Scoring::guard_against_score_overflow (src/lib.rs:506-537).  Fuzzy callers pass the amortised bonuses (src/matcher/algo.rs:311-325),
// the literal matcher its own (src/literal/algo.rs:314-322).
std::string overflow_guard(const fzb_scoring& s, size_t rows, int bonus_per_char = -1, int one_time = -1) {
    const u16 bpc = bonus_per_char < 0 ? max_per_char_bonus(s) : (u16)bonus_per_char;
    const u16 ot = one_time < 0 ? max_one_time_bonus(s) : (u16)one_time;
    u16 max_per_char = sadd16(s.match_score, bpc);
    if (max_per_char == 0) return "";
    u16 headroom = ssub16(ssub16(ssub16(ssub16(0xFFFF, s.prefix_bonus), s.exact_match_bonus), s.mismatch_penalty), ot);
    u16 max_needle_len = (u16)(headroom / max_per_char);
    if (rows > (size_t)max_needle_len)
        return "needle too long and could overflow the u16 score: " + std::to_string(rows) + " > " + std::to_string(max_needle_len);
    size_t max_gap = 32 * (size_t)s.gap_extend_penalty + (size_t)s.gap_open_penalty;
    if (max_gap > 0xFFFF) return "gap penalties too large and could overflow the u16 score: " + std::to_string(max_gap) + " > 65535";
    return "";
}
size_t max_matrix_score(const fzb_scoring& s, size_t needle_len) {
    size_t max_per_char = (size_t)s.match_score + (size_t)max_per_char_bonus(s);
    return max_per_char * needle_len + (size_t)max_one_time_bonus(s) + (size_t)s.prefix_bonus;
}
bool fits_in_u8(size_t needle_len, const fzb_scoring& s) {
    size_t max_constant = std::max<size_t>({(size_t)s.match_score + (size_t)s.mismatch_penalty, s.gap_open_penalty, s.gap_extend_penalty,
                                            s.matching_case_bonus, s.capitalization_bonus, s.delimiter_bonus, s.prefix_bonus});
    if (max_constant > 255) return false;
    if (64 * (size_t)s.gap_extend_penalty + (size_t)s.gap_open_penalty > 255) return false;
    return max_matrix_score(s, needle_len) + (size_t)s.mismatch_penalty <= 255;
}

// Which (prefilter lanes, score lanes) pair `Matcher::get_backend` picks on this host (src/matcher/mod.rs:448-498)
void detect_host_lanes(bool use_u8, int& pf, int& sw) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_cpu_init();
    const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
    const bool vbmi = __builtin_cpu_supports("avx512vbmi");
    const bool avx2 = __builtin_cpu_supports("avx2");
    const bool sse = __builtin_cpu_supports("sse2") && __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("sse4.1");
    if (use_u8) {
        if (avx512 && bmi && vbmi) { pf = 64; sw = 64; return; }
        if (avx2) { pf = 32; sw = 32; return; }
        if (sse) { pf = 16; sw = 16; return; }
        pf = 16; sw = 16;  // scalar u8
    } else {
        if (avx512 && bmi) { pf = 64; sw = 32; return; }
        if (avx2) { pf = 32; sw = 16; return; }
        if (sse) { pf = 16; sw = 8; return; }
        pf = 16; sw = 8;  // scalar
    }
#else
    pf = 16;
    sw = use_u8 ? 16 : 8;  // NEON / scalar
#endif
}

hipError_t dev_alloc(void** p, size_t bytes) { return fzb_dev_alloc(p, bytes); }

}  // namespace

int fzb_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
void fzb_clear_error() { g_err.clear(); }

extern "C" {

const char* fzb_last_error(void) { return g_err.c_str(); }

}  // extern "C"
// ---- environment switches: parsed once (knobs.h) --------------------------------------------------------------------------------
namespace {
FzbKnobs parse_knobs() {
    FzbKnobs k;
    auto on = [](const char* name) { const char* e = getenv(name); return e != nullptr && atoi(e) != 0; };
    auto set = [](const char* name) { return getenv(name) != nullptr; };
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    k.debug_sync = set("FZB_DEBUG_SYNC");
    k.typo_exact_window = on("FZB_TYPO_EXACT_WINDOW");
    k.no_filter_view = getenv("FZB_FILTER_VIEW") != nullptr && atoi(getenv("FZB_FILTER_VIEW")) == 0;
    k.verify_promises = num("FZB_VERIFY_PROMISES", 1) != 0;
    k.no_lcs_dfa = set("FZB_NO_LCS_DFA");
    k.no_cdfa = set("FZB_NO_CDFA");
    k.no_dp_classes = set("FZB_NO_DP_CLASSES");
    k.no_fused_classify = set("FZB_NO_FUSED_CLASSIFY");
    k.no_dp_cfm = set("FZB_NO_DP_CFM");
    k.no_dp_cfu = set("FZB_NO_DP_CFU");
    k.unicode_multi = num("FZB_UNICODE_MULTI", -1);
    k.park_lds_kb = std::max(0, std::min(60, num("FZB_PARK_LDS_KB", 37)));
    k.coop_below = num("FZB_COOP_BELOW", -1);
    k.shard_gather_copy = getenv("FZB_SHARD_GATHER") != nullptr && !strcmp(getenv("FZB_SHARD_GATHER"), "copy");
    k.shard_inline = num("FZB_SHARD_INLINE", -1);
    k.spin_wait_us = num("FZB_SPIN_WAIT_US", 1000);
    return k;
}
FzbKnobs& knobs_storage() {
    static FzbKnobs k = parse_knobs();
    return k;
}
}  // namespace
const FzbKnobs& fzb_knobs() { return knobs_storage(); }
extern "C" {
// test hook: re-read the environment (tests/test_gpu_knobs.py switches paths inside one process; matchers created BEFORE the call keep
// what was decided at their creation - the LCS automaton, the unicode scorer's form)
void fzb_debug_reload_knobs(void) { knobs_storage() = parse_knobs(); }

void fzb_config_default(fzb_config* out) {
    if (!out) return;
    memset(out, 0, sizeof(*out));
    out->max_typos = 0;
    out->casing = FZB_CASE_SMART;
    out->unicode = FZB_UNICODE_SMART;
    out->sort = FZB_SORT_SCORE_THEN_INDEX_ASC;
    out->scoring = fzb_scoring{12, 6, 5, 1, 12, 4, 4, 8, 4};  // src/const.rs:1-10
    out->matching = FZB_MATCH_FUZZY;
}

static void free_workspace(Workspace& w) {
    // (table / dfa / uni_dfa / lcs_dfa / cdfa point into tables_blob)
    void* ptrs[] = {w.bitmap, w.tile_counts, w.surv_idx, w.win, w.overflow, w.dp_scratch, w.sort_tmp, w.sort_hist, w.bitmap2, w.tile_counts2, w.items2, w.win2, w.counters, w.tables_blob,
                    w.trace_cells, w.bitmap_m, w.tile_counts_m, w.marg_list, w.reject_bits, w.tile_rejects, w.rej_prefix, w.cls_win, w.cls_lists};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (w.tables_ev_pending) (void)hipEventSynchronize(w.tables_ev);
    if (w.tables_host) (void)hipHostFree(w.tables_host);
    if (w.tables_ev) (void)hipEventDestroy(w.tables_ev);
    w = Workspace{};
}

// The matcher's byte tables on the device: [filter table 2 KB | subsequence DFA | unicode DFA | LCS automaton | class-composite automaton], each
// with room for any needle's (fzb_matcher_set_pattern re-uploads in place).  One copy out of the pinned staging buffer, asynchronous on `st`
// (the stream the query's kernels follow on) - or synchronous when there is no stream to order it on (fzb_matcher_reserve).
namespace {
constexpr size_t TB_TABLE = 0, TB_DFA = 2048, TB_SLOT = 256 * 256 + 128, TB_UNI = TB_DFA + TB_SLOT, TB_LCS = TB_UNI + TB_SLOT, TB_CDFA = TB_LCS + TB_SLOT,
                 TB_TOTAL = TB_CDFA + 256 + 16384 + 128;
}
static int upload_tables(fzb_matcher* m, hipStream_t st, bool have_stream) {
    Workspace& w = m->ws;
    if (w.tables_ev_pending) {  // (the previous copy out of the staging buffer: long done unless two needle changes follow each other without a query's end between)
        HIPCHK(hipEventSynchronize(w.tables_ev));
        w.tables_ev_pending = false;
    }
    size_t end = TB_DFA;
    memcpy(w.tables_host + TB_TABLE, m->table.data(), 256 * 8);
    auto put = [&](size_t off, const std::vector<u8>& v) {
        if (v.empty()) return;
        memcpy(w.tables_host + off, v.data(), v.size());
        end = std::max(end, off + v.size());
    };
    put(TB_DFA, m->dfa);
    put(TB_UNI, m->uni_dfa);
    put(TB_LCS, m->lcs_dfa);
    put(TB_CDFA, m->cdfa);
    if (have_stream) {
        HIPCHK(hipMemcpyAsync(w.tables_blob, w.tables_host, end, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(w.tables_ev, st));
        w.tables_ev_pending = true;
    } else {
        HIPCHK(hipMemcpy(w.tables_blob, w.tables_host, end, hipMemcpyHostToDevice));
    }
    w.tables_stale = false;
    return FZB_OK;
}

// ---- fzb_matcher_create, piece by piece ------------------------------------------------------------------------------------------
// NeedleDev's scalars: what `Prefilter::new` / `SmithWaterman::new` precompute (src/prefilter/algo/mod.rs:30-42, src/smith_waterman/algo/mod.rs:21-42)
static void fill_needle_scalars(fzb_matcher* m, size_t needle_len, size_t n_scalars) {
    const fzb_config& config = m->config;
    const fzb_scoring& sc = config.scoring;
    NeedleDev& nd = m->nd;
    memset(&nd, 0, sizeof(nd));
    nd.rows = m->rows;
    nd.nbytes = (int)needle_len;
    nd.max_typos = config.max_typos;
    nd.min_haystack_len = config.max_typos < 0 ? 0 : (int)(n_scalars > (size_t)config.max_typos ? n_scalars - (size_t)config.max_typos : 0);  // algo.rs:62-65
    nd.unicode = m->unicode;
    nd.lane_mask = m->use_u8 ? 0xFF : 0xFFFF;
    nd.match_plus_mismatch = sadd16(sc.match_score, sc.mismatch_penalty);
    nd.mismatch = sc.mismatch_penalty;
    nd.gex = sc.gap_extend_penalty;
    nd.gopm = ssub16(sc.gap_open_penalty, sc.gap_extend_penalty);
    nd.prefix = sc.prefix_bonus;
    nd.capitalization = sc.capitalization_bonus;
    nd.matching_case = sc.matching_case_bonus;
    nd.exact_bonus = sc.exact_match_bonus;
    nd.delimiter = sc.delimiter_bonus;
    nd.match_score = sc.match_score;
    nd.gap_open = sc.gap_open_penalty;
}
static u8 flip_ascii_byte(const fzb_matcher* m, u8 c) { return m->case_sensitive ? c : (c >= 'a' && c <= 'z') ? (u8)(c - 32) : (c >= 'A' && c <= 'Z') ? (u8)(c + 32) : c; }

// A needle beyond NeedleDev's by-value arrays: the arrays go to one host blob (uploaded on first use), the scalars to NeedleLongDev, and the
// stage configuration is "lane-exact prefilter kernel (or the streaming subsequence automaton) first, wave- or thread-per-window scorer"
static void build_long_needle(fzb_matcher* m, const uint8_t* needle_utf8, size_t needle_len, const std::vector<u32>& cps) {
    const fzb_config* config = &m->config;
    const fzb_scoring& sc = config->scoring;
    NeedleDev& nd = m->nd;
    auto flip_ascii = [&](u8 c) { return flip_ascii_byte(m, c); };
    // [raw | c | f | uc | uf | ulen], 16-byte aligned sections (device pointers are set when the blob is uploaded)
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t nb = needle_len, nr = cps.size();
    m->long_off_c = al(nb);
    m->long_off_f = m->long_off_c + al(nb);
    m->long_off_uc = m->long_off_f + al(nb);
    m->long_off_uf = m->long_off_uc + al(4 * nr);
    m->long_off_ulen = m->long_off_uf + al(4 * nr);
    m->long_blob_host.assign(m->long_off_ulen + al(nr) + 16, 0);
    u8* blob = m->long_blob_host.data();
    for (size_t i = 0; i < nb; i++) {  // case_needle (src/prefilter/mod.rs:49-65)
        blob[i] = needle_utf8[i];
        blob[m->long_off_c + i] = needle_utf8[i];
        blob[m->long_off_f + i] = flip_ascii(needle_utf8[i]);
    }
    for (size_t i = 0; i < nr; i++) {  // case_needle_unicode (src/prefilter/mod.rs:71-96)
        blob[m->long_off_ulen + i] = (u8)encode_utf8(cps[i], blob + m->long_off_uc + 4 * i);
        encode_utf8(m->case_sensitive ? cps[i] : flip_same_width(cps[i]), blob + m->long_off_uf + 4 * i);
    }
    NeedleLongDev& l = m->ndl;
    memset(&l, 0, sizeof(l));
    l.rows = nd.rows; l.nbytes = nd.nbytes; l.max_typos = nd.max_typos; l.min_haystack_len = nd.min_haystack_len; l.unicode = nd.unicode; l.lane_mask = nd.lane_mask;
    l.match_plus_mismatch = nd.match_plus_mismatch; l.mismatch = nd.mismatch; l.gex = nd.gex; l.gopm = nd.gopm;
    l.prefix = nd.prefix; l.capitalization = nd.capitalization; l.matching_case = nd.matching_case; l.exact_bonus = nd.exact_bonus; l.delimiter = nd.delimiter;
    l.match_score = nd.match_score; l.gap_open = nd.gap_open;
    // stage configuration: no streaming filter; either no prefilter at all or the lane-exact prefilter kernel as the first stage
    const int k = config->max_typos;
    LaunchCfg& lc = m->lc;
    lc.filter_mode = 0;
    lc.filter_exact = 0;  // (sizes the second-level arrays the lane-exact prefilter writes)
    lc.window_mode = (k < 0 || k >= m->rows) ? 2 : 0;
    lc.pad_ok = lc.cf_ok = lc.cfm_ok = 0;
    // (the biased gap scan of dp_multi_chunk, as for short needles below: the largest biased value stays inside 16 bits)
    lc.bias_ok = max_matrix_score(sc, (size_t)m->rows) + (size_t)sc.mismatch_penalty + 130 * (size_t)sc.gap_extend_penalty + 64 <= 0xFFFF;
    // dp_cfm.h's arithmetic for the four-lanes-per-window scorer (k2d_dp_long_quad; set_scorer_forms' condition with this needle's rows in the
    // bias: lanes up to 3/2 chunks + rows + 1 of gap_extend)
    lc.cfm_ok = !m->unicode && m->rows <= FZB_LONG_LDS_ROWS && 2 * (u32)sc.gap_extend_penalty <= (u32)sc.mismatch_penalty &&
                max_matrix_score(sc, (size_t)m->rows) + (size_t)sc.mismatch_penalty + (137 + (size_t)m->rows) * (size_t)sc.gap_extend_penalty + 64 < 0x7C00;
    m->long_upper = false;
    if (!m->unicode)
        for (size_t i = 0; i < nb; i++) m->long_upper = m->long_upper || (blob[m->long_off_c + i] >= 'A' && blob[m->long_off_c + i] <= 'Z');
    m->table.assign(256, 0);
    // 0 typos, ASCII: accept <=> the needle is a case-folded ordered subsequence (src/prefilter/algo/ascii.rs:6-54; lane-width independent),
    // and that automaton has rows + 1 states whatever the needle's length: up to 200 rows its table fits a workgroup's LDS (58 KB) and the
    // STREAMING filter decides the list (round 5; rounds 3-4 ran the chunked prefilter over every haystack: 0.75 of the bench row's 9.7 ms)
    if (!m->unicode && !m->literal_mode && k == 0 && m->rows <= 200) {
        m->long_dfa = true;
        bool used[256] = {false};
        lc.pad_ok = 1;
        for (size_t i = 0; i < nb; i++) {
            used[blob[m->long_off_c + i]] = used[blob[m->long_off_f + i]] = true;
            if (needle_utf8[i] == 0) lc.pad_ok = 0;
        }
        lc.dead_byte = 0;
        for (int b = 255; b >= 0; b--)
            if (!used[b]) { lc.dead_byte = (u32)b; break; }
        m->dfa.assign((size_t)(m->rows + 1) * 256, 0);
        for (int st = 0; st <= m->rows; st++)
            for (int b = 0; b < 256; b++)
                m->dfa[(size_t)st * 256 + b] = (u8)((st < m->rows && (b == blob[m->long_off_c + (size_t)st] || b == blob[m->long_off_f + (size_t)st])) ? st + 1 : st);
    }
}

// The streaming filter's byte tables: `table[b]` = needle rows byte b can match (the LCS filter's M), the dead byte, the ordered-subsequence
// DFA (state s = rows matched so far) - or, for the literal substring mode on the ASCII path, the needle's Knuth-Morris-Pratt automaton
static void build_filter_tables(fzb_matcher* m) {
    const NeedleDev& nd = m->nd;
    LaunchCfg& lc = m->lc;
    m->table.assign(256, 0);
    for (int r = 0; r < m->rows; r++) {
        if (m->unicode) {  // a scalar can only match where its LAST byte matches (either case): conservative
            m->table[nd.uc[r][nd.ulen[r] - 1]] |= (u64)1 << r;
            m->table[nd.uf[r][nd.ulen[r] - 1]] |= (u64)1 << r;
        } else {
            m->table[nd.c[r]] |= (u64)1 << r;
            m->table[nd.f[r]] |= (u64)1 << r;
        }
    }
    lc.dead_byte = 0;
    for (int b = 255; b >= 0; b--)
        if (m->table[b] == 0) { lc.dead_byte = (u32)b; break; }
    // ordered-subsequence DFA: state s = rows matched so far; a byte that can match row s advances it
    m->dfa.assign((size_t)(m->rows + 1) * 256, 0);
    for (int st = 0; st <= m->rows; st++)
        for (int b = 0; b < 256; b++) m->dfa[(size_t)st * 256 + b] = (u8)((st < m->rows && ((m->table[b] >> st) & 1)) ? st + 1 : st);
    if (m->literal_mode == FZB_MATCH_SUBSTRING && !m->unicode) {
        // Substring accept on the ASCII path = "the text drives the needle's Knuth-Morris-Pratt automaton into its final state":
        // the same table shape, so the streaming DFA filter kernels run it unchanged.  Position k matches byte b iff b is
        // needle[k] or its case flip; both case forms of a needle byte take the automaton to the same state (the flip is
        // an involution on every position's byte set), so the usual single restart state works for the folded alphabet.
        const int n = m->rows;
        auto hit = [&](int k, int b) { return b == nd.c[k] || b == nd.f[k]; };
        for (int b = 0; b < 256; b++) m->dfa[b] = (u8)(hit(0, b) ? 1 : 0);
        int x = 0;  // restart state: where the automaton is after reading needle[1..k)
        for (int k = 1; k < n; k++) {
            for (int b = 0; b < 256; b++) m->dfa[(size_t)k * 256 + b] = (u8)(hit(k, b) ? k + 1 : m->dfa[(size_t)x * 256 + b]);
            x = m->dfa[(size_t)x * 256 + nd.c[k]];
        }
        for (int b = 0; b < 256; b++) m->dfa[(size_t)n * 256 + b] = (u8)n;  // found: absorbing
    }
}

static void build_unicode_dfa(fzb_matcher* m) {
    const fzb_config* config = &m->config;
    const NeedleDev& nd = m->nd;
    const LaunchCfg& lc = m->lc;
    // Unicode path, 0 typos: the prefilter accepts iff the needle's scalars occur, in order, at increasing byte positions, each as its own
    // bytes or as the bytes of its same-width case flip (src/prefilter/algo/unicode.rs:118-219; the same at 16 / 32 / 64 lanes - checked
    // against the oracle by tests/test_host_abi.py::test_unicode_dfa_is_the_unicode_prefilter).  That is a byte-level DFA: state = (scalars
    // matched, bytes of the current scalar matched, which of the two variants are still alive); a mismatch inside a scalar falls back to
    // "is this byte the scalar's first byte" (UTF-8 lead bytes never occur inside a scalar, so no longer border exists).  With <= 226
    // states it runs in the streaming DFA filter kernels unchanged and replaces superset filter + lane-exact window pass + second compaction.
    m->uni_dfa_states = 0;
    if (m->unicode && !m->literal_mode && config->max_typos == 0 && m->rows >= 1 && lc.filter_mode == 1) {
        struct St { int i, k, alive; };
        std::vector<St> states;
        auto find = [&](int i, int k, int alive) {
            for (size_t q = 0; q < states.size(); q++)
                if (states[q].i == i && states[q].k == k && states[q].alive == alive) return (int)q;
            states.push_back(St{i, k, alive});
            return (int)states.size() - 1;
        };
        std::vector<std::vector<int>> trans;
        find(0, 0, 3);
        bool ok = true;
        for (size_t q = 0; q < states.size() && ok; q++) {
            const St cur = states[q];
            std::vector<int> row(256, (int)q);
            if (cur.i < m->rows) {
                const u8* va = nd.uc[cur.i];
                const u8* vb = nd.uf[cur.i];
                const int len = nd.ulen[cur.i];
                auto from_start = [&](int b) {  // state (i, 0) reading b
                    const int alive = (va[0] == b ? 1 : 0) | (vb[0] == b ? 2 : 0);
                    if (!alive) return find(cur.i, 0, 3);
                    return len == 1 ? find(cur.i + 1, 0, 3) : find(cur.i, 1, alive);
                };
                for (int b = 0; b < 256; b++) {
                    if (cur.k == 0) { row[b] = from_start(b); continue; }
                    const int alive = ((cur.alive & 1) && va[cur.k] == b ? 1 : 0) | ((cur.alive & 2) && vb[cur.k] == b ? 2 : 0);
                    if (alive) row[b] = cur.k + 1 == len ? find(cur.i + 1, 0, 3) : find(cur.i, cur.k + 1, alive);
                    else row[b] = from_start(b);
                }
            }  // i == rows: accepting, absorbing
            trans.push_back(row);
            if (states.size() > 226) ok = false;  // 226 x 288 bytes (dfa_lds.h's row stride) + the tile counter fit the 64 KiB of dynamic LDS
        }
        if (ok) {
            // renumber so that the accepting state is the LAST one (the kernels test `state == number of states - 1`)
            const int ns = (int)states.size();
            int acc = -1;
            for (int q = 0; q < ns; q++)
                if (states[q].i == m->rows) acc = q;
            std::vector<int> renum(ns);
            for (int q = 0, nx = 0; q < ns; q++) renum[q] = q == acc ? ns - 1 : nx++;
            m->uni_dfa.assign((size_t)ns * 256, 0);
            for (int q = 0; q < ns; q++)
                for (int b = 0; b < 256; b++) m->uni_dfa[(size_t)renum[q] * 256 + b] = (u8)renum[trans[q][b]];
            m->uni_dfa_states = ns;  // start state 0 = (0, 0): first created, never the accepting one (rows >= 1)
        }
    }
}

// Unicode typo configurations: the same criterion over SCALARS.  The reference's unicode typo algorithms (unicode_typos.rs:15-466) are the
// ASCII ones over occurrence masks of whole scalars - row i occurs at byte position p iff the bytes at p equal the scalar's or its case flip's
// (unicode.rs:74-117), occurrences never overlap (UTF-8: lead byte, then continuation bytes) - so what they decide on a haystack that fits one
// prefilter chunk is LCS(needle scalars, the haystack's occurrences) + k >= n, and beyond one chunk they deviate from it only when there is
// nothing to spare (tests/test_oracle_reference_properties.py::test_single_chunk_unicode_typo_prefilter_is_the_scalar_lcs_criterion,
// ::test_the_unicode_typo_prefilter_only_ever_deviates_on_marginal_inputs).  As a byte-level automaton: state = (reachable bit-vector V, node of
// the trie over the needle's distinct scalar byte strings = the bytes of the scalar being read); a byte that completes a string applies the LCS
// step with M = the rows spelt by it; a byte that does not continue the current string restarts from the root WITH that byte (a lead byte never
// occurs inside a scalar).  States are numbered by ascending LCS of V (an unfinished scalar counts for nothing), start state first.
// (Rounds 2-5 ran the byte-level LCS over the scalars' LAST bytes here - a looser superset that needed the lane-exact window kernel for every
// survivor; with the exact criterion single-chunk lists are decided in the stream: pipe_unicode_typo_fast_path.)
static bool build_scalar_lcs_dfa(fzb_matcher* m) {
    const NeedleDev& nd = m->nd;
    const int rows = m->rows, k = m->config.max_typos;
    const u64 mask = rows >= 64 ? ~(u64)0 : (((u64)1 << rows) - 1);
    // trie over the distinct byte strings uc[i] / uf[i]: node 0 = root; term[node] = rows spelt by the complete string ending there (0: internal)
    struct Node { int child[256]; u64 term; };
    std::vector<Node> trie(1);
    for (int b = 0; b < 256; b++) trie[0].child[b] = -1;
    trie[0].term = 0;
    auto insert = [&](const u8* str, int len, int row) {
        int cur = 0;
        for (int j = 0; j < len; j++) {
            if (trie[cur].child[str[j]] < 0) {
                Node nn;
                for (int b = 0; b < 256; b++) nn.child[b] = -1;
                nn.term = 0;
                trie.push_back(nn);
                trie[cur].child[str[j]] = (int)trie.size() - 1;
            }
            cur = trie[cur].child[str[j]];
        }
        trie[cur].term |= (u64)1 << row;
    };
    for (int r = 0; r < rows; r++) {
        if (nd.ulen[r] < 1 || nd.ulen[r] > 4) return false;
        insert(nd.uc[r], nd.ulen[r], r);
        insert(nd.uf[r], nd.ulen[r], r);
    }
    for (const Node& nn : trie)  // (UTF-8 is prefix-free: a complete scalar is never the prefix of another; anything else is not a needle this automaton models)
        if (nn.term)
            for (int b = 0; b < 256; b++)
                if (nn.child[b] >= 0) return false;
    struct St { u64 v; int node; };
    std::vector<St> states{St{mask, 0}};
    auto key = [](u64 v, int node) { return std::make_pair(v, node); };
    std::map<std::pair<u64, int>, int> id{{key(mask, 0), 0}};
    auto find = [&](u64 v, int node) {
        auto it = id.find(key(v, node));
        if (it != id.end()) return it->second;
        states.push_back(St{v, node});
        id.emplace(key(v, node), (int)states.size() - 1);
        return (int)states.size() - 1;
    };
    auto lcs_step = [&](u64 v, u64 mrows) { const u64 u = v & mrows; return ((v + u) | (v & ~mrows)) & mask; };
    auto from_root = [&](u64 v, int b) {  // state (v, root) reading byte b
        const int c = trie[0].child[b];
        if (c < 0) return find(v, 0);
        return trie[c].term ? find(lcs_step(v, trie[c].term), 0) : find(v, c);
    };
    std::vector<std::vector<int>> next;
    for (size_t q = 0; q < states.size(); q++) {
        const St cur = states[q];
        std::vector<int> row(256);
        for (int b = 0; b < 256; b++) {
            if (cur.node == 0) { row[b] = from_root(cur.v, b); continue; }
            const int c = trie[cur.node].child[b];
            if (c < 0) row[b] = from_root(cur.v, b);
            else row[b] = trie[c].term ? find(lcs_step(cur.v, trie[c].term), 0) : find(cur.v, c);
        }
        next.push_back(row);
        if (states.size() > 226) return false;  // 226 x 288 bytes (dfa_lds.h's row stride) + the tile counter fit the 64 KiB of dynamic LDS
    }
    const int ns = (int)states.size(), need = rows - k;
    std::vector<int> lcs(ns), order(ns), renum(ns);
    for (int q = 0; q < ns; q++) lcs[q] = __builtin_popcountll(~states[q].v & mask), order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lcs[a] < lcs[b]; });  // the start state (LCS 0, first created) stays first
    int acc = ns;
    for (int pos = 0; pos < ns; pos++) {
        renum[order[pos]] = pos;
        if (lcs[order[pos]] >= need && acc == ns) acc = pos;
    }
    m->lcs_dfa.assign((size_t)ns * 256, 0);
    for (int q = 0; q < ns; q++)
        for (int b = 0; b < 256; b++) m->lcs_dfa[(size_t)renum[q] * 256 + b] = (u8)renum[next[q][b]];
    m->lcs_states = ns;
    m->lcs_acc_lo = acc;
    m->lcs_scalar = true;
    return true;
}

static void build_lcs_dfa(fzb_matcher* m) {
    const LaunchCfg& lc = m->lc;
    const int k = m->config.max_typos;
    m->lcs_states = 0;
    m->lcs_scalar = false;
    if (m->unicode && !m->literal_mode && lc.filter_mode == 2 && m->rows >= 1 && m->rows <= 63 && !fzb_knobs().no_lcs_dfa && build_scalar_lcs_dfa(m)) return;
    // Typo configurations: the streaming filter's LCS criterion `LCS(needle, haystack) >= rows - k` (Hyyro's bit-vector recurrence
    // V' = (V + (V & M)) | (V & ~M), M = the rows byte b can match) as a table-driven automaton over its REACHABLE bit-vectors - 40-odd
    // states for a 6-row needle - so that the filter is the same v_perm + ds_read_u8 per byte as the 0-typo one (k1_dfa: 55 us on the
    // 10 M x 32 B list) instead of a table lookup + four vector operations (k1_filter: 71 us).  States are numbered by ascending LCS, so
    // "accepts" is one compare; the start state (LCS 0, only reachable as itself) is state 0.  More than 226 states (long needles with
    // many distinct letters): the bit-vector kernel stays.  FZB_NO_LCS_DFA=1 keeps it for comparison.
    m->lcs_states = 0;
    if (lc.filter_mode == 2 && m->rows >= 1 && m->rows <= 63 && !fzb_knobs().no_lcs_dfa) {
        const u64 mask = m->rows >= 64 ? ~(u64)0 : (((u64)1 << m->rows) - 1);
        std::vector<u64> masks;  // distinct M over the 256 byte values
        std::vector<int> mask_of(256);
        for (int b = 0; b < 256; b++) {
            const u64 mb = m->table[b] & mask;
            size_t q = 0;
            while (q < masks.size() && masks[q] != mb) q++;
            if (q == masks.size()) masks.push_back(mb);
            mask_of[b] = (int)q;
        }
        std::vector<u64> states{mask};  // V0: all ones in the low `rows` bits
        states.reserve(256);
        const size_t nm = masks.size();
        std::vector<int> next;  // [state][mask]
        next.reserve(256 * nm);
        bool ok = true;
        for (size_t q = 0; q < states.size() && ok; q++) {
            const u64 v = states[q];
            for (size_t t = 0; t < nm; t++) {
                const u64 u = v & masks[t];
                const u64 nv = ((v + u) | (v & ~masks[t])) & mask;
                size_t id = 0;  // (at most 226 states: a linear search beats a hash table's allocations)
                while (id < states.size() && states[id] != nv) id++;
                if (id == states.size()) {
                    states.push_back(nv);
                    if (states.size() > 226) ok = false;
                }
                next.push_back((int)id);
            }
        }
        if (ok) {
            const int ns = (int)states.size();
            const int need = m->rows - k;
            std::vector<int> lcs(ns), order(ns), renum(ns);
            for (int q = 0; q < ns; q++) lcs[q] = __builtin_popcountll(~states[q] & mask), order[q] = q;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lcs[a] < lcs[b]; });  // V0 (LCS 0, first created) stays first
            int acc = ns;
            for (int pos = 0; pos < ns; pos++) {
                renum[order[pos]] = pos;
                if (lcs[order[pos]] >= need && acc == ns) acc = pos;
            }
            m->lcs_dfa.assign((size_t)ns * 256, 0);
            for (int q = 0; q < ns; q++)
                for (int b = 0; b < 256; b++) m->lcs_dfa[(size_t)renum[q] * 256 + b] = (u8)renum[next[(size_t)q * nm + mask_of[b]]];
            m->lcs_states = ns;
            m->lcs_acc_lo = acc;
        }
    }
}

static void build_cdfa(fzb_matcher* m) {
    const LaunchCfg& lc = m->lc;
    // The class-composite form of the automaton the streaming filter runs (ragged lists: kernels_filter.hip, k1_cdfa_ragged): bytes with
    // identical columns are one class (K of them), G transitions are composed into one table indexed by
    // state * K^G + c0 + K c1 + ... (c0 = the class of the FIRST byte), G = 4 if states * K^4 <= 16 KB, else 2, else none.
    m->cdfa.clear();
    m->cdfa_src = m->cdfa_K = m->cdfa_G = 0;
    {
        const std::vector<u8>* fa = nullptr;
        int fstates = 0, src = 0;
        if (m->literal_mode == FZB_MATCH_SUBSTRING && !m->unicode) { fa = &m->dfa; fstates = m->rows + 1; src = 1; }
        else if (m->literal_mode) {}
        else if (m->uni_dfa_states) { fa = &m->uni_dfa; fstates = m->uni_dfa_states; src = 2; }
        else if (lc.filter_mode == 2 && m->lcs_states) { fa = &m->lcs_dfa; fstates = m->lcs_states; src = 3; }
        else if (lc.filter_mode == 1) { fa = &m->dfa; fstates = m->rows + 1; src = 1; }
        if (fa && fstates >= 1 && fstates <= 255) {
            std::vector<int> cls(256, -1);
            std::vector<int> rep;  // a representative byte per class
            for (int b = 0; b < 256; b++) {
                for (size_t q = 0; q < rep.size() && cls[b] < 0; q++) {
                    bool same = true;
                    for (int stt = 0; stt < fstates && same; stt++) same = (*fa)[(size_t)stt * 256 + b] == (*fa)[(size_t)stt * 256 + rep[q]];
                    if (same) cls[b] = (int)q;
                }
                if (cls[b] < 0) { cls[b] = (int)rep.size(); rep.push_back(b); }
            }
            const size_t K = rep.size();
            int G = 0;
            if ((size_t)fstates * K * K * K * K <= 16384) G = 4;
            else if ((size_t)fstates * K * K <= 16384) G = 2;
            if (G) {
                size_t KG = 1;
                for (int i = 0; i < G; i++) KG *= K;
                m->cdfa.assign(((256 + (size_t)fstates * KG) + 15) & ~(size_t)15, 0);
                for (int b = 0; b < 256; b++) m->cdfa[b] = (u8)cls[b];
                // two transitions composed first (c0, the least significant digit, is consumed first), then - G = 4 - two of those: no division per entry
                const size_t K2 = K * K;
                std::vector<u8> t2((size_t)fstates * K2);
                for (int stt = 0; stt < fstates; stt++)
                    for (size_t c1 = 0; c1 < K; c1++)
                        for (size_t c0 = 0; c0 < K; c0++)
                            t2[(size_t)stt * K2 + c0 + K * c1] = (*fa)[(size_t)(*fa)[(size_t)stt * 256 + rep[c0]] * 256 + rep[c1]];
                u8* comp = m->cdfa.data() + 256;
                if (G == 2) memcpy(comp, t2.data(), t2.size());
                else
                    for (int stt = 0; stt < fstates; stt++)
                        for (size_t hi = 0; hi < K2; hi++) {
                            u8* row = comp + (size_t)stt * KG + K2 * hi;
                            const u8* first = &t2[(size_t)stt * K2];
                            for (size_t lo = 0; lo < K2; lo++) row[lo] = t2[(size_t)first[lo] * K2 + hi];
                        }
                m->cdfa_src = src;
                m->cdfa_K = (int)K;
                m->cdfa_G = G;
            }
        }
    }
}

// which forms of the scorers this needle and scoring allow
static void set_scorer_forms(fzb_matcher* m, const uint8_t* needle_utf8, size_t needle_len) {
    const fzb_scoring& sc = m->config.scoring;
    LaunchCfg& lc = m->lc;
    lc.pad_ok = 1;
    for (size_t i = 0; i < needle_len; i++)
        if (needle_utf8[i] == 0) lc.pad_ok = 0;
    // biased gap propagation needs max cell value + lanes*gex (+ headroom) to stay below 2^16
    lc.bias_ok = max_matrix_score(sc, (size_t)m->rows) + (size_t)sc.mismatch_penalty + 130 * (size_t)sc.gap_extend_penalty + 64 <= 0xFFFF;
    // dp_cf.h / dp_cfm.h: every biased value stays below 0x7C00, so that the cell's three-way maximum can be v_pk_maximum3_f16 (exact on
    // non-negative finite binary16 patterns: dp_body.h, p_max3_s); scorings beyond that - needle rows worth thousands of points - take the first forms
    lc.cfm_ok = 2 * (u32)sc.gap_extend_penalty <= (u32)sc.mismatch_penalty &&
                max_matrix_score(sc, (size_t)m->rows) + (size_t)sc.mismatch_penalty + 200 * (size_t)sc.gap_extend_penalty + 64 < 0x7C00;  // lanes up to 3/2 chunks + rows of bias
    lc.cf_ok = lc.pad_ok && 2 * (u32)sc.gap_extend_penalty <= (u32)sc.mismatch_penalty &&
               max_matrix_score(sc, (size_t)m->rows) + (size_t)sc.mismatch_penalty + 130 * (size_t)sc.gap_extend_penalty + 64 < 0x7C00;
    lc.cfu_ok = lc.bias_ok && 2 * (u32)sc.gap_extend_penalty <= (u32)sc.mismatch_penalty && !fzb_knobs().no_dp_cfu;  // (knob: the unicode scorer's first form)
}

int fzb_matcher_create(const fzb_config* config, const uint8_t* needle_utf8, size_t needle_len, fzb_matcher** out) {
    if (!config || !out || (!needle_utf8 && needle_len)) return fail(FZB_ERR_INVALID, "null argument");
    if (config->casing < 0 || config->casing > 2 || config->unicode < 0 || config->unicode > 2 || config->sort < 0 || config->sort > 3 || config->max_typos < -1 ||
        config->max_typos > 0xFFFF)
        return fail(FZB_ERR_INVALID, "config enum/range out of bounds");
    std::vector<u32> cps;
    if (!decode_utf8(needle_utf8, needle_len, cps)) return fail(FZB_ERR_INVALID, "needle is not valid UTF-8");
    std::unique_ptr<fzb_matcher> mp(new fzb_matcher());
    fzb_matcher* m = mp.get();
    m->config = *config;
    m->needle.assign((const char*)needle_utf8, needle_len);
    m->empty = needle_len == 0;
    const fzb_scoring& sc = config->scoring;
    m->use_u8 = fits_in_u8(needle_len, sc);  // byte length (src/matcher/mod.rs:453)
    int pf = config->pf_lanes, sw = config->sw_lanes;
    if (pf == 0 && sw == 0) detect_host_lanes(m->use_u8, pf, sw);
    else if (sw == 0) sw = m->use_u8 ? pf : pf / 2;  // the score width of the ISA family whose prefilter has pf lanes (64: AVX-512, 32: AVX2, 16: SSE/scalar)
    if (!(pf == 16 || pf == 32 || pf == 64) || !(sw == 8 || sw == 16 || sw == 32 || sw == 64))
        return fail(FZB_ERR_INVALID, "pf_lanes must be 16/32/64 and sw_lanes 8/16/32/64 (or both 0 = auto)");
    m->lc.pf_lanes = pf;
    m->lc.sw_lanes = sw;
    if (m->empty) {  // CompiledPatterns::Empty (src/matcher/mod.rs:194-196)
        *out = mp.release();
        return FZB_OK;
    }
    // CaseMatching::respects_case_for (src/lib.rs:370-376), UnicodeMatching::respects_unicode_for (:394-400)
    bool any_upper = false, ascii = true;
    for (u32 cp : cps) { any_upper |= is_uppercase(cp); ascii &= cp < 0x80; }
    m->case_sensitive = config->casing == FZB_CASE_RESPECT || (config->casing == FZB_CASE_SMART && any_upper);
    m->unicode = config->unicode == FZB_UNICODE_ALWAYS || (config->unicode == FZB_UNICODE_SMART && !ascii);
    m->rows = (int)(m->unicode ? cps.size() : needle_len);
    if (config->matching < FZB_MATCH_FUZZY || config->matching > FZB_MATCH_SUBSTRING) return fail(FZB_ERR_INVALID, "bad matching mode");
    m->literal_mode = config->matching;
    // guard_against_score_overflow: fuzzy src/matcher/algo.rs:311-325 (rows); literal src/literal/algo.rs:33, 314-322 (needle bytes, its own per-char bonus)
    std::string perr = m->literal_mode ? overflow_guard(sc, needle_len, sadd16(std::max(sc.capitalization_bonus, sc.delimiter_bonus), sc.matching_case_bonus), 0)
                                       : overflow_guard(sc, (size_t)m->rows);
    if (!perr.empty()) return fail(FZB_ERR_PANIC, perr);
    // beyond NeedleDev's by-value arrays: the needle's arrays go to device memory (NeedleLongDev) and the query runs through the
    // kernels that take it from there (run_pipeline_long) - any length the reference's guard above accepted
    m->long_needle = needle_len > FZB_MAX_NEEDLE_BYTES || m->rows > FZB_MAX_ROWS;
    fill_needle_scalars(m, needle_len, cps.size());
    if (m->long_needle) {
        build_long_needle(m, needle_utf8, needle_len, cps);
        *out = mp.release();
        return FZB_OK;
    }
    NeedleDev& nd = m->nd;
    for (size_t i = 0; i < needle_len; i++) {  // case_needle (src/prefilter/mod.rs:49-65)
        nd.raw[i] = nd.c[i] = needle_utf8[i];
        nd.f[i] = flip_ascii_byte(m, needle_utf8[i]);
    }
    for (size_t i = 0; i < cps.size() && cps.size() <= FZB_MAX_ROWS; i++) {  // case_needle_unicode (src/prefilter/mod.rs:71-96)
        nd.ulen[i] = (u8)encode_utf8(cps[i], nd.uc[i]);
        encode_utf8(m->case_sensitive ? cps[i] : flip_same_width(cps[i]), nd.uf[i]);
    }
    // ---- filter-stage configuration --------------------------------------------------------------
    const int k = config->max_typos;
    LaunchCfg& lc = m->lc;
    if (k < 0 || k >= m->rows) {        // NO_PREFILTER, or `needle_len <= max_typos => (true, 0, len)`
        lc.filter_mode = 0; lc.filter_exact = 1; lc.window_mode = 2;
    } else if (!m->unicode && k == 0) { // exact: ordered subsequence; window = first/last occurrence
        lc.filter_mode = 1; lc.filter_exact = 1; lc.window_mode = 1;
    } else {                            // superset filter, lane-exact prefilter re-decides
        lc.filter_mode = k == 0 ? 1 : 2; lc.filter_exact = 0; lc.window_mode = 0;
    }
    build_filter_tables(m);
    build_unicode_dfa(m);
    build_lcs_dfa(m);
    build_cdfa(m);
    set_scorer_forms(m, needle_utf8, needle_len);
    *out = mp.release();
    return FZB_OK;
}

// `Matcher::set_pattern` / `Matcher::set_config` (src/matcher/mod.rs:154-176): the matcher is rebuilt for the new needle or config
// exactly as fzb_matcher_create would build it, but keeps its device workspace (sized by the corpus, not by the needle), so a
// re-query after every keystroke costs two small table uploads instead of a round of device allocations.
static int rebuild_matcher(fzb_matcher* m, const fzb_config* config, const uint8_t* needle_utf8, size_t needle_len) {
    fzb_matcher* fresh = nullptr;
    int rc = fzb_matcher_create(config, needle_utf8, needle_len, &fresh);
    if (rc) return rc;  // m is left as it was
    // device-side state moves over to the rebuilt matcher ...
    std::swap(fresh->ws, m->ws);
    fresh->ws.tables_stale = true;
    std::swap(fresh->out_dev, m->out_dev);
    std::swap(fresh->out_cap, m->out_cap);
    std::swap(fresh->count_dev, m->count_dev);
    std::swap(fresh->fetch, m->fetch);
    std::swap(fresh->long_scratch, m->long_scratch);
    std::swap(fresh->long_scratch_bytes, m->long_scratch_bytes);
    std::swap(fresh->aux_stream, m->aux_stream);
    std::swap(fresh->ev_fork, m->ev_fork);
    std::swap(fresh->ev_join, m->ev_join);
    fresh->device = m->device;
    fresh->lc.num_cus = m->lc.num_cus;
    fresh->profiling = m->profiling;
    fresh->prof_calls = m->prof_calls;
    for (int i = 0; i < fzb_matcher::PROF_SLOTS; i++) {
        for (int k = 0; k < 5; k++) std::swap(fresh->evring[i][k], m->evring[i][k]);
        fresh->ev_filter[i] = m->ev_filter[i];
    }
    std::swap(fresh->shard_stream, m->shard_stream);  // the multi-device form's state: a clone's stream / event / count, the parent's workers
    std::swap(fresh->shard_event, m->shard_event);
    std::swap(fresh->shard_count_host, m->shard_count_host);
    std::swap(fresh->shard_device, m->shard_device);
    std::swap(fresh->shard_workers, m->shard_workers);
    std::swap(fresh->shard_clones, m->shard_clones);
    // ... and then the handle the caller holds takes the rebuilt matcher's place
    std::swap(*fresh, *m);
    fzb_matcher_free(fresh);
    // the per-shard clones of the multi-device form follow (same needle, same config with the resolved lane pair); their device
    // workspaces stay where they are
    if (!m->shard_clones.empty()) {
        fzb_config ccfg = m->config;
        ccfg.pf_lanes = (uint16_t)m->lc.pf_lanes;
        ccfg.sw_lanes = (uint16_t)m->lc.sw_lanes;
        bool ok = true;
        for (fzb_matcher* cm : m->shard_clones) ok = ok && rebuild_matcher(cm, &ccfg, needle_utf8, needle_len) == FZB_OK;
        if (!ok) {  // cannot happen for a needle / config the main matcher accepted; drop the clones rather than keep stale ones
            std::vector<fzb_matcher*> drop;
            drop.swap(m->shard_clones);
            for (fzb_matcher* cm : drop) fzb_matcher_free(cm);
            fzb_clear_error();
        }
    }
    return FZB_OK;
}

int fzb_matcher_set_pattern(fzb_matcher* m, const uint8_t* needle_utf8, size_t needle_len) {
    if (!m || (!needle_utf8 && needle_len)) return fail(FZB_ERR_INVALID, "null argument");
    if (m->needle.size() == needle_len && (needle_len == 0 || memcmp(m->needle.data(), needle_utf8, needle_len) == 0)) return FZB_OK;  // "skipped if the pattern is the same"
    const fzb_config cfg = m->config;
    return rebuild_matcher(m, &cfg, needle_utf8, needle_len);
}

int fzb_matcher_set_config(fzb_matcher* m, const fzb_config* config) {
    if (!m || !config) return fail(FZB_ERR_INVALID, "null argument");
    {  // field by field: fzb_config has padding bytes a C caller need not have cleared
        const fzb_config &a = m->config, &b = *config;
        const bool same = a.max_typos == b.max_typos && a.casing == b.casing && a.unicode == b.unicode && a.sort == b.sort && a.pf_lanes == b.pf_lanes && a.sw_lanes == b.sw_lanes &&
                          a.matching == b.matching && a.scoring.match_score == b.scoring.match_score && a.scoring.mismatch_penalty == b.scoring.mismatch_penalty &&
                          a.scoring.gap_open_penalty == b.scoring.gap_open_penalty && a.scoring.gap_extend_penalty == b.scoring.gap_extend_penalty &&
                          a.scoring.prefix_bonus == b.scoring.prefix_bonus && a.scoring.capitalization_bonus == b.scoring.capitalization_bonus &&
                          a.scoring.matching_case_bonus == b.scoring.matching_case_bonus && a.scoring.exact_match_bonus == b.scoring.exact_match_bonus &&
                          a.scoring.delimiter_bonus == b.scoring.delimiter_bonus;
        if (same) return FZB_OK;
    }
    const std::string needle = m->needle;
    return rebuild_matcher(m, config, (const uint8_t*)needle.data(), needle.size());
}

int fzb_matcher_clone(const fzb_matcher* src, fzb_matcher** out) {
    if (!src || !out) return fail(FZB_ERR_INVALID, "null argument");
    fzb_config cfg = src->config;
    cfg.pf_lanes = (uint16_t)src->lc.pf_lanes;
    cfg.sw_lanes = (uint16_t)src->lc.sw_lanes;
    return fzb_matcher_create(&cfg, (const uint8_t*)src->needle.data(), src->needle.size(), out);
}

void fzb_matcher_free(fzb_matcher* m) {
    if (!m) return;
    free_workspace(m->ws);
    if (m->out_dev) (void)hipFree(m->out_dev);
    if (m->count_dev) (void)hipFree(m->count_dev);
    for (void* p : {(void*)m->trace_sel, (void*)m->trace_pos, (void*)m->trace_npos})
        if (p) (void)hipFree(p);
    for (auto& tr : m->evring)
        for (auto& e : tr)
            if (e) (void)hipEventDestroy(e);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->aux_stream) (void)hipStreamDestroy(m->aux_stream);
    if (m->long_blob_dev) (void)hipFree(m->long_blob_dev);
    if (m->long_scratch) (void)hipFree(m->long_scratch);
    if (m->fetch.count_host) (void)hipHostFree(m->fetch.count_host);
    if (m->shard_workers) fzb_shard_workers_free(m->shard_workers);  // joins the worker threads before their clones go
    m->shard_workers = nullptr;
    if (m->shard_stream) (void)hipStreamDestroy(m->shard_stream);
    if (m->shard_event) (void)hipEventDestroy(m->shard_event);
    if (m->shard_count_host) (void)hipHostFree(m->shard_count_host);
    if (!m->shard_clones.empty()) {
        // a clone's device state lives on its shard's device
        int cur = 0;
        const bool have_cur = hipGetDevice(&cur) == hipSuccess;
        for (fzb_matcher* cm : m->shard_clones) {
            if (cm->shard_device >= 0) (void)hipSetDevice(cm->shard_device);
            fzb_matcher_free(cm);
        }
        m->shard_clones.clear();
        if (have_cur) (void)hipSetDevice(cur);
    }
    delete m;
}

int fzb_matcher_info(const fzb_matcher* m, int32_t out[6]) {
    if (!m || !out) return fail(FZB_ERR_INVALID, "null argument");
    out[0] = m->lc.pf_lanes; out[1] = m->lc.sw_lanes; out[2] = m->use_u8; out[3] = m->case_sensitive; out[4] = m->unicode; out[5] = m->rows;
    return FZB_OK;
}

// ---- corpus (fzb_corpus_upload: host_upload.hip) -------------------------------------------------------
}  // extern "C"
// A promise about BORROWED memory (uniform length / longest haystack) selects kernels that compute spans instead of reading the end
// offsets, or that skip launches: a wrong one would silently mis-span every haystack.  One pass over the end offsets checks it when it is
// made (a set-up call: it synchronises): offsets non-decreasing in the padded-16 layout and inside the buffer, every length == uniform_len
// (when given), every length <= max_len (when given).  out[0] = number of violations, out[1] = the first offending index.
template <typename ET>
__global__ __launch_bounds__(256) void k_verify_promise(const ET* __restrict__ ends, u64 n, u64 total_bytes, u32 uniform_len, u32 max_len, unsigned long long* __restrict__ out) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u32 bad = 0;
    u64 first_bad = ~0ull;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 e = (u64)ends[i];
        const u64 s = i ? (((u64)ends[i - 1] + 15ull) & ~15ull) : 0ull;
        bool ok = e >= s && e <= total_bytes;
        const u64 len = e - s;
        if (uniform_len) ok = ok && len == (u64)uniform_len;
        if (max_len) ok = ok && len <= (u64)max_len;
        if (!ok) {
            bad++;
            first_bad = first_bad < i ? first_bad : i;
        }
    }
    if (bad) {
        atomicAdd(&out[0], (unsigned long long)bad);
        atomicMin(&out[1], (unsigned long long)first_bad);
    }
}
static int verify_promise(const fzb_corpus* c, u32 uniform_len, u32 max_len, const char* what) {
    if (!fzb_knobs().verify_promises || !c->dev.n || (!uniform_len && !max_len)) return FZB_OK;
    unsigned long long* d = nullptr;
    // the borrowed buffers may live on another device than the caller's current one: the check runs where the end offsets are
    int prev_dev = 0, own_dev = 0;
    HIPCHK(hipGetDevice(&prev_dev));
    own_dev = prev_dev;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, c->dev.ends) == hipSuccess && attr.type == hipMemoryTypeDevice) own_dev = attr.device;
    else (void)hipGetLastError();
    struct DeviceGuard {
        int prev, cur;
        ~DeviceGuard() { if (cur != prev) (void)hipSetDevice(prev); }
    } guard{prev_dev, own_dev};
    if (own_dev != prev_dev) HIPCHK(hipSetDevice(own_dev));
    HIPCHK(hipDeviceSynchronize());  // the caller may have filled the buffers on any stream of that device
    HIPCHK(hipMalloc((void**)&d, 16));
    const unsigned long long init[2] = {0ull, ~0ull};
    hipError_t e = hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int grid = (int)std::min<u64>((c->dev.n + 255) / 256, 4096);
        if (c->dev.ends_u64) hipLaunchKernelGGL((k_verify_promise<u64>), dim3(grid), dim3(256), 0, 0, (const u64*)c->dev.ends, c->dev.n, c->dev.total_bytes, uniform_len, max_len, d);
        else hipLaunchKernelGGL((k_verify_promise<u32>), dim3(grid), dim3(256), 0, 0, (const u32*)c->dev.ends, c->dev.n, c->dev.total_bytes, uniform_len, max_len, d);
        e = hipGetLastError();
    }
    unsigned long long got[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(got, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(FZB_ERR_HIP, std::string("verifying the corpus promise: ") + hipGetErrorString(e));
    if (got[0])
        return fail(FZB_ERR_INVALID, std::string(what) + ": the end offsets contradict it for " + std::to_string(got[0]) + " haystack(s), first at index " + std::to_string(got[1]) +
                                         " (checked on the device; FZB_VERIFY_PROMISES=0 skips the check)");
    return FZB_OK;
}
extern "C" {
int fzb_corpus_from_device(const void* dev_bytes, const void* dev_ends, int ends_are_u64, size_t n, uint64_t total_bytes, fzb_corpus** out) {
    if (!out || (n && (!dev_bytes || !dev_ends))) return fail(FZB_ERR_INVALID, "null argument");
    if (((uintptr_t)dev_bytes & 15) != 0) return fail(FZB_ERR_INVALID, "dev_bytes must be 16-byte aligned");
    if (n > 0xFFFFFFFFull) return fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string(n) + " > 4294967295 (index offset: 0)");
    auto c = new fzb_corpus();
    c->dev.bytes = (const u8*)dev_bytes;
    c->dev.ends = dev_ends;
    c->dev.ends_u64 = ends_are_u64 != 0;
    c->dev.n = n;
    c->dev.total_bytes = total_bytes;
    *out = c;
    return FZB_OK;
}

void fzb_corpus_free(fzb_corpus* c) {
    if (!c) return;
    if (c->own_bytes) (void)hipFree(c->own_bytes);
    if (c->own_ends) (void)hipFree(c->own_ends);
    for (void* q : c->own_view)
        if (q) (void)hipFree(q);
    delete c;
}
size_t fzb_corpus_len(const fzb_corpus* c) { return c ? (size_t)c->dev.n : 0; }
int fzb_corpus_set_uniform_len(fzb_corpus* c, uint32_t len) {
    if (!c) return fail(FZB_ERR_INVALID, "null argument");
    // an uploaded corpus knows its lengths: only the detected value (a no-op) is accepted, anything else would make the kernels that
    // compute spans and the ones that read the end offsets disagree within one query
    if (c->own_bytes) {
        if (len == c->dev.uniform_len) return FZB_OK;
        return fail(FZB_ERR_INVALID, "the corpus was uploaded by fzb_corpus_upload, which detected its lengths itself (uniform length " + std::to_string(c->dev.uniform_len) +
                                         ", 0 = not uniform); the promise can only be made for borrowed device memory");
    }
    if (len && (u64)((len + 15u) & ~15u) * (c->dev.n ? c->dev.n - 1 : 0) + len > c->dev.total_bytes) return fail(FZB_ERR_INVALID, "uniform length does not fit the corpus buffer");
    if (len) {  // the promise is checked against the end offsets before any kernel relies on it
        const int vrc = verify_promise(c, len, 0, ("a uniform length of " + std::to_string(len) + " bytes was promised").c_str());
        if (vrc) return vrc;
    }
    if (!len && c->dev.uniform_len && c->dev.max_len == c->dev.uniform_len) c->dev.max_len = 0;  // clearing the promise also clears the bound it implied
    c->dev.uniform_len = len;
    if (len) c->dev.max_len = len;  // (overwrites an earlier fzb_corpus_set_max_len)
    return FZB_OK;
}

int fzb_corpus_set_max_len(fzb_corpus* c, uint32_t max_len) {
    if (!c) return fail(FZB_ERR_INVALID, "null argument");
    // an uploaded corpus knows its longest haystack: a hint can only repeat or loosen it (ignored), and one BELOW it is refused - kernels are
    // selected by this bound and would skip bytes
    if (c->own_bytes) {
        if (max_len == 0 || max_len >= c->dev.max_len) return FZB_OK;
        return fail(FZB_ERR_INVALID, "the corpus was uploaded by fzb_corpus_upload, which measured its longest haystack (" + std::to_string(c->dev.max_len) + " bytes); " +
                                         std::to_string(max_len) + " is not an upper bound");
    }
    if (c->dev.uniform_len) {  // a (verified) uniform length implies its own, tighter bound: a looser one is accepted and changes nothing
        if (max_len == 0 || max_len >= c->dev.uniform_len) return FZB_OK;
        return fail(FZB_ERR_INVALID, "the corpus promises a uniform length of " + std::to_string(c->dev.uniform_len) + " bytes; " + std::to_string(max_len) + " is not an upper bound");
    }
    if (max_len && max_len != c->dev.uniform_len) {  // (a uniform length already verified implies its own bound)
        const int vrc = verify_promise(c, 0, max_len, ("a longest haystack of " + std::to_string(max_len) + " bytes was promised").c_str());
        if (vrc) return vrc;
    }
    c->dev.max_len = max_len;
    return FZB_OK;
}

}  // extern "C"
// A matcher's device state (workspace, staging, sort buffers) lives on the device that was current at its first query; later queries
// must find the same device current.
int fzb_bind_device(fzb_matcher* m) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (m->device >= 0) {
        if (dev != m->device)
            return fail(FZB_ERR_INVALID, "matcher is bound to device " + std::to_string(m->device) + " (its workspace lives there) but device " + std::to_string(dev) + " is current");
        return FZB_OK;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    m->device = dev;
    m->lc.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return FZB_OK;
}
extern "C" {

// ---- pipeline -------------------------------------------------------------------------------------------
// ASCII typo configuration whose scorer can be the short-haystack kernel: the filter's LCS criterion decides, only its marginal
// survivors (LCS == rows - k) are re-decided at the exact lane width, and the scorer computes the lane-free window itself
// (whether the corpus allows it - max_len - is known per call: run_pipeline)
static bool typo_fast_path_configured(const fzb_matcher* m) {
    // FZB_TYPO_EXACT_WINDOW=1: every typo query takes the reference's chunked multi-path scan at the exact lane width for every survivor
    // (superset filter -> k2a_window -> k_compact2), i.e. nothing rests on the two properties of DESIGN.md section 3e
    if (fzb_knobs().typo_exact_window) return false;
    return !m->literal_mode && !m->empty && m->lc.filter_mode == 2 && !m->nd.unicode && m->lc.cf_ok && (m->lc.sw_lanes == 64 || m->lc.sw_lanes == 32);
}

static int ensure_workspace(fzb_matcher* m, size_t count, hipStream_t st = nullptr, bool have_stream = false) {
    Workspace& w = m->ws;
    const bool need_l2 = !m->lc.filter_exact;
    const bool need_marg = typo_fast_path_configured(m);
    const bool need_cls = !m->literal_mode && !m->empty && !m->nd.unicode && m->lc.cf_ok;  // classified scoring (fzb_launch_dp_classes)
    // (cap_items >= count + FZB_UNICODE_FWD_CAP: run_pipeline anchors the queue's back at count + FZB_UNICODE_FWD_CAP entries - a workspace
    // allocated for a smaller range holds count0 + count0/8 + 4096 entries and must not be reused for a range within 4096 of that)
    if (w.cap_items >= count + FZB_UNICODE_FWD_CAP && (!need_l2 || w.cap_level2 >= count) && (!need_marg || w.cap_marg >= count) && (!need_cls || w.cap_cls >= count) && w.counters) {
        if (w.tables_stale) return upload_tables(m, st, have_stream);  // fzb_matcher_set_pattern / set_config kept the device buffers: only the tables change
        return FZB_OK;
    }
    free_workspace(w);
    const size_t cap = count + count / 8 + 4096;
    const size_t ntiles = (cap + FZB_TILE - 1) / FZB_TILE + 2;
    HIPCHK(dev_alloc((void**)&w.bitmap, (cap / 64 + 17) * 8));
    HIPCHK(dev_alloc((void**)&w.tile_counts, ntiles * 4));
    HIPCHK(dev_alloc((void**)&w.surv_idx, cap * 4));
    HIPCHK(dev_alloc((void**)&w.overflow, cap * 16));
    HIPCHK(dev_alloc((void**)&w.counters, 64));
    // (room for any needle's automata: short needles 64 states, a long needle's subsequence DFA up to 201, unicode <= 255 + 1, LCS <= 226)
    HIPCHK(dev_alloc((void**)&w.tables_blob, TB_TOTAL));
    HIPCHK(hipHostMalloc((void**)&w.tables_host, TB_TOTAL, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&w.tables_ev, hipEventDisableTiming));
    w.table = (u64*)(w.tables_blob + TB_TABLE);
    w.dfa = w.tables_blob + TB_DFA;
    w.uni_dfa = w.tables_blob + TB_UNI;
    w.lcs_dfa = w.tables_blob + TB_LCS;
    w.cdfa = w.tables_blob + TB_CDFA;
    {
        const int rc_t = upload_tables(m, st, have_stream);
        if (rc_t) return rc_t;
    }
    w.cap_items = cap;
    if (need_l2) {
        HIPCHK(dev_alloc((void**)&w.win, cap * 8));
        HIPCHK(dev_alloc((void**)&w.bitmap2, (cap / 64 + 17) * 8));
        HIPCHK(dev_alloc((void**)&w.tile_counts2, ntiles * 4));
        HIPCHK(dev_alloc((void**)&w.items2, cap * 4));
        HIPCHK(dev_alloc((void**)&w.win2, cap * 8));
        w.cap_level2 = cap;
    }
    if (need_marg) {
        HIPCHK(dev_alloc((void**)&w.bitmap_m, (cap / 64 + 17) * 8));
        HIPCHK(dev_alloc((void**)&w.tile_counts_m, ntiles * 4));
        HIPCHK(dev_alloc((void**)&w.marg_list, cap * 4));
        HIPCHK(dev_alloc((void**)&w.reject_bits, (cap / 64 + 17) * 8));
        HIPCHK(dev_alloc((void**)&w.tile_rejects, ntiles * 4));
        HIPCHK(dev_alloc((void**)&w.rej_prefix, ntiles * 4));
        w.cap_marg = cap;
    }
    if (need_cls) {
        HIPCHK(dev_alloc((void**)&w.cls_win, cap * 16));  // 16-byte records (window, flag, source address)
        HIPCHK(dev_alloc((void**)&w.cls_lists, cap * 28));  // 3 single-chunk classes + 4 multi-chunk tail classes
        w.cap_cls = cap;
    }
    return FZB_OK;
}

// ---- buffers beyond the per-range workspace, each grown by ONE helper so that fzb_matcher_reserve can size all of them ahead of
// the first query (a re-query after every keystroke must not meet a hipFree / hipMalloc) --------------------------------------
static int ensure_aux_stream(fzb_matcher* m) {
    if (m->aux_stream && m->ev_fork && m->ev_join) return FZB_OK;
    // all three or none: created into locals and published together, so a failure half way leaves the matcher on its single-stream path
    hipStream_t s = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ef, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (ej) (void)hipEventDestroy(ej);
        if (ef) (void)hipEventDestroy(ef);
        if (s) (void)hipStreamDestroy(s);
        return fail(FZB_ERR_HIP, std::string("second stream of the query: ") + hipGetErrorString(e));
    }
    m->aux_stream = s;
    m->ev_fork = ef;
    m->ev_join = ej;
    return FZB_OK;
}
static int ensure_dp_scratch(fzb_matcher* m, int mgrid) {  // parked rows of the multi-chunk scorer
    Workspace& w = m->ws;
    const size_t words = (size_t)(m->nd.rows + 1) * (size_t)(m->lc.sw_lanes / 2) * (size_t)mgrid * 128;
    if (w.dp_scratch_words >= words) return FZB_OK;
    if (w.dp_scratch) HIPCHK(hipFree(w.dp_scratch));
    w.dp_scratch = nullptr;
    w.dp_scratch_words = 0;
    HIPCHK(dev_alloc((void**)&w.dp_scratch, words * 4));
    w.dp_scratch_words = words;
    return FZB_OK;
}
static int ensure_sort_buffers(fzb_matcher* m, size_t cap) {  // ping-pong buffer + tile histograms of the device radix sort
    Workspace& w = m->ws;
    if (w.sort_cap >= cap && w.sort_tmp) return FZB_OK;
    if (w.sort_tmp) HIPCHK(hipFree(w.sort_tmp));
    if (w.sort_hist) HIPCHK(hipFree(w.sort_hist));
    w.sort_tmp = nullptr; w.sort_hist = nullptr; w.sort_cap = 0;
    HIPCHK(dev_alloc((void**)&w.sort_tmp, (cap + 16) * sizeof(fzb_match_rec)));
    const size_t hist_words = (size_t)2 * 256 * (cap / 2048 + 2);  // tile histograms + their scan; behind them two sets of digit totals + the phase word
    HIPCHK(dev_alloc((void**)&w.sort_hist, (hist_words + 1024) * 4));
    HIPCHK(hipMemset(w.sort_hist + hist_words, 0, 1024 * 4));
    w.sort_cap = cap;
    return FZB_OK;
}
}  // extern "C"
int fzb_ensure_out_staging(fzb_matcher* m, size_t count) {  // device-side result of the synchronous entry points
    if (m->out_cap >= count && m->count_dev && m->out_dev) return FZB_OK;
    if (m->out_dev) (void)hipFree(m->out_dev);
    m->out_dev = nullptr;
    m->out_cap = 0;
    HIPCHK(dev_alloc((void**)&m->out_dev, (count + 16) * sizeof(fzb_match_rec)));
    m->out_cap = count;
    if (!m->count_dev) HIPCHK(dev_alloc((void**)&m->count_dev, 64));
    return FZB_OK;
}
extern "C" {

// matched-indices form of a query: where the positions go
struct TraceOut {
    u32* pos;
    u32* npos;
    u32 stride;
};
static int ensure_long_needle(fzb_matcher* m, size_t scratch_bytes) {  // the needle's arrays + the scratch its kernels need, on the device
    if (!m->long_blob_dev) {
        HIPCHK(dev_alloc(&m->long_blob_dev, m->long_blob_host.size()));
        HIPCHK(hipMemcpy(m->long_blob_dev, m->long_blob_host.data(), m->long_blob_host.size(), hipMemcpyHostToDevice));
        const u8* b = (const u8*)m->long_blob_dev;
        m->ndl.raw = b;
        m->ndl.c = b + m->long_off_c;
        m->ndl.f = b + m->long_off_f;
        m->ndl.uc = (const u8(*)[4])(b + m->long_off_uc);
        m->ndl.uf = (const u8(*)[4])(b + m->long_off_uf);
        m->ndl.ulen = b + m->long_off_ulen;
    }
    if (m->long_scratch_bytes < scratch_bytes) {
        if (m->long_scratch) HIPCHK(hipFree(m->long_scratch));
        m->long_scratch = nullptr;
        m->long_scratch_bytes = 0;
        HIPCHK(dev_alloc(&m->long_scratch, scratch_bytes));
        m->long_scratch_bytes = scratch_bytes;
    }
    return FZB_OK;
}

// The pipeline of a LONG needle (> 64 bytes or > 63 rows; the reference takes them up to Scoring::max_needle_len(), src/lib.rs:480-503).
// Such a needle only matches haystacks about as long as itself, so nothing here is tuned for throughput; it is the same arithmetic through
// the kernels that take the needle from device memory:  [length test + the reference's prefilter at the exact lane width ->
// order-preserving compaction] -> the wave-per-haystack scorer (every window width; match_greedy beyond 1024 bytes; traced form for the
// matched-indices entry points) - or the two literal kernels.
// (profiling: [0] start, [2]..[3] around the first stage - the lane-exact prefilter over every haystack, when there is one -, [4] before the
// scorer, [1] end: the same five events fzb_last_stage_timings reads for the short-needle pipeline)
static int run_pipeline_long(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, const u32* items_in, const u32* n_items_in,
                             fzb_match* dev_out, u32 cap32, uint32_t* dev_count, hipStream_t st, const TraceOut* trace) {
    hipEvent_t* pev = nullptr;
    if (m->profiling) {
        const int slot = (int)(m->prof_calls % fzb_matcher::PROF_SLOTS);
        pev = m->evring[slot];
        m->ev_filter[slot] = (!m->literal_mode && m->lc.window_mode == 0) ? 1 : 0;
        m->prof_calls++;
        for (int i = 0; i < 5; i++)
            if (!pev[i]) HIPCHK(hipEventCreate(&pev[i]));
        HIPCHK(hipEventRecord(pev[0], st));
    }
    Workspace& w = m->ws;
    const CorpusDev& cd = c->dev;
    const int cus = m->lc.num_cus;
    const u32 cnt = (u32)count;
    u32* cnt_c = w.counters;
    const size_t budget = (size_t)256 << 20;  // global scratch per matcher: the grids below shrink to stay inside it
    if (m->literal_mode) {
        int rc = ensure_long_needle(m, 0);
        if (rc) return rc;
        fzb_launch_literal_filter_long(cd, first, cnt, items_in, n_items_in, m->ndl, m->literal_mode, w.bitmap, w.tile_counts, cus * 8, st);
        fzb_launch_compact1(w.bitmap, w.tile_counts, cnt, items_in ? n_items_in : nullptr, items_in, w.surv_idx, &cnt_c[0], cus * 2, st);
        if (pev) HIPCHK(hipEventRecord(pev[4], st));
        fzb_launch_literal_score_long(cd, first, index_offset, w.surv_idx, &cnt_c[0], m->ndl, m->literal_mode, (fzb_match_rec*)dev_out, cap32, dev_count, trace ? trace->pos : nullptr,
                                      trace ? trace->npos : nullptr, trace ? trace->stride : 0u, cus * 4, st);
        if (pev) HIPCHK(hipEventRecord(pev[1], st));
        HIPCHK(hipGetLastError());
        return FZB_OK;
    }
    const bool prefilter = m->lc.window_mode == 0;
    // grids: bounded by the scratch budget (a 10 922-row needle has 0.7 MB of previous-chunk vectors per wave, 50 MB of traced cells)
    int wgrid = std::max(1, cus * 2);
    if (prefilter && m->ndl.max_typos >= 3) {
        const size_t per_block = fzb_window_long_scratch_bytes(m->ndl, 1);
        wgrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)wgrid, budget / std::max<size_t>(per_block, 1)));
    }
    int ggrid = std::max(1, cus * 2);
    {
        const size_t per_block = fzb_generic_long_adj_bytes(m->ndl, m->lc.sw_lanes, 1) + (trace ? fzb_trace_scratch_words_long(m->ndl, 1) * 4 : 0);
        ggrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)ggrid, budget / std::max<size_t>(per_block, 1)));
        if (trace) ggrid = std::min(ggrid, (int)std::max<size_t>(1, (count + 3) / 4));
    }
    // ASCII windows of up to 1024 bytes: one THREAD per window (k2d_dp_long) when the parked-row slab lets enough of them run (a needle of
    // thousands of rows leaves room for a few thousand threads only: the wave-per-haystack kernel keeps those); matched indices and unicode
    // stay with the wave-per-haystack kernel, which also takes what k2d_dp_long queues (windows beyond 1024 bytes: match_greedy)
    const size_t dpl_words = fzb_dp_long_scratch_words_per_thread(m->ndl, m->lc.sw_lanes);
    const size_t dfit = std::min<size_t>((size_t)cus * 8, budget / std::max<size_t>(dpl_words * 4 * 128, 1));  // 128-thread workgroups the slab has room for
    const int dgrid = (int)std::max<size_t>(1, std::min<size_t>(dfit, (count + 127) / 128));
    const bool thread_per_window = !trace && !m->ndl.unicode && dfit >= (size_t)std::max(1, cus / 2);
    // four lanes per window (dp_quad.h) when dp_cfm.h's preconditions hold: a seventh of the slab per window, four times the wavefronts
    const size_t qwords = fzb_dp_long_quad_words_per_block(m->ndl, m->lc.sw_lanes);
    const bool quad = thread_per_window && m->lc.cfm_ok && !fzb_knobs().no_dp_cfm && qwords != 0;
    const int qgrid = quad ? (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)cus * 8, budget / (qwords * 4)), (count + 63) / 64)) : 0;
    const size_t dpl_bytes = quad ? qwords * 4 * (size_t)qgrid : thread_per_window ? dpl_words * 4 * 128 * (size_t)dgrid : 0;
    const bool greedy_possible = !(cd.max_len != 0 && cd.max_len <= FZB_MAX_HAYSTACK_LEN);
    if (thread_per_window && !greedy_possible) ggrid = 1;  // (no launch of the wave-per-haystack kernel: its slab is not needed)
    const size_t win_bytes = prefilter ? fzb_window_long_scratch_bytes(m->ndl, wgrid) : 0;
    const size_t adj_bytes = std::max(fzb_generic_long_adj_bytes(m->ndl, m->lc.sw_lanes, ggrid), dpl_bytes);
    const size_t cell_bytes = trace ? fzb_trace_scratch_words_long(m->ndl, ggrid) * 4 : 0;
    // (the prefilter's path state and the scorer's vectors are never live at the same time: they share the front of the scratch)
    const size_t front = (std::max(win_bytes, adj_bytes) + 255) & ~(size_t)255;
    int rc = ensure_long_needle(m, front + cell_bytes + 256);
    if (rc) return rc;
    const u32* items = items_in;
    const u32* win = nullptr;
    const u32* n_items_ptr = &cnt_c[0];
    int wmode = 2;
    if (items_in) HIPCHK(hipMemcpyAsync(&cnt_c[0], n_items_in, 4, hipMemcpyDeviceToDevice, st));
    else HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&cnt_c[0], (int)cnt, 1, st));
    // the streaming filter as first stage (0 typos, ASCII, up to 200 rows: fzb_matcher_create built the automaton) over a contiguous range;
    // the scorer then finds the lane-free window itself (wmode 1), as for short needles
    const bool use_dfa = m->long_dfa && thread_per_window && !items_in;
    if (use_dfa) {
        if (pev) HIPCHK(hipEventRecord(pev[2], st));
        fzb_launch_filter(cd, first, cnt, w.table, w.dfa, m->lc.dead_byte, m->ndl.rows, 1, m->ndl.rows, (u32)m->ndl.min_haystack_len, w.bitmap, w.tile_counts, w.counters, cus * 8, st, nullptr,
                          nullptr, nullptr, nullptr, m->lc.pad_ok, -1, nullptr, 0u, 0, 0);
        if (pev) HIPCHK(hipEventRecord(pev[3], st));
        fzb_launch_compact1(w.bitmap, w.tile_counts, cnt, nullptr, nullptr, w.surv_idx, &cnt_c[0], cus * 4, st);
        items = w.surv_idx;
        n_items_ptr = &cnt_c[0];
        wmode = 1;
    } else if (prefilter) {
        if (pev) HIPCHK(hipEventRecord(pev[2], st));
        fzb_launch_window_long(cd, first, items_in, &cnt_c[0], m->ndl, m->lc.pf_lanes, w.win, w.bitmap2, w.tile_counts2, m->long_scratch, wgrid, st);
        if (pev) HIPCHK(hipEventRecord(pev[3], st));
        fzb_launch_compact2(w.bitmap2, w.tile_counts2, &cnt_c[0], items_in, w.win, w.items2, w.win2, &cnt_c[1], cus * 2, st);
        items = w.items2;
        win = w.win2;
        n_items_ptr = &cnt_c[1];
        wmode = 0;
    } else {
        HIPCHK(hipMemcpyAsync(&cnt_c[1], &cnt_c[0], 4, hipMemcpyDeviceToDevice, st));
    }
    if (pev) HIPCHK(hipEventRecord(pev[4], st));
    if (thread_per_window) {
        if (quad)
            fzb_launch_dp_long_quad(cd, first, index_offset, items, win, wmode, n_items_ptr, m->ndl, m->lc.sw_lanes, m->long_upper ? 1 : 0, (fzb_match_rec*)dev_out, cap32, dev_count,
                                    (u32*)m->long_scratch, w.overflow, cnt_c, qgrid, st);
        else
        fzb_launch_dp_long(cd, first, index_offset, items, win, wmode, n_items_ptr, m->ndl, m->lc.sw_lanes, m->lc.bias_ok, (fzb_match_rec*)dev_out, cap32, dev_count, (u32*)m->long_scratch,
                           w.overflow, cnt_c, dgrid, st);
        // (the queue's entries are read after the slab's last use: the two kernels are one behind the other on the stream and share the scratch)
        if (greedy_possible)
            fzb_launch_generic_long(cd, first, index_offset, items, win, wmode, &cnt_c[3], m->ndl, m->lc.sw_lanes, (fzb_match_rec*)dev_out, cap32, nullptr, cnt_c, (u16*)m->long_scratch,
                                    nullptr, nullptr, nullptr, 0u, std::max(1, std::min(ggrid, cus / 4 + 1)), st, w.overflow);
    } else
    fzb_launch_generic_long(cd, first, index_offset, items, win, wmode, n_items_ptr, m->ndl, m->lc.sw_lanes, (fzb_match_rec*)dev_out, cap32, dev_count, cnt_c, (u16*)m->long_scratch,
                            trace ? (const u32*)((u8*)m->long_scratch + front) : nullptr, trace ? trace->pos : nullptr, trace ? trace->npos : nullptr, trace ? trace->stride : 0u, ggrid, st);
    if (pev) HIPCHK(hipEventRecord(pev[1], st));
    HIPCHK(hipGetLastError());
    return FZB_OK;
}

// ---- the pipeline of a SHORT needle (NeedleDev by value), one function per path ------------------------------------------------------
// counters: [0]=filter survivors [1]=kept by the lane-exact prefilter [3]=multi-chunk queue [4]=generic (greedy / wide unicode) queue
// [5]=marginal survivors [6]=rejected marginal survivors [8..10]=single-chunk classes [12..15]=multi-chunk tail classes
struct Pipe {
    fzb_matcher* m;
    const CorpusDev& cd;
    size_t first;
    u32 cnt, index_offset;
    const u32* items_in;     // item list (indices relative to `first`) or nullptr = the contiguous range
    const u32* n_items_in;
    fzb_match_rec* out;
    u32 cap32;
    uint32_t* dev_count;
    hipStream_t st;
    const TraceOut* trace;
    hipEvent_t* pev;         // profiling events of this call (nullptr: not profiled)
    int cus;
    // what the filter stage leaves for the scorers
    const u32* items = nullptr;
    const u32* win = nullptr;
    const u32* n_items_ptr = nullptr;
    int wmode = 0;
    bool classified = false;  // the compaction launch classified its survivors too (k_compact1_classify): pipe_score_ascii starts at the class scorers
};
#define FZB_STAGE(name)                                                                                        \
    do {                                                                                                       \
        if (fzb_knobs().debug_sync) { /* debugging aid: synchronise and report after every stage */            \
            hipError_t e_ = hipStreamSynchronize(p.st);                                                        \
            fprintf(stderr, "[fzb] stage %s: %s\n", name, hipGetErrorString(e_));                              \
            if (e_ != hipSuccess) return fail(FZB_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e_)); \
        }                                                                                                      \
    } while (0)
#define FZB_PEV(i)                                                    \
    do {                                                              \
        if (p.pev) HIPCHK(hipEventRecord(p.pev[i], p.st));            \
    } while (0)

// literal modes: accept pass (one bit per haystack) -> compaction -> scoring pass over the survivors (kernels_literal.hip)
static int pipe_literal(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const NeedleDev& nd = m->nd;
    u32* cnt_c = w.counters;
    if (m->literal_mode == FZB_MATCH_SUBSTRING && !nd.unicode && !p.items_in)  // the streaming DFA filter over the needle's KMP automaton
        fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.dfa, m->lc.dead_byte, nd.rows, 1, nd.rows, (u32)nd.nbytes, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, nullptr, nullptr, nullptr, nullptr,
                          m->lc.pad_ok, -1, m->cdfa_src == 1 ? w.cdfa : nullptr, (u32)m->cdfa.size(), m->cdfa_K, m->cdfa_G);
    else
        fzb_launch_literal_filter(p.cd, p.first, p.cnt, p.items_in, p.n_items_in, nd, m->literal_mode, w.bitmap, w.tile_counts, p.cus * 8, p.st);
    FZB_STAGE("literal filter");
    fzb_launch_compact1(w.bitmap, w.tile_counts, p.cnt, p.items_in ? p.n_items_in : nullptr, p.items_in, w.surv_idx, &cnt_c[0], p.cus * 2, p.st);
    FZB_STAGE("literal compact");
    fzb_launch_literal_score(p.cd, p.first, p.index_offset, w.surv_idx, &cnt_c[0], nd, m->literal_mode, p.out, p.cap32, p.dev_count, p.trace ? p.trace->pos : nullptr,
                             p.trace ? p.trace->npos : nullptr, p.trace ? p.trace->stride : 0u, p.cus * 4, p.st);
    FZB_STAGE("literal score");
    HIPCHK(hipGetLastError());
    return FZB_OK;
}

// Typo configuration, every haystack fits half a score chunk (C3): the LCS criterion decides in the stream (with the "nothing to spare" bit when
// haystacks can span several PREFILTER chunks: only those marginal survivors are re-decided at the exact lane width, a reject sets a bit), and
// the short scorer computes the lane-free window itself (DESIGN.md "Typo configurations").  A haystack that fits ONE prefilter chunk is decided
// exactly by the criterion: the reference's multi-path scan loses a candidate only when a lower path that found nothing more in the current
// chunk advances again in a later one (tests/test_oracle_reference_properties.py::test_single_chunk_typo_prefilter_is_the_lcs_criterion).
static int pipe_typo_fast_path(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const LaunchCfg& lc = m->lc;
    const NeedleDev& nd = m->nd;
    u32* cnt_c = w.counters;
    const int need = nd.rows - nd.max_typos;
    const u32 ntiles = (p.cnt + FZB_TILE - 1) / FZB_TILE;
    const RejectOut rej{w.reject_bits, w.tile_rejects, w.rej_prefix, &cnt_c[6]};
    const bool single_chunk = p.cd.max_len <= (u32)lc.pf_lanes;
    FZB_PEV(2);
    if (single_chunk && m->lcs_states)  // the LCS criterion as an automaton in the streaming DFA kernel
        fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.lcs_dfa, lc.dead_byte, m->lcs_states - 1, 1, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, nullptr,
                          nullptr, nullptr, nullptr, lc.pad_ok, m->lcs_acc_lo, m->cdfa_src == 3 ? w.cdfa : nullptr, (u32)m->cdfa.size(), m->cdfa_K, m->cdfa_G);
    else if (single_chunk)
        fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.dfa, lc.dead_byte, nd.rows, 2, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st);
    else
        fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.dfa, lc.dead_byte, nd.rows, 2, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, w.bitmap_m,
                          w.tile_counts_m, w.reject_bits, w.tile_rejects);
    FZB_PEV(3);
    FZB_STAGE("filter(lcs)");
    fzb_launch_compact1(w.bitmap, w.tile_counts, p.cnt, nullptr, nullptr, w.surv_idx, &cnt_c[0], p.cus * 4, p.st);
    if (!single_chunk) {
        fzb_launch_compact1(w.bitmap_m, w.tile_counts_m, p.cnt, nullptr, nullptr, w.marg_list, &cnt_c[5], p.cus * 4, p.st);
        FZB_STAGE("compact1 x2");
        fzb_launch_window(p.cd, p.first, w.marg_list, &cnt_c[5], nd, lc.pf_lanes, nullptr, nullptr, nullptr, cnt_c, p.cus * 4, p.st, &rej);
        FZB_STAGE("window(decide)");
        fzb_launch_scan_rejects(w.tile_rejects, ntiles, &cnt_c[6], w.rej_prefix, p.st);
    }
    FZB_PEV(4);
    fzb_launch_dp(p.cd, p.first, p.index_offset, w.surv_idx, nullptr, &cnt_c[0], nd, lc.sw_lanes, 2, 3, lc.pad_ok, p.out, p.cap32, p.dev_count, w.overflow, p.cnt, cnt_c, p.cus, p.st, &rej);
    FZB_STAGE("dp(short, typo windows)");
    return FZB_OK;
}

// Filter stage of every other fuzzy query: streaming filter (or its item-list form) -> compaction [-> lane-exact window kernel -> second
// compaction when the stream stage was a superset].  Leaves p.items / p.win / p.n_items_ptr / p.wmode for the scorers.
static bool ascii_split_classes(const fzb_matcher* m, const CorpusDev& cd);
static u32 pipe_qcap(const Pipe& p);
static int pipe_filter_stage(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const LaunchCfg& lc = m->lc;
    const NeedleDev& nd = m->nd;
    u32* cnt_c = w.counters;
    p.n_items_ptr = &cnt_c[0];
    p.wmode = lc.window_mode;
    const int need = nd.rows - (nd.max_typos > 0 ? nd.max_typos : 0);
    int exact_wmode = 0;  // != 0: a unicode automaton decided exactly in the stream - no lane-exact window pass, the scorer computes the window (mode 1: 0 typos, 3: typos)
    // unicode typo query over a list whose haystacks all fit ONE prefilter chunk: the scalar-level LCS automaton IS the reference's decision
    // (build_scalar_lcs_dfa); longer lists keep it as the (tight) superset in front of the lane-exact window kernel
    const bool uni_typo_exact = nd.unicode && lc.filter_mode == 2 && m->lcs_scalar && m->lcs_states && lc.bias_ok && !p.trace && !fzb_knobs().typo_exact_window && p.cd.max_len != 0 &&
                                p.cd.max_len <= (u32)lc.pf_lanes;
    if (p.items_in) {
        if (lc.filter_mode == 0) {
            HIPCHK(hipMemcpyAsync(&cnt_c[0], p.n_items_in, 4, hipMemcpyDeviceToDevice, p.st));
            p.items = p.items_in;
        } else {
            fzb_launch_filter_items(p.cd, p.first, p.items_in, p.n_items_in, w.table, nd.rows, lc.filter_mode, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, p.cus * 4, p.st);
            FZB_STAGE("filter(items)");
            fzb_launch_compact1(w.bitmap, w.tile_counts, 0, p.n_items_in, p.items_in, w.surv_idx, &cnt_c[0], p.cus * 2, p.st);
            FZB_STAGE("compact1(items)");
            p.items = w.surv_idx;
        }
    } else if (lc.filter_mode == 0) {
        // nothing filtered (max_typos = None or >= rows): the survivors are the identity list (counters[0] = the range's size: set by the caller)
    } else if (m->uni_dfa_states && lc.bias_ok && !p.trace) {
        // unicode path, 0 typos: the exact prefilter as a byte-level DFA in the streaming filter; the scorer finds the window itself
        FZB_PEV(2);
        fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.uni_dfa, lc.dead_byte, m->uni_dfa_states - 1, 1, 0, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, nullptr,
                          nullptr, nullptr, nullptr, lc.pad_ok, -1, m->cdfa_src == 2 ? w.cdfa : nullptr, (u32)m->cdfa.size(), m->cdfa_K, m->cdfa_G);
        FZB_PEV(3);
        FZB_STAGE("filter(unicode dfa)");
        fzb_launch_compact1(w.bitmap, w.tile_counts, p.cnt, nullptr, nullptr, w.surv_idx, &cnt_c[0], p.cus * 4, p.st, &cnt_c[1]);  // (kept by the exact prefilter = the filter's survivors)
        FZB_STAGE("compact1");
        p.items = w.surv_idx;
        exact_wmode = 1;
    } else {
        FZB_PEV(2);
        if (lc.filter_mode == 2 && m->lcs_states)  // typo configurations: the LCS automaton in the streaming DFA kernels (short and ragged lists)
            fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.lcs_dfa, lc.dead_byte, m->lcs_states - 1, 1, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, nullptr,
                              nullptr, nullptr, nullptr, lc.pad_ok, m->lcs_acc_lo, m->cdfa_src == 3 ? w.cdfa : nullptr, (u32)m->cdfa.size(), m->cdfa_K, m->cdfa_G);
        else
            fzb_launch_filter(p.cd, p.first, p.cnt, w.table, w.dfa, lc.dead_byte, nd.rows, lc.filter_mode, need, (u32)nd.min_haystack_len, w.bitmap, w.tile_counts, w.counters, p.cus * 8, p.st, nullptr, nullptr,
                              nullptr, nullptr, lc.pad_ok, -1, (lc.filter_mode == 1 && m->cdfa_src == 1) ? w.cdfa : nullptr, (u32)m->cdfa.size(), m->cdfa_K, m->cdfa_G);
        FZB_PEV(3);
        FZB_STAGE("filter");
        // ragged ASCII list, exact filter (0 typos) or whole-haystack windows: compaction and classification in ONE launch (k_compact1's grid: its
        // workgroups classify the survivors they have just listed).  Step, us, two launches -> one (profiles/r06_fused_classify.txt): lists of 100 k /
        // 300 k paths 34.7 -> 27.7 / 41.1 -> 36.1, 1.4 M paths 89.0 -> 86.5, the C4 shard 384 -> 384 (2, 3 or 4 workgroups per CU alike; 8 per CU
        // lose 10 us on the two large lists: more workgroups than are resident at once, and each is a chain of dependent round trips)
        // ... for lists of up to two tiles per CU (~ 0.5 M items: where a launch is a fifth of the query).  Beyond, the fused form gains nothing (above) and has a
        // failure mode the two-launch form does not: a workgroup classifies what ITS tiles hold, so survivors that cluster in a few tiles - every file below
        // one directory - are classified by a few workgroups, round after round (12.5 M items, all survivors in one stretch: 16 rounds in 52 workgroups),
        // whereas k2w_classify spreads any survivor list evenly.  On a small list the worst case is two rounds.
        const bool fuse = !nd.unicode && !p.trace && lc.filter_exact && !uni_typo_exact && (p.wmode == 1 || p.wmode == 2) && !fzb_knobs().no_fused_classify && w.cls_win && w.cls_lists &&
                          p.cnt <= (u32)p.cus * 2u * FZB_TILE && ascii_split_classes(m, p.cd);
        if (fuse) {
            fzb_launch_compact1_classify(p.cd, p.first, w.bitmap, w.tile_counts, p.cnt, w.surv_idx, &cnt_c[0], nd, lc.sw_lanes, p.wmode, p.cap32, p.dev_count, w.overflow, pipe_qcap(p), cnt_c,
                                         w.cls_win, w.cls_lists, (u32)w.cap_cls, p.cus * 4, p.st, 1);
            p.classified = true;
            FZB_STAGE("compact1 + classify");
        } else {
            fzb_launch_compact1(w.bitmap, w.tile_counts, p.cnt, nullptr, nullptr, w.surv_idx, &cnt_c[0], p.cus * 4, p.st, uni_typo_exact ? &cnt_c[1] : nullptr);
            FZB_STAGE("compact1");
        }
        p.items = w.surv_idx;
        if (uni_typo_exact) exact_wmode = 3;
    }
    if (exact_wmode) {
        p.wmode = exact_wmode;
    } else if (!lc.filter_exact) {
        // the stream stage was a superset: re-decide every survivor with the reference's chunked algorithm at its exact lane width
        fzb_launch_window(p.cd, p.first, p.items, &cnt_c[0], nd, lc.pf_lanes, w.win, w.bitmap2, w.tile_counts2, cnt_c, p.cus * 4, p.st, nullptr, p.items_in ? 0u : p.cnt);
        FZB_STAGE("window");
        fzb_launch_compact2(w.bitmap2, w.tile_counts2, &cnt_c[0], p.items, w.win, w.items2, w.win2, &cnt_c[1], p.cus * 2, p.st);
        FZB_STAGE("compact2");
        p.items = w.items2;
        p.win = w.win2;
        p.n_items_ptr = &cnt_c[1];
        p.wmode = 0;
    }
    return FZB_OK;
}

// The queue of windows wider than one chunk: multi-chunk entries from the front, generic-kernel entries from the back.  Every item is queued at
// most once by its scorer (front + back <= cnt), and the thread-per-haystack unicode scorer may hand up to FZB_UNICODE_FWD_CAP of the front's
// windows on to the back WHILE the front is still being read: the back gets that many entries of room of its own below the front's reach
// (ensure_workspace: the allocation holds at least count + FZB_UNICODE_FWD_CAP entries)
// Ragged ASCII list whose scorers are ONE launch over the classifier's lists (k2_classes_all: single-chunk classes + multi-chunk windows by the
// width of their last chunk's tail): classified scoring, windows wider than a chunk possible, dp_cfm.h's form for them
static bool ascii_split_classes(const fzb_matcher* m, const CorpusDev& cd) {
    const LaunchCfg& lc = m->lc;
    const FzbKnobs& kn = fzb_knobs();
    const bool no_wide = cd.max_len != 0 && cd.max_len <= (u32)lc.sw_lanes;
    const bool classes = lc.cf_ok && !kn.no_dp_classes && !fzb_dp_short_applies(cd, lc.sw_lanes, 2);
    return classes && !no_wide && lc.cfm_ok && !kn.no_dp_cfm;
}
static u32 pipe_qcap(const Pipe& p) { return (u32)std::min<u64>((u64)p.cnt + FZB_UNICODE_FWD_CAP, 0xFFFFFFFFull); }

// matched indices: one traced generic scorer for every window width, ASCII and unicode (kernels_generic.hip)
static int pipe_score_traced(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const int tgrid = (int)std::max<size_t>(1, std::min<size_t>((size_t)p.cus / 2, ((size_t)p.cnt + 3) / 4));
    const size_t words = fzb_trace_scratch_words(m->nd, tgrid);
    if (w.trace_cells_words < words) {
        if (w.trace_cells) HIPCHK(hipFree(w.trace_cells));
        w.trace_cells = nullptr;
        w.trace_cells_words = 0;
        HIPCHK(dev_alloc((void**)&w.trace_cells, words * 4));
        w.trace_cells_words = words;
    }
    fzb_launch_generic_trace(p.cd, p.first, p.index_offset, p.items, p.win, p.wmode, p.n_items_ptr, m->nd, m->lc.sw_lanes, m->nd.unicode, p.out, p.cap32, p.dev_count, w.counters, w.trace_cells,
                             p.trace->pos, p.trace->npos, p.trace->stride, tgrid, p.st);
    FZB_STAGE("generic(trace)");
    return FZB_OK;
}

// Unicode scorers.  Single-chunk windows: k2u_dp_unicode (thread per haystack).  Windows wider than a chunk: up to 1024 bytes into the FRONT of the
// queue, beyond that (greedy fallback) into its back.  The front has two takers, chosen on the device by its length: the thread-per-haystack
// multi-chunk scorer (k2u_dp_unicode_multi: several times fewer instructions, but one wave per SIMD and ~ 10-20 us per chunk - a fixed latency of
// ~ 100 us, then 0.3 ns per window) from `umin` windows on, the wave-per-haystack kernel (1.6 - 2.7 ns per window, no fixed cost) below
// (45 k windows 0.215 ms wave per haystack / 0.204 thread per haystack, 361 k windows 1.285 / 0.815).  FZB_UNICODE_MULTI=0 / 1: never / always.
static int pipe_score_unicode(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const LaunchCfg& lc = m->lc;
    const NeedleDev& nd = m->nd;
    const FzbKnobs& kn = fzb_knobs();
    const CorpusDev& cd = p.cd;
    u32* cnt_c = w.counters;
    const int cus = p.cus;
    const u32 qcap = pipe_qcap(p);
    const bool no_wide = cd.max_len != 0 && cd.max_len <= (u32)lc.sw_lanes;  // no haystack is longer than a chunk
    int rc;
    // (a range smaller than the switch point cannot queue that many windows: the thread-per-haystack scorer is not even launched)
    const u32 umin = no_wide ? 0xFFFFFFFFu : kn.unicode_multi == 0 ? 0xFFFFFFFFu : kn.unicode_multi == 1 ? 0u : p.cnt < (u32)cus * 128u ? 0xFFFFFFFFu : (u32)cus * 128u;
    const int ugrid = cus * 2;  // multi-chunk unicode scorer: one wave per SIMD (two-wave workgroups)
    if (umin != 0xFFFFFFFFu && (rc = ensure_dp_scratch(m, ugrid))) return rc;  // first use only (or fzb_matcher_reserve)
    // Whole-haystack windows (max_typos: None): the wide ones are known from the end offsets, so they are queued FIRST and their scorers run
    // on the second stream beside the single-chunk scorer (Arabic-shaped list, All Scores: the two took 90 + 100 us one after the other)
    bool presplit = p.wmode == 2 && !p.items && !no_wide && !kn.debug_sync;  // (no item list: the number of windows is the range's size)
    if (presplit && ensure_aux_stream(m) != FZB_OK) {  // no second stream: the scorer queues them itself (the error text is dropped with the fallback)
        presplit = false;
        fzb_clear_error();
    }
    hipStream_t wst = p.st;  // the stream of the wide windows' scorers
    if (presplit) {
        fzb_launch_unicode_split_wide(cd, p.first, p.items, p.n_items_ptr, lc.sw_lanes, p.cap32, w.overflow, qcap, cnt_c, (int)std::min<u32>((p.cnt + 2047u) / 2048u, (u32)cus * 4u), p.st);
        HIPCHK(hipEventRecord(m->ev_fork, p.st));
        HIPCHK(hipStreamWaitEvent(m->aux_stream, m->ev_fork, 0));
        wst = m->aux_stream;
    }
    // (presplit: the wide windows' kernels are enqueued FIRST - a few hundred long-running waves that take their SIMDs and keep them - and the
    // single-chunk scorer runs as one short-lived workgroup per 128 items, which fills the rest of the chip and every SIMD a wide wave frees)
    auto launch_single = [&]() {
        fzb_launch_dp_unicode(cd, p.first, p.index_offset, p.items, p.win, p.n_items_ptr, nd, lc.sw_lanes, p.wmode, p.out, p.cap32, p.dev_count, w.overflow, qcap, cnt_c, cus, p.st, lc.cfu_ok,
                              presplit ? 2 : 1, presplit ? (int)((p.cnt + 127) / 128) : 0);
    };
    if (!presplit) launch_single();
    FZB_STAGE("dp(unicode)");
    if (no_wide) return FZB_OK;
    // (the thread-per-haystack scorer hands windows beyond four chunks - up to FZB_UNICODE_FWD_CAP of them - on to the queue's back)
    const u32 fwd_cap = umin != 0xFFFFFFFFu ? FZB_UNICODE_FWD_CAP : 0u;
    if (umin != 0xFFFFFFFFu)
        fzb_launch_dp_unicode_multi(cd, p.first, p.index_offset, w.overflow, &cnt_c[3], nd, lc.sw_lanes, p.out, p.cap32, w.dp_scratch, ugrid, wst, umin, lc.cfu_ok, cnt_c, w.overflow + 4 * (size_t)qcap, fwd_cap);
    // The queue's back - handed-on stragglers (DP) and windows beyond 1024 bytes (greedy) - has a launch of its own only where windows beyond
    // 1024 bytes can exist; otherwise the front's wave-per-haystack launch (12 workgroups per CU: its LDS follows the needle's rows, so its
    // registers decide the occupancy), which has nothing to do exactly when the thread-per-haystack scorer ran, walks the back then
    const bool back_launch = !(cd.max_len != 0 && cd.max_len <= FZB_MAX_HAYSTACK_LEN);
    const bool may_fwd = fwd_cap != 0 && !(cd.max_len != 0 && cd.max_len <= 4u * (u32)lc.sw_lanes);
    if (umin != 0u)
        fzb_launch_generic(cd, p.first, p.index_offset, p.items, p.win, p.wmode, w.overflow, &cnt_c[3], nd, lc.sw_lanes, 1, p.out, p.cap32, nullptr, cnt_c, cus * 12, wst, 1, umin,
                           (may_fwd && !back_launch) ? w.overflow + 4 * (size_t)qcap : nullptr, &cnt_c[4]);
    FZB_STAGE("dp(unicode, wide windows)");
    if (back_launch || (may_fwd && umin == 0u)) {
        fzb_launch_generic(cd, p.first, p.index_offset, p.items, p.win, p.wmode, w.overflow + 4 * (size_t)qcap, &cnt_c[4], nd, lc.sw_lanes, 1, p.out, p.cap32, nullptr, cnt_c,
                           may_fwd ? cus * 4 : cus / 4 + 1, wst);
        FZB_STAGE("generic(unicode, stragglers + greedy)");
    }
    if (presplit) {
        launch_single();
        HIPCHK(hipEventRecord(m->ev_join, m->aux_stream));
        HIPCHK(hipStreamWaitEvent(p.st, m->ev_join, 0));
    }
    return FZB_OK;
}

// ASCII scorers.  Lists whose haystacks fit half a chunk: k2b_dp_short (inside fzb_launch_dp).  Ragged lists under dp_cf.h's preconditions:
// k2w_classify finds window and class per survivor, then ONE launch (k2_classes_all) scores the three single-chunk classes and the
// multi-chunk windows by the width of their last chunk's tail (DESIGN.md "Scorers of a ragged list").  Scorings outside dp_cfm.h's preconditions
// keep the classes but send multi-chunk windows through the queue (k2d_dp_multi on the second stream beside the class launches); scorings
// outside dp_cf.h's run the per-wave form k2b_dp.  Windows beyond 1024 bytes: the greedy fallback in the wave-per-haystack kernel.
static int pipe_score_ascii(Pipe& p) {
    fzb_matcher* m = p.m;
    Workspace& w = m->ws;
    const LaunchCfg& lc = m->lc;
    const NeedleDev& nd = m->nd;
    const FzbKnobs& kn = fzb_knobs();
    const CorpusDev& cd = p.cd;
    u32* cnt_c = w.counters;
    const int cus = p.cus;
    const u32 qcap = pipe_qcap(p);
    const bool no_wide = cd.max_len != 0 && cd.max_len <= (u32)lc.sw_lanes;  // no haystack is longer than a chunk
    int rc;
    const bool classes = lc.cf_ok && !kn.no_dp_classes && !fzb_dp_short_applies(cd, lc.sw_lanes, 2);
    const int mgrid = cus * 4;  // multi-chunk scorer: 2 waves per SIMD (the kernel is capped at 256 VGPRs)
    const int mmode = (lc.cfm_ok && !kn.no_dp_cfm) ? 2 : lc.bias_ok ? 1 : 0;
    if (!no_wide && (rc = ensure_dp_scratch(m, mgrid))) return rc;  // first use only (or fzb_matcher_reserve)
    const int split = classes && !no_wide && mmode == 2;  // multi-chunk windows as k2w_classify's tail-class lists (needs dp_cfm.h's form; cf_ok includes pad_ok) == ascii_split_classes()
    if (p.classified && !split) return fail(FZB_ERR_INVALID, "internal: the compaction classified a list whose scorers do not take class lists");
    bool fork = classes && !no_wide && !split;
    if (fork && ensure_aux_stream(m) != FZB_OK) {  // no second stream: everything on the caller's stream (the error text is dropped with the fallback)
        fork = false;
        fzb_clear_error();
    }
    if (split) {
        if (!p.classified)
            fzb_launch_dp_classes(cd, p.first, p.index_offset, p.items, p.win, p.n_items_ptr, nd, lc.sw_lanes, p.wmode, p.out, p.cap32, p.dev_count, w.overflow, qcap, cnt_c, w.cls_win, w.cls_lists,
                                  (u32)w.cap_cls, cus, p.st, 1, split);
        fzb_launch_classes_all(cd, p.first, p.index_offset, p.items, w.cls_win, w.cls_lists, (u32)w.cap_cls, cnt_c, nd, lc.sw_lanes, p.out, p.cap32, w.dp_scratch, mgrid, cus, p.st);
    } else if (classes)
        fzb_launch_dp_classes(cd, p.first, p.index_offset, p.items, p.win, p.n_items_ptr, nd, lc.sw_lanes, p.wmode, p.out, p.cap32, p.dev_count, w.overflow, qcap, cnt_c, w.cls_win, w.cls_lists,
                              (u32)w.cap_cls, cus, p.st, fork ? 1 : 0, 0);
    else
        fzb_launch_dp(cd, p.first, p.index_offset, p.items, p.win, p.n_items_ptr, nd, lc.sw_lanes, lc.cf_ok ? 2 : lc.bias_ok ? 1 : 0, p.wmode, lc.pad_ok, p.out, p.cap32, p.dev_count, w.overflow, qcap, cnt_c, cus, p.st);
    FZB_STAGE("dp");
    if (fork) {  // the queued multi-chunk windows on the second stream beside the three class launches (both start from the classifier's output, disjoint records)
        HIPCHK(hipEventRecord(m->ev_fork, p.st));
        HIPCHK(hipStreamWaitEvent(m->aux_stream, m->ev_fork, 0));
        fzb_launch_dp_multi(cd, p.first, p.index_offset, w.overflow, &cnt_c[3], nd, lc.sw_lanes, mmode, p.out, p.cap32, w.dp_scratch, mgrid, m->aux_stream);
        HIPCHK(hipEventRecord(m->ev_join, m->aux_stream));
        fzb_launch_dp_classes(cd, p.first, p.index_offset, p.items, p.win, p.n_items_ptr, nd, lc.sw_lanes, p.wmode, p.out, p.cap32, p.dev_count, w.overflow, qcap, cnt_c, w.cls_win, w.cls_lists,
                              (u32)w.cap_cls, cus, p.st, 2, 0);
        HIPCHK(hipStreamWaitEvent(p.st, m->ev_join, 0));
        FZB_STAGE("dp classes + dp_multi (second stream)");
    }
    if (no_wide) return FZB_OK;
    if (!fork && !split) {
        fzb_launch_dp_multi(cd, p.first, p.index_offset, w.overflow, &cnt_c[3], nd, lc.sw_lanes, mmode, p.out, p.cap32, w.dp_scratch, mgrid, p.st);
        FZB_STAGE("dp_multi");
    }
    if (!(cd.max_len != 0 && cd.max_len <= FZB_MAX_HAYSTACK_LEN)) {  // > 1024-byte windows: the greedy fallback
        fzb_launch_generic(cd, p.first, p.index_offset, p.items, p.win, p.wmode, w.overflow + 4 * (size_t)qcap, &cnt_c[4], nd, lc.sw_lanes, 0, p.out, p.cap32, nullptr, cnt_c, cus / 4 + 1, p.st);
        FZB_STAGE("generic(greedy)");
    }
    return FZB_OK;
}

// events of a profiled call: [0] start, [2]..[3] around the streaming filter, [4] before the scorers, [1] end
static int pipe_profile_begin(fzb_matcher* m, bool has_filter, hipStream_t st, hipEvent_t** pev) {
    *pev = nullptr;
    if (!m->profiling) return FZB_OK;
    const int slot = (int)(m->prof_calls % fzb_matcher::PROF_SLOTS);
    hipEvent_t* ev = m->evring[slot];
    m->ev_filter[slot] = has_filter ? 1 : 0;
    m->prof_calls++;
    for (int i = 0; i < 5; i++)
        if (!ev[i]) HIPCHK(hipEventCreate(&ev[i]));
    HIPCHK(hipEventRecord(ev[0], st));
    *pev = ev;
    return FZB_OK;
}

// The pipeline.  items_in == nullptr: the haystacks are the contiguous range [first, first + count).  Otherwise they are the listed ones,
// items_in[j] = index relative to `first`, *n_items_in of them (a device-side count <= count): the narrowing step of the multi-pattern
// composition.  `trace` != nullptr: the matched-indices form - every record also gets its matched byte positions (the traced generic scorer
// replaces the fast scorers; literal modes write the needle run).
static int run_pipeline(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, const u32* items_in, const u32* n_items_in,
                        fzb_match* dev_out, size_t capacity, uint32_t* dev_count, void* stream, const TraceOut* trace = nullptr) {
    if (!m || !c || !dev_count || (!dev_out && capacity)) return fail(FZB_ERR_INVALID, "null argument");
    // an item list may repeat haystacks, so only the contiguous form bounds `count` by the corpus
    if (first > c->dev.n || (!items_in && count > c->dev.n - first)) return fail(FZB_ERR_INVALID, "range outside the corpus");
    // guard_against_haystack_overflow (src/matcher/mod.rs:438-446)
    if ((u64)count + (u64)index_offset > 0xFFFFFFFFull)
        return fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string((u64)count + index_offset) + " > 4294967295 (index offset: " + std::to_string(index_offset) + ")");
    if (m->empty) return fail(FZB_ERR_INVALID, "empty needle: handled on the host by fzb_match_list / fzb_match_list_into");
    hipStream_t st = (hipStream_t)stream;
    int rc = fzb_bind_device(m);
    if (rc) return rc;
    rc = ensure_workspace(m, count, st, true);
    if (rc) return rc;
    Workspace& w = m->ws;
    const LaunchCfg& lc = m->lc;
    const u32 cap32 = (u32)std::min<size_t>(capacity, 0xFFFFFFFFu);
    // the streaming filter kernels clear the counter block themselves (one launch less on the hot path)
    const bool filter_resets = count != 0 && !items_in && !m->long_needle && (m->literal_mode ? (m->literal_mode == FZB_MATCH_SUBSTRING && !m->nd.unicode) : lc.filter_mode != 0);
    // nothing filtered (max_typos = None or >= rows): the survivors are the identity list - the counter block is cleared and its first word set
    // to the range's size by ONE small kernel (a memset and a 32-bit fill were two fill kernels with a dispatch gap each: ~ 20 us of a step)
    const bool identity_list = count != 0 && !items_in && !m->long_needle && !m->literal_mode && lc.filter_mode == 0;
    if (identity_list) fzb_launch_init_counters(w.counters, (u32)count, st);
    else if (!filter_resets) HIPCHK(hipMemsetAsync(w.counters, 0, 64, st));
    if (count == 0) {
        HIPCHK(hipMemsetAsync(dev_count, 0, 8, st));
        return FZB_OK;
    }
    if (m->long_needle) return run_pipeline_long(m, c, first, count, index_offset, items_in, n_items_in, dev_out, cap32, dev_count, st, trace);
    Pipe p{m, c->dev, first, (u32)count, index_offset, items_in, n_items_in, (fzb_match_rec*)dev_out, cap32, dev_count, st, trace, nullptr, lc.num_cus};
    if (m->literal_mode) return pipe_literal(p);
    if ((rc = pipe_profile_begin(m, lc.filter_mode && !items_in, st, &p.pev))) return rc;
    if (!items_in && lc.filter_mode != 0 && typo_fast_path_configured(m) && !trace && fzb_dp_short_applies(c->dev, lc.sw_lanes, 2)) {
        rc = pipe_typo_fast_path(p);
    } else {
        if ((rc = pipe_filter_stage(p))) return rc;
        FZB_PEV(4);
        if (trace) rc = pipe_score_traced(p);
        else if (m->nd.unicode && lc.bias_ok) rc = pipe_score_unicode(p);
        else if (m->nd.unicode) {
            fzb_launch_generic(c->dev, first, index_offset, p.items, p.win, p.wmode, nullptr, p.n_items_ptr, m->nd, lc.sw_lanes, 1, p.out, cap32, dev_count, w.counters, lc.num_cus * 4, st);
            FZB_STAGE("generic(unicode)");
        } else rc = pipe_score_ascii(p);
    }
    if (rc) return rc;
    FZB_PEV(1);
    HIPCHK(hipGetLastError());
    return FZB_OK;
}
#undef FZB_STAGE
#undef FZB_PEV

// Sizes every device buffer a query over `c` can need (range workspace incl. the typo-path arrays, multi-chunk scorer scratch when
// the corpus has haystacks wider than a chunk, staging + sort buffers of the synchronous / sorted entry points), so that the queries
// that follow - also after fzb_matcher_set_pattern / fzb_matcher_set_config, which keep the workspace - never allocate.
// (Matched-indices queries size their trace scratch from the selection, on first use.)
int fzb_matcher_reserve(fzb_matcher* m, const fzb_corpus* c) {
    if (!m || !c) return fail(FZB_ERR_INVALID, "null argument");
    if (m->empty || c->dev.n == 0) return FZB_OK;
    int rc = fzb_bind_device(m);
    if (rc) return rc;
    const size_t n = c->dev.n;
    rc = ensure_workspace(m, n);
    if (rc) return rc;
    const bool no_wide = c->dev.max_len != 0 && c->dev.max_len <= (u32)m->lc.sw_lanes;
    if (!m->long_needle && !m->literal_mode && !m->nd.unicode && !no_wide && ((rc = ensure_dp_scratch(m, m->lc.num_cus * 4)) || (rc = ensure_aux_stream(m)))) return rc;
    if (!m->long_needle && !m->literal_mode && m->nd.unicode && m->lc.bias_ok && !no_wide && fzb_knobs().unicode_multi != 0 && (rc = ensure_dp_scratch(m, m->lc.num_cus * 2))) return rc;
    if (!m->long_needle && !m->literal_mode && m->nd.unicode && m->lc.bias_ok && !no_wide && m->nd.max_typos < 0 && (rc = ensure_aux_stream(m))) return rc;  // whole-haystack windows: the wide ones on the second stream
    if ((rc = fzb_ensure_out_staging(m, n))) return rc;
    if ((rc = ensure_sort_buffers(m, n))) return rc;
    return FZB_OK;
}

int fzb_match_list_device(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match* dev_out, size_t capacity,
                          uint32_t* dev_count, void* stream) {
    return run_pipeline(m, c, first, count, index_offset, nullptr, nullptr, dev_out, capacity, dev_count, stream);
}

int fzb_match_list_sorted_device(fzb_matcher* m, const fzb_corpus* c, fzb_match* dev_out, size_t capacity, uint32_t* dev_count, void* stream) {
    if (!m || !c) return fail(FZB_ERR_INVALID, "null argument");
    return fzb_sorted_range_device(m, c, 0, c->dev.n, 0, dev_out, capacity, dev_count, stream);
}

}  // extern "C"
// The ordered form over any sub-range, records numbered from index_offset: what one worker of `match_list_parallel` produces for its
// share (per-run reverse / radix sort, src/matcher/parallel.rs:66-76) - fzb_match_list_parallel_sharded runs one per device
int fzb_sorted_range_device(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match* dev_out, size_t capacity, uint32_t* dev_count,
                            void* stream) {
    if (!m || !c) return fail(FZB_ERR_INVALID, "null argument");
    if (first > c->dev.n || count > c->dev.n - first) return fail(FZB_ERR_INVALID, "range outside the corpus");
    const size_t cap = std::min<size_t>(capacity, count);
    OrderPlan plan;
    int rc;
    // (the range workspace first: growing it releases every workspace buffer, the sort's included)
    if (!m->empty && count != 0 && ((rc = fzb_bind_device(m)) || (rc = ensure_workspace(m, count, (hipStream_t)stream, true)))) return rc;
    if ((rc = fzb_order_begin(m, m->empty || count == 0 ? 0 : cap, (fzb_match_rec*)dev_out, &plan))) return rc;
    rc = fzb_match_list_device(m, c, first, count, index_offset, (fzb_match*)plan.in, plan.via_tmp ? cap : capacity, dev_count, stream);
    if (rc) return rc;
    if (count == 0) return FZB_OK;
    return fzb_order_finish(m, plan, (fzb_match_rec*)dev_out, dev_count, (hipStream_t)stream);
}

// The ordering post-step of `match_list` (src/matcher/mod.rs:215-221: reverse for the *Desc strategies, radix_sort_matches for the Score*
// strategies) over index-ordered records that are, or are about to be, in device memory.  fzb_order_begin sizes the sort's buffers for
// `cap` records and says where the producer should write them (`plan.in`): with a single radix pass that is the sort's SECOND buffer and
// the pass scatters into the caller's array - no copy back.  fzb_order_finish launches reverse / sort on `stream`; the record count is
// read from device memory.  Producers: the pipeline (above), the concatenation of per-shard runs (host_shard.hip).
int fzb_order_begin(fzb_matcher* m, size_t cap, fzb_match_rec* dev_out, OrderPlan* p) {
    const int sort = m->config.sort;
    p->reversed = sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;          // src/matcher/mod.rs:215-217
    p->by_score = sort == FZB_SORT_SCORE_THEN_INDEX_ASC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;  // :218-220
    // one radix pass is enough when no score can reach 256 (Scoring::guard's bound on the matrix + the exact-match bonus added after it)
    p->one_pass = !m->literal_mode && max_matrix_score(m->config.scoring, (size_t)m->rows) + (size_t)m->config.scoring.exact_match_bonus < 256;
    p->via_tmp = p->by_score && p->one_pass && cap != 0;
    p->in = dev_out;
    if (p->by_score) {
        int rc = ensure_sort_buffers(m, cap);
        if (rc) return rc;
        if (p->via_tmp) p->in = m->ws.sort_tmp;
    }
    return FZB_OK;
}
int fzb_order_finish(fzb_matcher* m, const OrderPlan& p, fzb_match_rec* dev_out, const u32* dev_count, hipStream_t stream) {
    if (!p.reversed && !p.by_score) return FZB_OK;
    Workspace& w = m->ws;
    if (p.by_score && !w.sort_tmp) return fail(FZB_ERR_INVALID, "fzb_order_finish without fzb_order_begin");
    fzb_launch_sort(dev_out, w.sort_tmp, dev_count, w.sort_hist, (u32)(w.sort_cap / 2048 + 2), p.reversed, p.by_score, m->lc.num_cus * 2, stream, p.via_tmp ? -1 : p.one_pass ? 1 : 2);
    HIPCHK(hipGetLastError());
    return FZB_OK;
}
extern "C" {

// Result lists handed to the caller live in page-locked host memory so that the device-to-host copy of the records runs at
// DMA speed (a pageable destination is staged through bounce buffers: ~3x slower for a 4 MB list).  Pinning is expensive, so
// the buffers are pooled: fzb_matches_free returns a buffer to the pool and a steady stream of queries keeps reusing the same
// few.  Lists that never were on the device (empty needle) come from malloc; fzb_matches_free tells the two apart.
}  // extern "C"
namespace {
struct PinnedPool {
    std::mutex mu;
    std::unordered_map<void*, size_t> live;             // handed out: pointer -> capacity in bytes
    std::vector<std::pair<void*, size_t>> free_list;    // ready for reuse
    static constexpr size_t kMaxFree = 32;  // result lists + the staging buffers of an upload (two per worker thread)
    void* get(size_t bytes) {
        std::lock_guard<std::mutex> g(mu);
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); i++)
            if (free_list[i].second >= bytes && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
        void* p = nullptr;
        size_t cap = 0;
        if (best != free_list.size()) {
            p = free_list[best].first;
            cap = free_list[best].second;
            free_list.erase(free_list.begin() + best);
        } else {
            cap = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 16);
            if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
        }
        live[p] = cap;
        return p;
    }
    bool put(void* p) {  // false: not one of ours
        std::lock_guard<std::mutex> g(mu);
        auto it = live.find(p);
        if (it == live.end()) return false;
        const size_t cap = it->second;
        live.erase(it);
        if (free_list.size() < kMaxFree) free_list.emplace_back(p, cap);
        else (void)hipHostFree(p);
        return true;
    }
};
PinnedPool& pinned_pool() {
    static PinnedPool* pool = new PinnedPool;  // never destroyed: the HIP runtime may already be gone at exit
    return *pool;
}
}  // namespace
// The wait of a synchronous entry point: a query's results arrive tens to hundreds of microseconds after the call, and a thread that went to sleep in
// hipStreamSynchronize wakes 10-20 us after the stream has drained; so the stream is POLLED first (hipStreamQuery, for at most FZB_SPIN_WAIT_US, default
// 1 ms - a cold 8 ms upload-and-query ends up blocking as before), then blocked on.
hipError_t fzb_stream_wait(hipStream_t st) {
    const int budget_us = fzb_knobs().spin_wait_us;
    if (budget_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t e = hipStreamQuery(st);
            if (e != hipErrorNotReady) return e;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= budget_us) break;
            __builtin_ia32_pause();
        }
        (void)hipGetLastError();  // (hipErrorNotReady is not an error of the caller's)
    }
    return hipStreamSynchronize(st);
}
// (dev_words: eight u32 in device memory, all copied to h.count_host; the record count is word `n_word`)
int fzb_fetch_records(FetchHint& h, const void* dev_records, const u32* dev_words, int n_word, size_t capacity, hipStream_t st, fzb_match** out, size_t* out_len) {
    if (!h.count_host) HIPCHK(hipHostMalloc((void**)&h.count_host, 32, hipHostMallocDefault));
    // the guess: the previous result's size and a little more (every speculated record that does not exist is copied for nothing - a
    // quarter more cost 19 us on a 4 MB result, more than the synchronisation it saves); the buffer has room for half as many again, so a
    // result that outgrew the guess usually only needs its remainder copied
    const size_t guess = h.last ? std::min(capacity, h.last + h.last / 64 + 64) : 0;
    size_t room = std::min(capacity, guess + guess / 2);
    fzb_match* r = (fzb_match*)pinned_pool().get(std::max<size_t>(room, 1) * sizeof(fzb_match));
    if (!r) return fail(FZB_ERR_HIP, "hipHostMalloc failed for the result list");
    hipError_t e = hipMemcpyAsync(h.count_host, dev_words, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && guess) e = hipMemcpyAsync(r, dev_records, guess * sizeof(fzb_match), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = fzb_stream_wait(st);
    const size_t n = e == hipSuccess ? (size_t)h.count_host[n_word] : 0;
    if (e == hipSuccess && n > guess) {  // the result outgrew the guess (or there was none): the rest in a second copy
        size_t have = guess;
        if (n > room) {
            pinned_pool().put(r);
            r = (fzb_match*)pinned_pool().get(n * sizeof(fzb_match));
            if (!r) return fail(FZB_ERR_HIP, "hipHostMalloc failed for the result list");
            have = 0;
        }
        e = hipMemcpyAsync(r + have, (const fzb_match*)dev_records + have, (n - have) * sizeof(fzb_match), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = fzb_stream_wait(st);
    }
    if (e != hipSuccess) {
        pinned_pool().put(r);
        return fail(FZB_ERR_HIP, std::string("device to host: ") + hipGetErrorString(e));
    }
    h.last = n;
    *out = r;
    *out_len = n;
    return FZB_OK;
}
void* fzb_pinned_get(size_t bytes) { return pinned_pool().get(bytes); }
bool fzb_pinned_put(void* p) { return pinned_pool().put(p); }
namespace {
// The result of a query -> the host: the record count and the records, into a pooled pinned buffer.  The count is not known when the
// copies are enqueued, so the records are copied SPECULATIVELY - as many as the matcher's previous query returned plus a quarter - right
// behind the count, and one synchronisation serves both (a re-query of a resident list - the next keystroke - returns about as many
// matches as the last one, usually fewer); only a result that outgrew the guess costs the second copy that every query used to pay.
int fetch_records(FetchHint& h, const void* dev_records, const u32* dev_count, size_t capacity, fzb_match** out, size_t* out_len) {
    return fzb_fetch_records(h, dev_records, dev_count, 0, capacity, nullptr, out, out_len);
}
}  // namespace
extern "C" {

int fzb_match_list_into(fzb_matcher* m, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match** out, size_t* out_len) {
    if (!m || !c || !out || !out_len) return fail(FZB_ERR_INVALID, "null argument");
    if (first > c->dev.n || count > c->dev.n - first) return fail(FZB_ERR_INVALID, "range outside the corpus");
    if ((u64)count + (u64)index_offset > 0xFFFFFFFFull)
        return fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string((u64)count + index_offset) + " > 4294967295 (index offset: " + std::to_string(index_offset) + ")");
    *out = nullptr;
    *out_len = 0;
    if (m->empty) {  // src/matcher/mod.rs:381-384
        fzb_match* r = (fzb_match*)malloc(std::max<size_t>(count, 1) * sizeof(fzb_match));
        if (!r) return fail(FZB_ERR_INVALID, "out of memory");
        for (size_t i = 0; i < count; i++) r[i] = fzb_match{(uint32_t)(index_offset + i), 0, 0, 0};
        *out = r;
        *out_len = count;
        return FZB_OK;
    }
    if (int rc_ = fzb_ensure_out_staging(m, count)) return rc_;
    int rc = fzb_match_list_device(m, c, first, count, index_offset, (fzb_match*)m->out_dev, m->out_cap, m->count_dev, nullptr);
    if (rc) return rc;
    return fetch_records(m->fetch, m->out_dev, m->count_dev, m->out_cap, out, out_len);
}

void fzb_radix_sort_matches(fzb_match* matches, size_t n) {  // src/sort.rs:6-40
    if (n < 2) return;
    std::vector<fzb_match> tmp(n);
    size_t hist[256];
    for (int pass = 0; pass < 2; pass++) {
        const int shift = pass * 8;
        fzb_match* src = pass == 0 ? matches : tmp.data();
        fzb_match* dst = pass == 0 ? tmp.data() : matches;
        memset(hist, 0, sizeof(hist));
        for (size_t i = 0; i < n; i++) hist[(src[i].score >> shift) & 0xFF]++;
        size_t off[256];
        size_t run = 0;
        for (int b = 255; b >= 0; b--) { off[b] = run; run += hist[b]; }  // descending buckets
        for (size_t i = 0; i < n; i++) dst[off[(src[i].score >> shift) & 0xFF]++] = src[i];
    }
}

int fzb_match_list(fzb_matcher* m, const fzb_corpus* c, fzb_match** out, size_t* out_len) {
    if (!m || !c || !out || !out_len) return fail(FZB_ERR_INVALID, "null argument");
    const int sort = m->config.sort;
    if (m->empty) {  // CompiledPatterns::Empty: every index, score 0, reversed if the strategy says so, never sorted (mod.rs:215-220, 381-384)
        int rc = fzb_match_list_into(m, c, 0, c->dev.n, 0, out, out_len);
        if (rc) return rc;
        if (sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC) std::reverse(*out, *out + *out_len);
        return FZB_OK;
    }
    const size_t count = c->dev.n;
    *out = nullptr;
    *out_len = 0;
    if (int rc_ = fzb_ensure_out_staging(m, count)) return rc_;
    // scoring AND the reverse / stable radix sort post-step run on the device; the host only receives the final list
    int rc = fzb_match_list_sorted_device(m, c, (fzb_match*)m->out_dev, m->out_cap, m->count_dev, nullptr);
    if (rc) return rc;
    return fetch_records(m->fetch, m->out_dev, m->count_dev, m->out_cap, out, out_len);
}

int fzb_match_list_parallel(fzb_matcher* m, const fzb_corpus* c, size_t threads, fzb_match** out, size_t* out_len) {
    if (!m || !c) return fail(FZB_ERR_INVALID, "null argument");
    if (threads == 0) return fail(FZB_ERR_PANIC, "threads must be positive");  // parallel.rs:24
    return fzb_match_list(m, c, out, out_len);
}

// ---- matched indices: Matcher::match_list_indices (src/matcher/mod.rs:234-275) ------------------------------------------------
}  // extern "C"
namespace {
// The haystack list of the *_indices entry points: a selection of the corpus, or all of it.
int check_selection(const fzb_corpus* c, const uint32_t* selection, size_t n_selection, size_t& count, uint32_t index_offset = 0) {
    count = selection ? n_selection : (size_t)c->dev.n;
    if ((u64)count + (u64)index_offset > 0xFFFFFFFFull)  // guard_against_haystack_overflow(haystacks.len(), 0), mod.rs:235
        return fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string((u64)count + index_offset) + " > 4294967295 (index offset: " +
                                       std::to_string(index_offset) + ")");
    for (size_t i = 0; i < n_selection; i++)
        if (selection[i] >= c->dev.n) return fail(FZB_ERR_INVALID, "selection entry outside the corpus");
    return FZB_OK;
}

// One pattern over the list, LIST order (match_list_indices_impl, algo.rs:196-227): recs[k].index = position in the list,
// its positions appended to `positions`.  An empty needle matches everything with no positions.
int indices_in_list_order(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t count, std::vector<fzb_match_indices>& recs, std::vector<u32>& positions) {
    recs.clear();
    if (m->empty) {
        recs.resize(count);
        for (size_t i = 0; i < count; i++) recs[i] = fzb_match_indices{(uint32_t)i, 0, 0, 0, (uint32_t)positions.size(), 0};
        return FZB_OK;
    }
    if (!count) return FZB_OK;
    const u32 stride = (u32)std::max(1, m->nd.nbytes);
    if (m->trace_cap < count || m->trace_pos_words < count * (size_t)stride) {
        for (void* p : {(void*)m->trace_sel, (void*)m->trace_pos, (void*)m->trace_npos})
            if (p) HIPCHK(hipFree(p));
        m->trace_sel = m->trace_pos = m->trace_npos = nullptr;
        m->trace_cap = m->trace_pos_words = 0;
        HIPCHK(dev_alloc((void**)&m->trace_sel, (count + 4) * 4));  // [count] = the list length
        HIPCHK(dev_alloc((void**)&m->trace_npos, count * 4));
        HIPCHK(dev_alloc((void**)&m->trace_pos, count * (size_t)stride * 4));
        m->trace_cap = count;
        m->trace_pos_words = count * (size_t)stride;
    }
    if (int rc_ = fzb_ensure_out_staging(m, count)) return rc_;
    const u32* items_dev = nullptr;
    const u32* n_items_dev = nullptr;
    if (selection) {
        const u32 n32 = (u32)count;
        HIPCHK(hipMemcpy(m->trace_sel, selection, count * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(m->trace_sel + count, &n32, 4, hipMemcpyHostToDevice));
        items_dev = m->trace_sel;
        n_items_dev = m->trace_sel + count;
    }
    const TraceOut tr{m->trace_pos, m->trace_npos, stride};
    int rc = run_pipeline(m, c, 0, count, 0, items_dev, n_items_dev, (fzb_match*)m->out_dev, m->out_cap, m->count_dev, nullptr, &tr);
    if (rc) return rc;
    u32 n = 0;
    HIPCHK(hipMemcpy(&n, m->count_dev, 4, hipMemcpyDeviceToHost));
    std::vector<fzb_match_rec> dev_recs(n);
    std::vector<u32> npos(n), pos((size_t)n * stride);
    if (n) {
        HIPCHK(hipMemcpy(dev_recs.data(), m->out_dev, (size_t)n * sizeof(fzb_match_rec), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(npos.data(), m->trace_npos, (size_t)n * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pos.data(), m->trace_pos, (size_t)n * stride * 4, hipMemcpyDeviceToHost));
    }
    recs.resize(n);
    size_t sel_at = 0;  // records come back in list order, so each one is the next selection entry that names its haystack
    for (u32 j = 0; j < n; j++) {
        u32 index = dev_recs[j].index;
        if (selection) {
            while (sel_at < count && selection[sel_at] != index) sel_at++;
            if (sel_at == count) return fail(FZB_ERR_HIP, "internal: record outside the selection");
            index = (u32)sel_at++;
        }
        const u32 len = std::min(npos[j], stride);
        recs[j] = fzb_match_indices{index, dev_recs[j].score, dev_recs[j].exact, 0, (uint32_t)positions.size(), len};
        positions.insert(positions.end(), pos.begin() + (size_t)j * stride, pos.begin() + (size_t)j * stride + len);
    }
    return FZB_OK;
}

// the ordering step (mod.rs:268-273) and the hand-over to the caller
int finish_indices(std::vector<fzb_match_indices>& recs, const std::vector<u32>& positions, int sort, bool sort_by_score, fzb_match_indices** out, size_t* out_len,
                   uint32_t** out_positions, uint32_t index_offset = 0) {
    if (index_offset)
        for (fzb_match_indices& r : recs) r.index += index_offset;
    const bool reversed = sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;
    const bool by_score = sort == FZB_SORT_SCORE_THEN_INDEX_ASC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;
    if (reversed) std::reverse(recs.begin(), recs.end());
    if (by_score && sort_by_score)  // `sort_by_key(|m| Reverse(m.score))`: stable
        std::stable_sort(recs.begin(), recs.end(), [](const fzb_match_indices& a, const fzb_match_indices& b) { return a.score > b.score; });
    fzb_match_indices* r = (fzb_match_indices*)malloc(std::max<size_t>(recs.size(), 1) * sizeof(fzb_match_indices));
    u32* p = (u32*)malloc(std::max<size_t>(positions.size(), 1) * 4);
    if (!r || !p) {
        free(r);
        free(p);
        return fail(FZB_ERR_INVALID, "out of memory");
    }
    if (!recs.empty()) memcpy(r, recs.data(), recs.size() * sizeof(fzb_match_indices));
    if (!positions.empty()) memcpy(p, positions.data(), positions.size() * 4);
    *out = r;
    *out_len = recs.size();
    *out_positions = p;
    return FZB_OK;
}
}  // namespace
extern "C" {

static int match_list_indices_impl(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, int sort, uint32_t index_offset,
                                   fzb_match_indices** out, size_t* out_len, uint32_t** out_positions) {
    if (!m || !c || !out || !out_len || !out_positions || (!selection && n_selection)) return fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    *out_len = 0;
    *out_positions = nullptr;
    size_t count = 0;
    int rc = check_selection(c, selection, n_selection, count, index_offset);
    if (rc) return rc;
    std::vector<fzb_match_indices> recs;
    std::vector<u32> positions;
    rc = indices_in_list_order(m, c, selection, count, recs, positions);
    if (rc) return rc;
    // CompiledPatterns::Empty returns before the score sort (mod.rs:237-246) - all scores are 0 anyway
    return finish_indices(recs, positions, sort, !m->empty, out, out_len, out_positions, index_offset);
}

int fzb_match_list_indices(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, fzb_match_indices** out, size_t* out_len,
                           uint32_t** out_positions) {
    return match_list_indices_impl(m, c, selection, n_selection, m ? m->config.sort : 0, 0, out, out_len, out_positions);
}

int fzb_match_list_indices_into(fzb_matcher* m, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, uint32_t index_offset, fzb_match_indices** out,
                                size_t* out_len, uint32_t** out_positions) {
    return match_list_indices_impl(m, c, selection, n_selection, FZB_SORT_INDEX_ASC, index_offset, out, out_len, out_positions);
}

void fzb_match_indices_free(fzb_match_indices* matches, uint32_t* positions) {
    free(matches);
    free(positions);
}

// ---- query syntax: Pattern::parse / Pattern::parse_query (src/pattern.rs:87-222) ---------------------------------------------
}  // extern "C"
namespace {
bool is_rust_whitespace(u32 c) {  // char::is_whitespace = Unicode White_Space
    return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F ||
           c == 0x205F || c == 0x3000;
}
void append_utf8(std::string& s, u32 cp) {
    u8 buf[4];
    const int n = encode_utf8(cp, buf);
    s.append((const char*)buf, (size_t)n);
}
struct ParsedAtom { std::string needle; bool negated; int matching; };
// one atom: `!` negates, `^` / `$` anchor, `'` asks for a substring, a bare negated atom is a substring too; `\x` escapes x
ParsedAtom parse_atom(const std::vector<u32>& atom) {
    std::vector<std::pair<u32, bool>> tok;  // (char, escaped)
    for (size_t i = 0; i < atom.size(); i++) {
        if (atom[i] == '\\' && i + 1 < atom.size()) { tok.push_back({atom[i + 1], true}); i++; }
        else tok.push_back({atom[i], false});
    }
    size_t lo = 0, hi = tok.size();
    auto first_is = [&](u32 op) { if (lo < hi && !tok[lo].second && tok[lo].first == op) { lo++; return true; } return false; };
    auto last_is = [&](u32 op) { if (lo < hi && !tok[hi - 1].second && tok[hi - 1].first == op) { hi--; return true; } return false; };
    ParsedAtom a;
    a.negated = first_is('!');
    const bool prefix = first_is('^');
    const bool substring = !prefix && first_is('\'');
    const bool suffix = last_is('$');
    for (size_t i = lo; i < hi; i++) {
        const u32 c = tok[i].first;
        const bool special = c == '!' || c == '^' || c == '\'' || c == '$' || is_rust_whitespace(c);
        if (tok[i].second && !special) a.needle.push_back('\\');  // a backslash before anything else stays literal
        append_utf8(a.needle, c);
    }
    a.matching = prefix && suffix ? FZB_MATCH_EXACT : prefix ? FZB_MATCH_PREFIX : suffix ? FZB_MATCH_SUFFIX : (substring || a.negated) ? FZB_MATCH_SUBSTRING : -1;
    return a;
}
}  // namespace
extern "C" {

int fzb_parse_query(const uint8_t* query_utf8, size_t query_len, fzb_pattern** out_patterns, size_t* out_n) {
    if (!out_patterns || !out_n || (query_len && !query_utf8)) return fail(FZB_ERR_INVALID, "null argument");
    std::vector<u32> cps;
    if (!decode_utf8(query_utf8, query_len, cps)) return fail(FZB_ERR_INVALID, "query is not valid UTF-8");
    std::vector<ParsedAtom> atoms;
    std::vector<u32> cur;
    bool in_atom = false, escaped = false;
    auto flush = [&]() {
        ParsedAtom a = parse_atom(cur);
        if (!a.needle.empty()) atoms.push_back(a);  // atoms with an empty needle (`!`, `^$`) are dropped
        cur.clear();
        in_atom = false;
    };
    for (u32 c : cps) {
        if (escaped) { escaped = false; cur.push_back(c); }
        else if (c == '\\') { in_atom = true; escaped = true; cur.push_back(c); }
        else if (is_rust_whitespace(c)) { if (in_atom) flush(); }
        else { in_atom = true; cur.push_back(c); }
    }
    if (in_atom) flush();
    fzb_pattern* arr = (fzb_pattern*)calloc(std::max<size_t>(atoms.size(), 1), sizeof(fzb_pattern));
    if (!arr) return fail(FZB_ERR_INVALID, "out of memory");
    for (size_t i = 0; i < atoms.size(); i++) {
        u8* n = (u8*)malloc(atoms[i].needle.size() + 1);
        if (!n) {
            for (size_t k = 0; k < i; k++) free((void*)arr[k].needle_utf8);
            free(arr);
            return fail(FZB_ERR_INVALID, "out of memory");
        }
        memcpy(n, atoms[i].needle.data(), atoms[i].needle.size());
        n[atoms[i].needle.size()] = 0;
        arr[i].needle_utf8 = n;
        arr[i].needle_len = atoms[i].needle.size();
        arr[i].negated = atoms[i].negated;
        arr[i].casing = -1;
        arr[i].unicode = -1;
        arr[i].matching = atoms[i].matching;
    }
    *out_patterns = arr;
    *out_n = atoms.size();
    return FZB_OK;
}

void fzb_patterns_free(fzb_pattern* patterns, size_t n) {
    if (!patterns) return;
    for (size_t i = 0; i < n; i++) free((void*)patterns[i].needle_utf8);
    free(patterns);
}

// ---- multi-pattern composition (src/matcher/multi.rs) ---------------------------------------------------
struct fzb_multi_matcher {
    fzb_config config{};
    struct Compiled { bool negated; fzb_matcher* m; };
    std::vector<Compiled> patterns;  // empty needles dropped (src/matcher/mod.rs:193-195)
    int num_cus = 0;
    // device buffers, grown on demand: two candidate lists (ping-pong), their lengths, the item list handed to the next pattern,
    // and the bitmap / per-tile counts of the negation's compaction
    size_t cap = 0;
    fzb_match_rec* cand[2] = {nullptr, nullptr};
    u32* counts = nullptr;  // [0],[4] = lengths of cand[0], cand[1]; [8] = hits of a negated pattern (each slot: count, untruncated total)
    u32* items = nullptr;
    u64* bitmap = nullptr;
    u32* tile_counts = nullptr;
    // ordering + staging for the synchronous API
    fzb_match_rec* out_dev = nullptr;
    size_t out_cap = 0;
    u32* count_dev = nullptr;
    fzb_match_rec* sort_tmp = nullptr;
    u32* sort_hist = nullptr;
    size_t sort_cap = 0;
    FetchHint fetch;
};

static void multi_free_buffers(fzb_multi_matcher* mm) {
    void* ptrs[] = {mm->cand[0], mm->cand[1], mm->counts, mm->items, mm->bitmap, mm->tile_counts};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    mm->cand[0] = mm->cand[1] = nullptr;
    mm->counts = mm->items = nullptr;
    mm->bitmap = nullptr;
    mm->tile_counts = nullptr;
    mm->cap = 0;
}

int fzb_multi_matcher_create(const fzb_config* config, const fzb_pattern* patterns, size_t n_patterns, fzb_multi_matcher** out) {
    if (!config || !out || (n_patterns && !patterns)) return fail(FZB_ERR_INVALID, "null argument");
    auto mm = new fzb_multi_matcher();
    mm->config = *config;
    for (size_t i = 0; i < n_patterns; i++) {
        const fzb_pattern& p = patterns[i];
        if (p.needle_len == 0) continue;  // Matcher::compile returns None for an empty needle
        if (!p.needle_utf8) { fzb_multi_matcher_free(mm); return fail(FZB_ERR_INVALID, "null needle"); }
        fzb_config rc = *config;  // PatternConfig::resolve (src/pattern.rs:250-262); `sort` is the matcher's and applies to the combined list only
        if (p.has_max_typos) rc.max_typos = p.max_typos;
        if (p.casing >= 0) rc.casing = p.casing;
        if (p.unicode >= 0) rc.unicode = p.unicode;
        if (p.has_scoring) rc.scoring = p.scoring;
        if (p.matching >= 0) rc.matching = p.matching;
        rc.sort = FZB_SORT_INDEX_ASC;
        fzb_matcher* m = nullptr;
        int rc_create = fzb_matcher_create(&rc, p.needle_utf8, p.needle_len, &m);
        if (rc_create) { fzb_multi_matcher_free(mm); return rc_create; }
        mm->patterns.push_back({p.negated != 0, m});
    }
    *out = mm;
    return FZB_OK;
}

void fzb_multi_matcher_free(fzb_multi_matcher* mm) {
    if (!mm) return;
    for (auto& p : mm->patterns) fzb_matcher_free(p.m);
    multi_free_buffers(mm);
    void* ptrs[] = {mm->out_dev, mm->count_dev, mm->sort_tmp, mm->sort_hist};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (mm->fetch.count_host) (void)hipHostFree(mm->fetch.count_host);
    delete mm;
}

size_t fzb_multi_matcher_len(const fzb_multi_matcher* mm) { return mm ? mm->patterns.size() : 0; }

int fzb_multi_match_list_device(fzb_multi_matcher* mm, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match* dev_out, size_t capacity,
                                uint32_t* dev_count, void* stream) {
    if (!mm || !c || !dev_count || (!dev_out && capacity)) return fail(FZB_ERR_INVALID, "null argument");
    if (first > c->dev.n || count > c->dev.n - first) return fail(FZB_ERR_INVALID, "range outside the corpus");
    if ((u64)count + (u64)index_offset > 0xFFFFFFFFull)
        return fail(FZB_ERR_PANIC, "too many items in haystack, will overflow the u32 index: " + std::to_string((u64)count + index_offset) + " > 4294967295 (index offset: " + std::to_string(index_offset) + ")");
    hipStream_t st = (hipStream_t)stream;
    if (!mm->num_cus) {
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        mm->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int cus = mm->num_cus;
    const u32 cap32 = (u32)std::min<size_t>(capacity, 0xFFFFFFFFu);
    auto& ps = mm->patterns;
    if (count == 0) {
        HIPCHK(hipMemsetAsync(dev_count, 0, 8, st));
        return FZB_OK;
    }
    // CompiledPatterns::{Empty, Single, Multi} (src/matcher/mod.rs:178-190; a single NEGATED pattern is Multi)
    if (ps.empty()) {
        if (count > capacity) return fail(FZB_ERR_CAPACITY, "output buffer smaller than the haystack list (no pattern: every haystack matches)");
        fzb_launch_identity_records((fzb_match_rec*)dev_out, (u32)count, index_offset, dev_count, cus * 2, st);
        HIPCHK(hipGetLastError());
        return FZB_OK;
    }
    if (ps.size() == 1 && !ps[0].negated) return fzb_match_list_device(ps[0].m, c, first, count, index_offset, dev_out, capacity, dev_count, stream);
    if (mm->cap < count) {
        multi_free_buffers(mm);
        const size_t cap = count + count / 8 + 4096;
        HIPCHK(dev_alloc((void**)&mm->cand[0], (cap + 16) * sizeof(fzb_match_rec)));
        HIPCHK(dev_alloc((void**)&mm->cand[1], (cap + 16) * sizeof(fzb_match_rec)));
        HIPCHK(dev_alloc((void**)&mm->counts, 64));
        HIPCHK(dev_alloc((void**)&mm->items, cap * 4));
        HIPCHK(dev_alloc((void**)&mm->bitmap, (cap / 64 + 17) * 8));
        HIPCHK(dev_alloc((void**)&mm->tile_counts, ((cap + FZB_TILE - 1) / FZB_TILE + 2) * 4));
        mm->cap = cap;
    }
    // match_list_multi_into (src/matcher/multi.rs:84-152)
    size_t base = ps.size();
    for (size_t i = 0; i < ps.size(); i++)
        if (!ps[i].negated) { base = i; break; }
    int cur = 0;  // cand[cur] / counts[cur] = the candidates
    int rc;
    if (base != ps.size()) {
        rc = fzb_match_list_device(ps[base].m, c, first, count, index_offset, (fzb_match*)mm->cand[0], mm->cap, &mm->counts[4 * 0], stream);
        if (rc) return rc;
    } else {
        fzb_launch_identity_records(mm->cand[0], (u32)count, index_offset, &mm->counts[4 * 0], cus * 2, st);  // all patterns negated: every haystack is a candidate
    }
    for (size_t pi = 0; pi < ps.size(); pi++) {
        if (pi == base) continue;
        // (the reference skips the remaining patterns once no candidate is left; with the count on the device the kernels just find nothing to do)
        fzb_launch_records_to_items(mm->cand[cur], &mm->counts[4 * cur], index_offset, mm->items, cus * 2, st);
        const int oth = cur ^ 1;
        if (!ps[pi].negated) {
            rc = run_pipeline(ps[pi].m, c, first, count, index_offset, mm->items, &mm->counts[4 * cur], (fzb_match*)mm->cand[oth], mm->cap, &mm->counts[4 * oth], stream);
            if (rc) return rc;
            fzb_launch_join_add(mm->cand[oth], &mm->counts[4 * oth], mm->cand[cur], &mm->counts[4 * cur], cus * 2, st);
            cur = oth;
        } else {
            // the negated pattern's hits land in the other slot; k_flag_absent reads them, then k_compact_records (next in stream
            // order) overwrites that same slot with the candidates that were not hit
            fzb_match_rec* hits = mm->cand[oth];
            rc = run_pipeline(ps[pi].m, c, first, count, index_offset, mm->items, &mm->counts[4 * cur], (fzb_match*)hits, mm->cap, &mm->counts[4 * 2], stream);
            if (rc) return rc;
            fzb_launch_remove_hits(mm->cand[cur], &mm->counts[4 * cur], hits, &mm->counts[4 * 2], mm->bitmap, mm->tile_counts, mm->cand[oth], &mm->counts[4 * oth], cus * 2, st);
            cur = oth;
        }
    }
    fzb_launch_copy_records(mm->cand[cur], &mm->counts[4 * cur], (fzb_match_rec*)dev_out, cap32, dev_count, cus * 2, st);
    HIPCHK(hipGetLastError());
    return FZB_OK;
}

int fzb_multi_match_list(fzb_multi_matcher* mm, const fzb_corpus* c, fzb_match** out, size_t* out_len) {
    if (!mm || !c || !out || !out_len) return fail(FZB_ERR_INVALID, "null argument");
    const size_t count = c->dev.n;
    *out = nullptr;
    *out_len = 0;
    if (mm->out_cap < count || !mm->count_dev) {
        if (mm->out_dev) (void)hipFree(mm->out_dev);
        mm->out_dev = nullptr;
        mm->out_cap = 0;
        HIPCHK(dev_alloc((void**)&mm->out_dev, (count + 16) * sizeof(fzb_match_rec)));
        mm->out_cap = count;
        if (!mm->count_dev) HIPCHK(dev_alloc((void**)&mm->count_dev, 64));
    }
    int rc = fzb_multi_match_list_device(mm, c, 0, count, 0, (fzb_match*)mm->out_dev, mm->out_cap, mm->count_dev, nullptr);
    if (rc) return rc;
    // Matcher::match_list (src/matcher/mod.rs:212-222): reverse, then the stable radix sort unless there is no pattern at all
    const int sort = mm->config.sort;
    const bool reversed = sort == FZB_SORT_INDEX_DESC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC;
    const bool by_score = !mm->patterns.empty() && (sort == FZB_SORT_SCORE_THEN_INDEX_ASC || sort == FZB_SORT_SCORE_THEN_INDEX_DESC);
    if ((reversed || by_score) && count) {
        if (by_score && mm->sort_cap < count) {
            if (mm->sort_tmp) (void)hipFree(mm->sort_tmp);
            if (mm->sort_hist) (void)hipFree(mm->sort_hist);
            mm->sort_tmp = nullptr; mm->sort_hist = nullptr; mm->sort_cap = 0;
            HIPCHK(dev_alloc((void**)&mm->sort_tmp, (count + 16) * sizeof(fzb_match_rec)));
            const size_t hist_words = (size_t)2 * 256 * (count / 2048 + 2);  // (+ the digit totals and the phase word: kernels_sort.hip)
            HIPCHK(dev_alloc((void**)&mm->sort_hist, (hist_words + 1024) * 4));
            HIPCHK(hipMemset(mm->sort_hist + hist_words, 0, 1024 * 4));
            mm->sort_cap = count;
        }
        fzb_launch_sort(mm->out_dev, mm->sort_tmp, mm->count_dev, mm->sort_hist, (u32)(mm->sort_cap / 2048 + 2), reversed, by_score, mm->num_cus * 2, nullptr);
        HIPCHK(hipGetLastError());
    }
    return fetch_records(mm->fetch, mm->out_dev, mm->count_dev, mm->out_cap, out, out_len);
}

// `Matcher::match_list_indices` over CompiledPatterns (mod.rs:234-275): Empty / Single as above; Multi = match_one_indices_multi
// (multi.rs:56-82) for every haystack of the list.  Like match_list_multi_into, every pattern only sees the haystacks that are
// still alive, and each pattern's positions come from the traced scorer on the device; the per-haystack union is host work.
static int multi_match_list_indices_impl(fzb_multi_matcher* mm, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, int sort, uint32_t index_offset,
                                         fzb_match_indices** out, size_t* out_len, uint32_t** out_positions) {
    if (!mm || !c || !out || !out_len || !out_positions || (!selection && n_selection)) return fail(FZB_ERR_INVALID, "null argument");
    *out = nullptr;
    *out_len = 0;
    *out_positions = nullptr;
    size_t count = 0;
    int rc = check_selection(c, selection, n_selection, count, index_offset);
    if (rc) return rc;
    std::vector<fzb_match_indices> recs;
    std::vector<u32> positions;
    if (mm->patterns.empty()) {  // CompiledPatterns::Empty
        recs.resize(count);
        for (size_t i = 0; i < count; i++) recs[i] = fzb_match_indices{(uint32_t)i, 0, 0, 0, 0, 0};
        return finish_indices(recs, positions, sort, false, out, out_len, out_positions, index_offset);
    }
    if (mm->patterns.size() == 1 && !mm->patterns[0].negated) {  // CompiledPatterns::Single
        rc = indices_in_list_order(mm->patterns[0].m, c, selection, count, recs, positions);
        if (rc) return rc;
        return finish_indices(recs, positions, sort, true, out, out_len, out_positions, index_offset);
    }
    // alive[k] = (position in the caller's list, corpus index); combined score / exact / positions per alive haystack
    std::vector<u32> alive_pos(count), alive_idx(count);
    for (size_t i = 0; i < count; i++) {
        alive_pos[i] = (u32)i;
        alive_idx[i] = selection ? selection[i] : (u32)i;
    }
    std::vector<u32> score(count, 0);
    std::vector<u8> exact(count, 0);
    std::vector<std::vector<u32>> found(count);
    std::vector<fzb_match_indices> hits;
    std::vector<u32> hit_positions;
    for (auto& p : mm->patterns) {
        if (alive_idx.empty()) break;
        hit_positions.clear();
        rc = indices_in_list_order(p.m, c, alive_idx.data(), alive_idx.size(), hits, hit_positions);
        if (rc) return rc;
        std::vector<u32> next_pos, next_idx;
        if (p.negated) {  // a negated pattern that matches drops the haystack
            size_t h = 0;
            for (size_t k = 0; k < alive_idx.size(); k++) {
                if (h < hits.size() && hits[h].index == k) { h++; continue; }
                next_pos.push_back(alive_pos[k]);
                next_idx.push_back(alive_idx[k]);
            }
        } else {  // every other pattern must match; scores add with saturation, exact flags OR, positions accumulate
            for (const fzb_match_indices& hit : hits) {
                const u32 at = alive_pos[hit.index];
                score[at] = std::min<u32>(0xFFFFu, score[at] + hit.score);
                exact[at] |= hit.exact;
                found[at].insert(found[at].end(), hit_positions.begin() + hit.positions_begin, hit_positions.begin() + hit.positions_begin + hit.positions_len);
                next_pos.push_back(at);
                next_idx.push_back(alive_idx[hit.index]);
            }
        }
        alive_pos.swap(next_pos);
        alive_idx.swap(next_idx);
    }
    recs.reserve(alive_pos.size());
    for (u32 at : alive_pos) {  // "Indices are reported in reverse order, and patterns may share matched chars" (multi.rs:75-77)
        std::vector<u32>& f = found[at];
        std::sort(f.begin(), f.end(), [](u32 a, u32 b) { return a > b; });
        f.erase(std::unique(f.begin(), f.end()), f.end());
        recs.push_back(fzb_match_indices{at, (uint16_t)score[at], exact[at], 0, (uint32_t)positions.size(), (uint32_t)f.size()});
        positions.insert(positions.end(), f.begin(), f.end());
    }
    return finish_indices(recs, positions, sort, true, out, out_len, out_positions, index_offset);
}

int fzb_multi_match_list_indices(fzb_multi_matcher* mm, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, fzb_match_indices** out, size_t* out_len,
                                 uint32_t** out_positions) {
    return multi_match_list_indices_impl(mm, c, selection, n_selection, mm ? mm->config.sort : 0, 0, out, out_len, out_positions);
}

int fzb_multi_match_list_indices_into(fzb_multi_matcher* mm, const fzb_corpus* c, const uint32_t* selection, size_t n_selection, uint32_t index_offset,
                                      fzb_match_indices** out, size_t* out_len, uint32_t** out_positions) {
    return multi_match_list_indices_impl(mm, c, selection, n_selection, FZB_SORT_INDEX_ASC, index_offset, out, out_len, out_positions);
}

// `Matcher::match_list_into` over CompiledPatterns (src/matcher/mod.rs:373-392): index order, host result
int fzb_multi_match_list_into(fzb_multi_matcher* mm, const fzb_corpus* c, size_t first, size_t count, uint32_t index_offset, fzb_match** out, size_t* out_len) {
    if (!mm || !c || !out || !out_len) return fail(FZB_ERR_INVALID, "null argument");
    if (first > c->dev.n || count > c->dev.n - first) return fail(FZB_ERR_INVALID, "range outside the corpus");
    *out = nullptr;
    *out_len = 0;
    if (mm->out_cap < count || !mm->count_dev) {
        if (mm->out_dev) (void)hipFree(mm->out_dev);
        mm->out_dev = nullptr;
        mm->out_cap = 0;
        HIPCHK(dev_alloc((void**)&mm->out_dev, (count + 16) * sizeof(fzb_match_rec)));
        mm->out_cap = count;
        if (!mm->count_dev) HIPCHK(dev_alloc((void**)&mm->count_dev, 64));
    }
    int rc = fzb_multi_match_list_device(mm, c, first, count, index_offset, (fzb_match*)mm->out_dev, mm->out_cap, mm->count_dev, nullptr);
    if (rc) return rc;
    return fetch_records(mm->fetch, mm->out_dev, mm->count_dev, mm->out_cap, out, out_len);
}

void fzb_matches_free(fzb_match* p) {
    if (p && !pinned_pool().put(p)) free(p);
}

static bool merge_less(int order, const fzb_match& l, const fzb_match& r) {  // src/k_merge.rs:14-53
    switch (order) {
        case FZB_SORT_SCORE_THEN_INDEX_ASC: return l.score > r.score || (l.score == r.score && l.index < r.index);
        case FZB_SORT_SCORE_THEN_INDEX_DESC: return l.score > r.score || (l.score == r.score && l.index > r.index);
        case FZB_SORT_INDEX_ASC: return l.index < r.index;
        default: return l.index > r.index;
    }
}

int fzb_k_merge_matches(int32_t sort, const fzb_match* runs, const size_t* run_lens, size_t nruns, fzb_match* out) {
    if (sort < 0 || sort > 3 || (nruns && (!run_lens))) return fail(FZB_ERR_INVALID, "bad argument");
    size_t total = 0;
    for (size_t k = 0; k < nruns; k++) total += run_lens[k];
    if (total && (!runs || !out)) return fail(FZB_ERR_INVALID, "null argument");
    std::vector<const fzb_match*> ptrs(nruns);
    size_t off = 0;
    for (size_t k = 0; k < nruns; k++) { ptrs[k] = runs + off; off += run_lens[k]; }
    return fzb_k_merge_runs(sort, ptrs.data(), run_lens, nruns, out);
}

}  // extern "C"
int fzb_k_merge_runs(int32_t sort, const fzb_match* const* runs, const size_t* run_lens, size_t nruns, fzb_match* out) {
    if (sort < 0 || sort > 3 || (nruns && (!run_lens || !runs))) return fail(FZB_ERR_INVALID, "bad argument");
    // tournament over run heads (any correct k-way merge of runs sorted under a TOTAL order yields the same sequence)
    std::vector<size_t> pos(nruns, 0);
    std::vector<size_t> heap;
    auto less_run = [&](size_t a, size_t b) { return merge_less(sort, runs[a][pos[a]], runs[b][pos[b]]); };
    auto cmp = [&](size_t a, size_t b) { return less_run(b, a); };  // std heap is a max-heap
    for (size_t k = 0; k < nruns; k++)
        if (run_lens[k]) heap.push_back(k);
    if (!heap.empty() && !out) return fail(FZB_ERR_INVALID, "null argument");
    std::make_heap(heap.begin(), heap.end(), cmp);
    size_t o = 0;
    while (!heap.empty()) {
        if (heap.size() == 1) {  // one run left: the rest is a copy
            const size_t k = heap[0];
            memcpy(out + o, runs[k] + pos[k], (run_lens[k] - pos[k]) * sizeof(fzb_match));
            break;
        }
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const size_t k = heap.back();
        out[o++] = runs[k][pos[k]];
        if (++pos[k] < run_lens[k]) std::push_heap(heap.begin(), heap.end(), cmp);
        else heap.pop_back();
    }
    return FZB_OK;
}
extern "C" {

int fzb_set_profiling(fzb_matcher* m, int enabled) {
    if (!m) return fail(FZB_ERR_INVALID, "null argument");
    m->profiling = enabled != 0;
    m->prof_calls = 0;
    return FZB_OK;
}

// Averages over the calls made since fzb_set_profiling(m, 1) (at most the last 32 = PROF_SLOTS):
// out_ms[0] = filter kernel, [1] = whole pipeline, [2] = calls averaged, [3] = 1 if a filter kernel ran
int fzb_last_timings(fzb_matcher* m, float out_ms[4]) {
    if (!m || !out_ms) return fail(FZB_ERR_INVALID, "null argument");
    if (!m->profiling || m->prof_calls == 0) return fail(FZB_ERR_INVALID, "profiling not enabled or no call recorded");
    const u64 n = std::min<u64>(m->prof_calls, fzb_matcher::PROF_SLOTS);
    double f = 0, t = 0;
    u64 n_filter = 0;
    for (u64 i = 0; i < n; i++) {
        const int slot = (int)((m->prof_calls - 1 - i) % fzb_matcher::PROF_SLOTS);
        hipEvent_t* e = m->evring[slot];
        HIPCHK(hipEventSynchronize(e[1]));
        float b = 0;
        HIPCHK(hipEventElapsedTime(&b, e[0], e[1]));
        t += b;
        if (m->ev_filter[slot]) {
            float a = 0;
            HIPCHK(hipEventElapsedTime(&a, e[2], e[3]));
            f += a;
            n_filter++;
        }
    }
    out_ms[0] = n_filter ? (float)(f / n_filter) : 0.0f;
    out_ms[1] = (float)(t / n);
    out_ms[2] = (float)n;
    out_ms[3] = (float)n_filter;
    return FZB_OK;
}

// Per-stage averages over the same calls: out_ms[0] = streaming filter kernel, [1] = compaction (+ the lane-exact prefilter and its
// compaction when the configuration has them), [2] = the scorers, [3] = whole pipeline, [4] = calls averaged, [5] = 1 if a filter ran
int fzb_last_stage_timings(fzb_matcher* m, float out_ms[6]) {
    if (!m || !out_ms) return fail(FZB_ERR_INVALID, "null argument");
    if (!m->profiling || m->prof_calls == 0) return fail(FZB_ERR_INVALID, "profiling not enabled or no call recorded");
    const u64 n = std::min<u64>(m->prof_calls, fzb_matcher::PROF_SLOTS);
    // calls with a streaming filter kernel and calls without one (item lists, no prefilter) are averaged separately: the filter
    // figure is the mean over the calls that had one, "compaction" the mean of what lies between filter and scorers in each kind
    double f = 0, mid = 0, sc = 0, t = 0;
    u64 n_filter = 0;
    for (u64 i = 0; i < n; i++) {
        const int slot = (int)((m->prof_calls - 1 - i) % fzb_matcher::PROF_SLOTS);
        hipEvent_t* e = m->evring[slot];
        HIPCHK(hipEventSynchronize(e[1]));
        float a = 0;
        HIPCHK(hipEventElapsedTime(&a, e[0], e[1]));
        t += a;
        HIPCHK(hipEventElapsedTime(&a, e[4], e[1]));
        sc += a;
        if (m->ev_filter[slot]) {
            n_filter++;
            HIPCHK(hipEventElapsedTime(&a, e[2], e[3]));
            f += a;
            HIPCHK(hipEventElapsedTime(&a, e[3], e[4]));
            mid += a;
        } else {
            HIPCHK(hipEventElapsedTime(&a, e[0], e[4]));
            mid += a;
        }
    }
    out_ms[0] = n_filter ? (float)(f / n_filter) : 0.0f;
    out_ms[1] = (float)(mid / n);
    out_ms[2] = (float)(sc / n);
    out_ms[3] = (float)(t / n);
    out_ms[4] = (float)n;
    out_ms[5] = (float)n_filter;  // calls (of the averaged ones) that ran a streaming filter kernel; 0 = none
    return FZB_OK;
}

// Test hook (host only, no GPU): runs the unicode 0-typo prefilter DFA the streaming filter would run over one haystack.
// Returns 1 / 0 = accept / reject, -1 if this matcher has no such DFA (not the unicode path, typos, or more than 255 states).
int fzb_debug_unicode_dfa_accepts(const fzb_matcher* m, const uint8_t* bytes, size_t len) {
    if (!m || m->uni_dfa_states == 0 || (!bytes && len)) return -1;
    u32 st = 0;
    for (size_t i = 0; i < len; i++) st = m->uni_dfa[(size_t)st * 256 + bytes[i]];
    return st == (u32)m->uni_dfa_states - 1 ? 1 : 0;
}

// Test hook (host only): the LCS automaton of a typo configuration run over one haystack.  1 / 0 = the streaming filter accepts / rejects
// (LCS >= rows - max_typos), -1 if the matcher has no such automaton; *out_states (optional) = its number of states.
int fzb_debug_lcs_dfa_accepts(const fzb_matcher* m, const uint8_t* bytes, size_t len, int32_t* out_states) {
    if (out_states) *out_states = m ? m->lcs_states : 0;
    if (!m || m->lcs_states == 0 || (!bytes && len)) return -1;
    u32 st = 0;
    for (size_t i = 0; i < len; i++) st = m->lcs_dfa[(size_t)st * 256 + bytes[i]];
    return st >= (u32)m->lcs_acc_lo ? 1 : 0;
}

// Test hook (host only): the class-composite form of the matcher's streaming automaton run over one haystack, G bytes per step (the
// tail padded with a byte of the "matches nothing" class, as the kernel sees zero fill): 1 / 0 = accepts / rejects, -1 if the matcher has none.
// out_kg (optional): [0] = K, [1] = G.
int fzb_debug_cdfa_state(const fzb_matcher* m, const uint8_t* bytes, size_t len, int32_t* out_kg) {
    if (out_kg) { out_kg[0] = m ? m->cdfa_K : 0; out_kg[1] = m ? m->cdfa_G : 0; }
    if (!m || m->cdfa.empty() || (!bytes && len)) return -1;
    const size_t K = (size_t)m->cdfa_K;
    size_t KG = 1;
    for (int i = 0; i < m->cdfa_G; i++) KG *= K;
    u32 st = 0;
    for (size_t i = 0; i < len; i += (size_t)m->cdfa_G) {
        size_t off = 0, mul = 1;
        for (int j = 0; j < m->cdfa_G; j++, mul *= K) off += mul * m->cdfa[i + j < len ? bytes[i + j] : (u8)m->lc.dead_byte];
        st = m->cdfa[256 + (size_t)st * KG + off];
    }
    const u32 acc = m->cdfa_src == 1 ? (u32)m->rows : m->cdfa_src == 2 ? (u32)m->uni_dfa_states - 1 : (u32)m->lcs_acc_lo;
    return st >= acc ? 1 : 0;
}

int fzb_last_counters(fzb_matcher* m, uint32_t out[4]) {
    if (!m || !out) return fail(FZB_ERR_INVALID, "null argument");
    if (m->ws.counters) {
        u32 all[16];
        HIPCHK(hipMemcpy(all, m->ws.counters, sizeof(all), hipMemcpyDeviceToHost));
        m->last_counters[0] = all[0];
        m->last_counters[1] = all[1];
        // (unicode: windows of 65..1024 bytes that the thread-per-haystack scorer handed on to the queue's back are in BOTH counts - all[7] claims,
        // at most 4096 granted: they stay "multi-chunk" here, and the greedy fallback's count is the back minus them)
        const u32 fwd = std::min<u32>(all[7], 4096u);
        m->last_counters[2] = all[4] >= fwd && m->nd.unicode ? all[4] - fwd : all[4];
        m->last_counters[3] = all[3] + all[12] + all[13] + all[14] + all[15];  // multi-chunk windows: the queue, or the four tail-class lists
    }
    memcpy(out, m->last_counters, 16);
    return FZB_OK;
}

}  // extern "C"
