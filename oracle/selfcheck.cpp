// ORACLE self-consistency fuzz (test infrastructure).  Checks the two properties the reference's own
// randomized tests assert (src/prefilter/mod.rs:894-908, 1013-1084, 552-586):
//   (1) ASCII prefilter accept  =>  LCS(needle, haystack) + max_typos >= len(needle)  (case-folded bytes).
//       The reference asserts "<=>" on 256 random cases; this fuzz shows the converse FAILS rarely
//       (LANES=32: ~1e-5 of cases are rejected although LCS accepts - confirmed with an independent
//       Python transcription), i.e. the accept decision is lane-width dependent.  Hence the GPU filter
//       stage may only use LCS as a conservative superset and must re-run the literal algorithm.
//   (2) the ASCII window is identical at LANES 16/32/64 whenever all widths accept
// and reports how often Smith-Waterman scores differ between lane widths (SURVEY finding 1).
#include "frizbee_oracle.hpp"
#include <cstdio>
#include <random>
using namespace fzo;

static size_t lcs(const std::vector<std::pair<u8,u8>>& n, const u8* h, size_t hl) {
    std::vector<size_t> prev(hl + 1, 0), cur(hl + 1, 0);
    for (auto& c : n) {
        cur[0] = 0;
        for (size_t j = 0; j < hl; j++) cur[j + 1] = (h[j] == c.first || h[j] == c.second) ? prev[j] + 1 : std::max(prev[j + 1], cur[j]);
        std::swap(prev, cur);
    }
    return prev[hl];
}
template <int L> static Window run(const Prefilter<L>& p, const u8* h, size_t hl, int k) {
    if (k == 0) return p.match_haystack(h, hl);
    if (k == 1) return p.match_haystack_1_typo(h, hl);
    if (k == 2) return p.match_haystack_2_typos(h, hl);
    return p.match_haystack_many_typos(h, hl, k);
}
int main(int argc, char** argv) {
    size_t iters = argc > 1 ? strtoull(argv[1], 0, 10) : 200000;
    std::mt19937_64 rng(12345);
    const char* alpha = "abcABC_-/ 01xyz";
    size_t nalpha = strlen(alpha);
    size_t superset_violation = 0, lcs16 = 0, lcs32 = 0, lcs64 = 0, bad_lcs = 0, bad_win = 0, sw_diff = 0, sw_n = 0, matched_n = 0;
    for (size_t it = 0; it < iters; it++) {
        size_t asz = 2 + rng() % (nalpha - 1);
        size_t nl = 1 + rng() % 10, hl = rng() % 150;
        if (rng() % 4 == 0) hl = (size_t[]){0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129}[rng() % 14];
        std::string needle, hay;
        for (size_t i = 0; i < nl; i++) needle += alpha[rng() % asz];
        for (size_t i = 0; i < hl; i++) hay += alpha[rng() % asz];
        bool cs = rng() % 3 == 0;
        int k = rng() % 5;
        Prefilter<16> p16(needle, cs); Prefilter<32> p32(needle, cs); Prefilter<64> p64(needle, cs);
        Window w16 = run(p16, (const u8*)hay.data(), hl, k), w32 = run(p32, (const u8*)hay.data(), hl, k), w64 = run(p64, (const u8*)hay.data(), hl, k);
        bool want = (size_t)k >= nl ? true : lcs(p16.needle_ascii, (const u8*)hay.data(), hl) + k >= nl;
        if ((w16.matched && !want) || (w32.matched && !want) || (w64.matched && !want)) {
            superset_violation++;
            printf("SUPERSET VIOLATION needle=%s hay=%s k=%d\n", needle.c_str(), hay.c_str(), k);
        }
        if (w16.matched != want || w32.matched != want || w64.matched != want) {
            if (w16.matched != want) lcs16++;
            if (w32.matched != want) lcs32++;
            if (w64.matched != want) lcs64++;
            if (bad_lcs++ < 5) printf("LCS MISMATCH needle=%s hay=%s k=%d cs=%d want=%d got=%d/%d/%d\n", needle.c_str(), hay.c_str(), k, cs, want, w16.matched, w32.matched, w64.matched);
        } else if (want) {
            matched_n++;
            if (w16.start != w64.start || w16.end != w64.end || w32.start != w64.start || w32.end != w64.end) {
                if (bad_win++ < 5) printf("WINDOW MISMATCH needle=%s hay=%s k=%d: (%zu,%zu) (%zu,%zu) (%zu,%zu)\n", needle.c_str(), hay.c_str(), k, w16.start, w16.end, w32.start, w32.end, w64.start, w64.end);
            }
            if (it % 8 == 0) {
                Scoring sc;
                SmithWaterman<64, u8> s64(needle, sc, cs); SmithWaterman<16, u8> s16(needle, sc, cs); SmithWaterman<8, u16> s8(needle, sc, cs);
                u16 a = s64.score_haystack((const u8*)hay.data(), hl, true), b = s16.score_haystack((const u8*)hay.data(), hl, true), c = s8.score_haystack((const u8*)hay.data(), hl, true);
                sw_n++;
                if (a != b || a != c) { if (sw_diff++ < 3) printf("SW lanes differ needle=%s hay=%s: 64u8=%u 16u8=%u 8u16=%u\n", needle.c_str(), hay.c_str(), a, b, c); }
            }
        }
    }
    printf("iters=%zu matched=%zu lcs_false_negatives=%zu (L16=%zu L32=%zu L64=%zu) superset_violations=%zu window_mismatch=%zu sw_compared=%zu sw_lane_dependent=%zu\n",
           iters, matched_n, bad_lcs, lcs16, lcs32, lcs64, superset_violation, bad_win, sw_n, sw_diff);
    return (superset_violation || bad_win) ? 1 : 0;
}
