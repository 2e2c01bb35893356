// ORACLE-side brute force (test infrastructure): on inputs that fit ONE prefilter chunk (len <= lanes), does the reference's multi-path typo
// prefilter ever deviate from the LCS criterion?  g++ -O2 -std=c++17 -Ioracle -o sc oracle/single_chunk_check.cpp && ./sc 6000000 1
// Round 2: 7 seeds x 6e6 needles x 3 widths = 1.26e8 cases, 1.45e7 of them marginal (LCS + k == n): 0 deviations at 16 / 32 / 64 lanes
// (multi-chunk inputs deviate about once in 1e5: oracle/selfcheck.cpp).  The argument why is in DESIGN.md section 3e.
#include "frizbee_oracle.hpp"
#include <cstdio>
#include <random>
using namespace fzo;
static size_t lcs(const std::vector<std::pair<u8,u8>>& n, const u8* h, size_t hl) {
    std::vector<size_t> prev(hl + 1, 0), cur(hl + 1, 0);
    for (auto& c : n) { cur[0] = 0; for (size_t j = 0; j < hl; j++) cur[j + 1] = (h[j] == c.first || h[j] == c.second) ? prev[j] + 1 : std::max(prev[j + 1], cur[j]); std::swap(prev, cur); }
    return prev[hl];
}
template <int L> static Window run(const Prefilter<L>& p, const u8* h, size_t hl, int k) {
    if (k == 1) return p.match_haystack_1_typo(h, hl);
    if (k == 2) return p.match_haystack_2_typos(h, hl);
    return p.match_haystack_many_typos(h, hl, k);
}
int main(int argc, char** argv) {
    size_t iters = strtoull(argv[1], 0, 10); unsigned seed = atoi(argv[2]);
    std::mt19937_64 rng(seed);
    const char* alpha = "abcABC_-/ 01xyz"; size_t nalpha = strlen(alpha);
    size_t dev[3] = {0,0,0}, marg = 0, acc = 0;
    for (size_t it = 0; it < iters; it++) {
        size_t asz = 2 + rng() % (nalpha - 1);
        size_t nl = 2 + rng() % 11;
        int k = 1 + rng() % 4; if ((size_t)k >= nl) continue;
        bool cs = rng() % 3 == 0;
        std::string needle; for (size_t i = 0; i < nl; i++) needle += alpha[rng() % asz];
        Prefilter<16> p16(needle, cs); Prefilter<32> p32(needle, cs); Prefilter<64> p64(needle, cs);
        for (int w = 0; w < 3; w++) {
            size_t lanes = 16u << w, hl = 1 + rng() % lanes;
            if (rng() % 3 == 0) hl = lanes - rng() % 3;
            std::string hay; for (size_t i = 0; i < hl; i++) hay += alpha[rng() % asz];
            size_t l = lcs(p16.needle_ascii, (const u8*)hay.data(), hl);
            bool want = l + k >= nl;
            bool got = w == 0 ? run(p16, (const u8*)hay.data(), hl, k).matched : w == 1 ? run(p32, (const u8*)hay.data(), hl, k).matched : run(p64, (const u8*)hay.data(), hl, k).matched;
            acc += want; marg += (l + k == nl);
            if (got != want) { if (dev[w]++ < 3) printf("DEVIATION lanes=%zu needle=%s hay=%s k=%d cs=%d want=%d got=%d\n", lanes, needle.c_str(), hay.c_str(), k, cs, want, got); }
        }
    }
    printf("iters=%zu accepted=%zu marginal=%zu deviations L16=%zu L32=%zu L64=%zu\n", iters, acc, marg, dev[0], dev[1], dev[2]);
}
