// The reference's own matcher tests, transcribed onto the C++ host side (include/frizbee_hip.hpp): they read like
// src/matcher/mod.rs:531-654, src/matcher/multi.rs:160-416 and src/literal/mod.rs:55-91 do.
//   ./test_facade        host-only part: defaults, query parsing, panics, "no GPU -> loud error"
//   ./test_facade gpu    + the matching tests (needs an MI355X)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "frizbee_hip.hpp"

using namespace frizbee;

static int failures = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                  \
        }                                                                \
    } while (0)

template <typename F>
static std::string panic_text(F f) {
    try {
        f();
    } catch (const Panic& p) {
        return p.what();
    }
    return "";
}
static std::vector<uint32_t> indices(const std::vector<Match>& ms) {
    std::vector<uint32_t> r;
    for (const Match& m : ms) r.push_back(m.index);
    return r;
}
static Matcher multi(const char* query, const Config& config) { return Matcher::from_patterns(Pattern::parse_query(query), config); }

static void host_side() {
    // Config::default() / Scoring::default() (src/lib.rs:260-271, src/const.rs:1-10)
    const Config c;
    CHECK(c.max_typos_ == std::optional<uint16_t>(0) && c.casing_ == CaseMatching::Smart && c.unicode_ == UnicodeMatching::Smart);
    CHECK(c.sort_ == SortStrategy::ScoreThenIndexAsc && c.matching_ == Matching::Fuzzy && c.scoring_ == Scoring{});
    CHECK(c.scoring_.match_score == 12 && c.scoring_.exact_match_bonus == 8 && c.scoring_.delimiter_bonus == 4);
    // Pattern::parse_query (src/pattern.rs:345-381)
    auto ps = Pattern::parse_query("foo !^bar");
    CHECK(ps.size() == 2 && ps[0].needle == "foo" && !ps[0].negated && !ps[0].config.matching);
    CHECK(ps[1].needle == "bar" && ps[1].negated && ps[1].config.matching == std::optional<Matching>(Matching::Prefix));
    CHECK(Pattern::parse_query("  foo \t bar  ").size() == 2);
    ps = Pattern::parse_query("foo\\ bar baz");
    CHECK(ps.size() == 2 && ps[0].needle == "foo bar" && ps[1].needle == "baz");
    CHECK(Pattern::parse_query("! ^$ '").empty() && Pattern::parse_query("").empty());
    ps = Pattern::parse_query("!foo 'x ^a$ b$");
    CHECK(ps[0].config.matching == std::optional<Matching>(Matching::Substring) && ps[1].config.matching == std::optional<Matching>(Matching::Substring));
    CHECK(ps[2].config.matching == std::optional<Matching>(Matching::Exact) && ps[3].config.matching == std::optional<Matching>(Matching::Suffix));
    // panics carry the reference's text: guard_against_score_overflow (src/matcher/algo.rs:372-380), threads == 0 (parallel.rs:24)
    Scoring huge;
    huge.capitalization_bonus = 60000;
    huge.matching_case_bonus = 40000;
    CHECK(panic_text([&] { Matcher m("f", Config().scoring(huge)); }).find("needle too long and could overflow the u16 score") == 0);
    CHECK(panic_text([&] { Matcher m(std::string(5000, 'a').c_str(), Config()); }).find("needle too long") == 0);  // tests/api_properties.rs:610-616
    CHECK(panic_text([&] { Matcher("a").match_list_parallel(std::vector<std::string>{"a"}, 0); }) == "threads must be positive");
}

static void no_gpu_fails_loudly() {
    try {
        Matcher("abc").match_list(std::vector<std::string>{"abc"});
        CHECK(!"scoring without a GPU must not succeed: there is no CPU fallback");
    } catch (const Error&) {
    }
}

static void gpu_side() {
    const std::vector<std::string> haystack = {"deadbeef", "deadbf", "deadbeefg", "deadbe"};
    {  // test_basic (src/matcher/mod.rs:531-546)
        auto matches = Matcher("deadbe", Config().max_typos(std::nullopt)).match_list(haystack);
        CHECK(matches.size() == 4 && matches[0].index == 3 && matches[1].index == 0 && matches[2].index == 2 && matches[3].index == 1);
    }
    // test_no_typos (:549-556)
    CHECK(Matcher("deadbe", Config().max_typos(0)).match_list(haystack).size() == 3);
    {  // test_exact_match (:559-571)
        auto matches = Matcher("deadbe").match_list(haystack);
        size_t exact = 0;
        for (const Match& m : matches)
            if (m.exact) { exact++; CHECK(m.index == 3 && haystack[m.index] == "deadbe"); }
        CHECK(exact == 1);
    }
    {  // test_small_needle (:596-602)
        auto matches = Matcher("1", Config().max_typos(2)).match_list(std::vector<std::string>{"1"});
        CHECK(matches.size() == 1 && matches[0].index == 0 && matches[0].exact);
    }
    {  // case modes (:618-654)
        const std::vector<std::string> hs = {"foo", "FOO", "fOo", "xxfooxx"};
        const Config ia = Config().sort(SortStrategy::IndexAsc);
        CHECK((indices(Matcher("foo", ia).match_list(hs)) == std::vector<uint32_t>{0, 1, 2, 3}));
        CHECK((indices(Matcher("foo", ia.casing(CaseMatching::Respect)).match_list(hs)) == std::vector<uint32_t>{0, 3}));
        CHECK((indices(Matcher("FoO", ia).match_list(std::vector<std::string>{"foo", "FOO", "FoO", "xxFoOxx"})) == std::vector<uint32_t>{2, 3}));
    }
    {  // README usage example (score hand-derived in SURVEY.md: 53)
        auto matches = Matcher("fBr").match_list(std::vector<std::string>{"fooBar", "foo_bar", "barfoo", "prelude", "println!"});
        CHECK(matches.size() == 1 && (matches[0] == Match{0, 53, false}));
    }
    {  // match_list_indices: the scorer's known answers (src/smith_waterman/mod.rs:323-325, 454-456), src/matcher/mod.rs:605-616, 724-735
        const Config none = Config().max_typos(std::nullopt);
        CHECK((Matcher("abc", none).match_list_indices(std::vector<std::string>{"xabcabc"})[0].indices == std::vector<uint32_t>{3, 2, 1}));
        CHECK((Matcher("ab", none).match_list_indices(std::vector<std::string>{"abab"})[0].indices == std::vector<uint32_t>{1, 0}));
        CHECK((Matcher("a\xc3\xa9", none).match_list_indices(std::vector<std::string>{"a\xc3\xa9"})[0].indices == std::vector<uint32_t>{2, 1, 0}));
        auto ix = Matcher("\xc3\xa9", Config().unicode(UnicodeMatching::Ignore)).match_list_indices(std::vector<std::string>{"xx\xc3\xa9"});
        CHECK(ix.size() == 1);
        if (ix.size() == 1) {
            std::sort(ix[0].indices.begin(), ix[0].indices.end());
            CHECK((ix[0].indices == std::vector<uint32_t>{2, 3}));
        }
        auto all = Matcher("").match_list_indices(std::vector<std::string>{"foo", "bar"});
        CHECK(all.size() == 2 && all[0].index == 0 && all[1].index == 1 && all[0].indices.empty());
        // multi_pattern_overlapping_indices_deduped (src/matcher/multi.rs:277-282)
        auto mi = multi("foo fo", Config()).match_list_indices(std::vector<std::string>{"foo"});
        CHECK(mi.size() == 1 && (mi[0].indices == std::vector<uint32_t>{2, 1, 0}));
        // positions for the top of a match_list result over a resident corpus
        const Corpus corpus(haystack);
        Matcher m("deadbe");
        auto top = m.match_list(corpus);
        std::vector<uint32_t> sel;
        for (const Match& t : top) sel.push_back(t.index);
        auto pos = m.match_list_indices(corpus, sel);
        CHECK(pos.size() == top.size());
        for (size_t i = 0; i < pos.size() && i < top.size(); i++)
            CHECK(pos[i].index == i && pos[i].score == top[i].score && pos[i].exact == top[i].exact && (pos[i].indices == std::vector<uint32_t>{5, 4, 3, 2, 1, 0}));
    }
    {  // match_iter / match_one / FuzzyMatchExt (src/matcher/mod.rs:655-734, src/matcher/iter.rs:158-222)
        const std::vector<std::string> hs = {"deadbeef", "deadbf", "deadbeefg", "deadbe", "no-match", "DeAdBe", "\xc3\xa9\xeb\x8b\xa4\xf0\x9f\x98\x80" "dead__be"};
        for (const char* needle : {"deadbe", "\xc3\xa9\xeb\x8b\xa4\xf0\x9f\x98\x80"}) {
            for (int t = -1; t <= 3; t++) {
                const Config cfg = Config().max_typos(t < 0 ? std::nullopt : std::optional<uint16_t>((uint16_t)t)).sort(SortStrategy::IndexAsc);
                Matcher m(needle, cfg);
                auto from_list = m.match_list(hs);
                CHECK(m.match_iter(hs) == from_list);
                CHECK(fuzzy_match(hs, needle, cfg) == from_list);
                CHECK(Matcher(needle, cfg.sort(SortStrategy::ScoreThenIndexAsc)).match_iter(hs) == from_list);  // match_iter never sorts
                auto ix_list = m.match_list_indices(hs);
                CHECK(m.match_iter_indices(hs) == ix_list);
                CHECK(fuzzy_match_indices(hs, needle, cfg) == ix_list);
                for (size_t i = 0; i < hs.size(); i++) {
                    auto one = m.match_one(hs[i], (uint32_t)i);
                    bool found = false;
                    for (const Match& w : from_list)
                        if (w.index == i) { found = true; CHECK(one && *one == w); }
                    if (!found) CHECK(!one);
                }
            }
        }
        auto all = Matcher("").match_iter(std::vector<std::string>{"foo", "bar"});
        CHECK(all.size() == 2 && all[0].index == 0 && all[1].index == 1);
        auto all_ix = fuzzy_match_indices(std::vector<std::string>{"foo", "bar"}, "");
        CHECK(all_ix.size() == 2 && all_ix[0].index == 0 && all_ix[1].index == 1 && all_ix[1].indices.empty());
        auto mone = multi("dead !bf", Config()).match_one("deadbeef", 7);
        CHECK(mone && mone->index == 7);
        CHECK(!multi("dead !bf", Config()).match_one("deadbf", 7));
    }
    {  // literal modes (src/literal/mod.rs:55-91)
        const Config ia = Config().sort(SortStrategy::IndexAsc);
        CHECK((indices(Matcher("foo", ia.matching(Matching::Exact)).match_list(std::vector<std::string>{"foo", "foobar", "xfoo", "FOO"})) == std::vector<uint32_t>{0, 3}));
        const std::vector<std::string> hs = {"foobar", "barfoo", "foo", "xfoobar"};
        CHECK((indices(Matcher("foo", ia.matching(Matching::Prefix)).match_list(hs)) == std::vector<uint32_t>{0, 2}));
        CHECK((indices(Matcher("foo", ia.matching(Matching::Suffix)).match_list(hs)) == std::vector<uint32_t>{1, 2}));
        CHECK((indices(Matcher("bar", ia.matching(Matching::Substring)).match_list(std::vector<std::string>{"xxbarxx", "bar", "nope", "foo_bar"})) == std::vector<uint32_t>{0, 1, 3}));
    }
    {  // multi-pattern (src/matcher/multi.rs:165-237)
        const Config ia = Config().sort(SortStrategy::IndexAsc);
        CHECK((indices(multi("foo !bar", ia).match_list(std::vector<std::string>{"foobar", "foo", "barfoo", "bar", "qux"})) == std::vector<uint32_t>{1}));
        const std::vector<std::string> hs = {"foo/bar", "bar/foo", "foo", "foobar"};
        CHECK((indices(multi("foo !^bar", ia).match_list(hs)) == std::vector<uint32_t>{0, 2, 3}));
        CHECK((indices(multi("foo !bar$", ia).match_list(hs)) == std::vector<uint32_t>{1, 2}));
        const std::vector<std::string> h3 = {"foo", "xfoox", "bar"};
        auto single = Matcher("foo", ia).match_list(h3);
        auto combined = multi("foo foo", ia).match_list(h3);
        CHECK(combined.size() == single.size());
        for (size_t i = 0; i < combined.size() && i < single.size(); i++)
            CHECK(combined[i].index == single[i].index && combined[i].score == single[i].score * 2 && combined[i].exact == single[i].exact);
        CHECK(multi("foo !foo", Config()).match_list(std::vector<std::string>{"foo", "foobar"}).empty());
        auto sorted = multi("foo bar", Config()).match_list(std::vector<std::string>{"xfoobarx", "foobar", "zzz"});
        CHECK(sorted.size() == 2 && sorted[0].index == 1 && sorted[0].score >= sorted[1].score);
        // pattern_max_typos_override_applies_per_pattern (:352-368)
        auto m = Matcher::from_patterns({Pattern("foo"), Pattern("barz").max_typos(1)}, ia.max_typos(0)).match_list(std::vector<std::string>{"foo bar", "fox bar"});
        CHECK((indices(m) == std::vector<uint32_t>{0}));
    }
    {  // set_pattern / set_config on a resident corpus (src/matcher/mod.rs:143-176), match_list_parallel == match_list (parallel.rs:104-130)
        std::vector<std::string> hs;
        for (int i = 0; i < 5000; i++) hs.push_back(i % 7 == 0 ? "xx_dead_be_xx" : i % 11 == 0 ? "deadbeef" : "nomatch-" + std::to_string(i));
        const Corpus corpus(hs);
        Matcher m("dead");
        auto a = m.match_list(corpus);
        m.set_pattern("deadbe");
        auto b = m.match_list(corpus);
        CHECK(b == Matcher("deadbe").match_list(corpus) && a.size() >= b.size() && !b.empty());
        m.set_config(Config().sort(SortStrategy::IndexDesc));
        auto d = m.match_list(corpus);
        CHECK(d.size() == b.size() && std::is_sorted(d.begin(), d.end(), [](const Match& x, const Match& y) { return x.index > y.index; }));
        for (size_t t : {1, 2, 8}) CHECK(m.match_list_parallel(corpus, t) == d);
        // match_list_parallel with one DEVICE per worker (parallel.rs:18-89): shards share the device when the box has fewer GPUs
        for (int shards : {1, 3}) {
            const ShardedCorpus sc(hs, shards, false, true);
            CHECK(sc.len() == hs.size() && sc.shards() == shards && m.match_list_parallel(sc) == d);
            Matcher byscore("deadbe");
            CHECK(byscore.match_list_parallel(sc) == b);
        }
        // ... and with one PROCESS per worker: the library's own RCCL communicator, here a world of the one rank this box allows
        {
            ShardComm comm(ShardComm::unique_id(), 0, 1);
            CHECK(comm.rank() == 0 && comm.world() == 1);
            Matcher byscore("deadbe");
            CHECK(byscore.match_list_parallel(corpus, 0, comm) == b && byscore.match_list_parallel(corpus, 0, comm, true) == b);
            auto shifted = m.match_list_parallel(corpus, 100, comm);
            CHECK(shifted.size() == d.size() && !shifted.empty() && shifted[0].index == d[0].index + 100);
        }
    }
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::string(argv[1]) == "gpu";
    host_side();
    if (gpu) gpu_side();
    else no_gpu_fails_loudly();
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("test_facade %s: ok\n", gpu ? "gpu" : "host");
    return 0;
}
