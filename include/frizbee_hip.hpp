// frizbee_hip.hpp - C++ host side over the C ABI of frizbee_hip.h, mirroring the reference crate's public interface for the
// list-matching path (the reference is Rust; there is no Rust toolchain in the build image, so the host-language mirror is C++):
// same type names, field names, defaults, method names and argument meaning, and the reference's panics become exceptions with
// the reference's panic text.  Header-only; link with -lfrizbee_hip.
//
//   reference                                            here
//   ---------------------------------------------------  ---------------------------------------------------------------
//   Scoring, Config (+ builder methods)   src/lib.rs:236-271, 439-478     frizbee::Scoring, frizbee::Config
//   CaseMatching / UnicodeMatching / SortStrategy / Matching  src/lib.rs:311-427   enum classes of the same names
//   Match { index, score, exact }          src/lib.rs:141-153             frizbee::Match
//   Pattern, PatternConfig, parse_query    src/pattern.rs:9-18, 186-262   frizbee::Pattern, frizbee::PatternConfig
//   Matcher::new / from_patterns / from_query / set_pattern / set_config / match_list / match_list_parallel / match_list_indices
//                                          src/matcher/mod.rs:90-222, parallel.rs:18-89          frizbee::Matcher
//   (the borrowed &[S: AsRef<str>])                                        frizbee::Corpus: the list packed once, resident in HBM
#pragma once
#include <cstdint>
#include <array>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "frizbee_hip.h"

namespace frizbee {

// The reference panics (assert!) in these situations; the message is the reference's panic text.
struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc == FZB_OK) return;
    const std::string msg = fzb_last_error();
    if (rc == FZB_ERR_PANIC) throw Panic(msg);
    throw Error(rc, msg);
}

enum class CaseMatching { Ignore = FZB_CASE_IGNORE, Smart = FZB_CASE_SMART, Respect = FZB_CASE_RESPECT };
enum class UnicodeMatching { Ignore = FZB_UNICODE_IGNORE, Smart = FZB_UNICODE_SMART, Always = FZB_UNICODE_ALWAYS };
enum class SortStrategy {
    ScoreThenIndexAsc = FZB_SORT_SCORE_THEN_INDEX_ASC,
    ScoreThenIndexDesc = FZB_SORT_SCORE_THEN_INDEX_DESC,
    IndexAsc = FZB_SORT_INDEX_ASC,
    IndexDesc = FZB_SORT_INDEX_DESC
};
enum class Matching { Fuzzy = FZB_MATCH_FUZZY, Exact = FZB_MATCH_EXACT, Prefix = FZB_MATCH_PREFIX, Suffix = FZB_MATCH_SUFFIX, Substring = FZB_MATCH_SUBSTRING };

struct Scoring {  // defaults: src/const.rs:1-10
    uint16_t match_score = 12, mismatch_penalty = 6, gap_open_penalty = 5, gap_extend_penalty = 1;
    uint16_t prefix_bonus = 12, capitalization_bonus = 4, matching_case_bonus = 4, exact_match_bonus = 8, delimiter_bonus = 4;
    fzb_scoring raw() const {
        return fzb_scoring{match_score, mismatch_penalty, gap_open_penalty, gap_extend_penalty, prefix_bonus, capitalization_bonus, matching_case_bonus, exact_match_bonus, delimiter_bonus};
    }
    bool operator==(const Scoring& o) const {
        return match_score == o.match_score && mismatch_penalty == o.mismatch_penalty && gap_open_penalty == o.gap_open_penalty && gap_extend_penalty == o.gap_extend_penalty &&
               prefix_bonus == o.prefix_bonus && capitalization_bonus == o.capitalization_bonus && matching_case_bonus == o.matching_case_bonus &&
               exact_match_bonus == o.exact_match_bonus && delimiter_bonus == o.delimiter_bonus;
    }
};

struct Config {  // src/lib.rs:236-271; the builder methods take and return by value like the Rust ones
    std::optional<uint16_t> max_typos_ = 0;
    CaseMatching casing_ = CaseMatching::Smart;
    UnicodeMatching unicode_ = UnicodeMatching::Smart;
    Matching matching_ = Matching::Fuzzy;
    SortStrategy sort_ = SortStrategy::ScoreThenIndexAsc;
    Scoring scoring_;
    // Not in the reference: which reference CPU backend the results are bit-exact against (see frizbee_hip.h). 0 / 0 = the one
    // frizbee would select on this host.
    uint16_t pf_lanes = 0, sw_lanes = 0;

    Config max_typos(std::optional<uint16_t> v) const { Config c = *this; c.max_typos_ = v; return c; }
    Config casing(CaseMatching v) const { Config c = *this; c.casing_ = v; return c; }
    Config unicode(UnicodeMatching v) const { Config c = *this; c.unicode_ = v; return c; }
    Config matching(Matching v) const { Config c = *this; c.matching_ = v; return c; }
    Config sort(SortStrategy v) const { Config c = *this; c.sort_ = v; return c; }
    Config scoring(const Scoring& v) const { Config c = *this; c.scoring_ = v; return c; }
    Config lanes(uint16_t pf, uint16_t sw = 0) const { Config c = *this; c.pf_lanes = pf; c.sw_lanes = sw; return c; }

    fzb_config raw() const {
        fzb_config r{};
        r.max_typos = max_typos_ ? (int32_t)*max_typos_ : -1;
        r.casing = (int32_t)casing_;
        r.unicode = (int32_t)unicode_;
        r.sort = (int32_t)sort_;
        r.scoring = scoring_.raw();
        r.pf_lanes = pf_lanes;
        r.sw_lanes = sw_lanes;
        r.matching = (int32_t)matching_;
        return r;
    }
};

struct Match {  // src/lib.rs:141-153
    uint32_t index;
    uint16_t score;
    bool exact;
    bool operator==(const Match& o) const { return index == o.index && score == o.score && exact == o.exact; }
};

struct MatchIndices {  // src/lib.rs:189-199
    uint16_t score;
    uint32_t index;
    bool exact;
    std::vector<uint32_t> indices;  // matched haystack byte positions, reverse order
    bool operator==(const MatchIndices& o) const { return index == o.index && score == o.score && exact == o.exact && indices == o.indices; }
};

struct PatternConfig {  // src/pattern.rs:230-244: every field optional, nullopt inherits the matcher's Config
    std::optional<uint16_t> max_typos;
    std::optional<CaseMatching> casing;
    std::optional<UnicodeMatching> unicode;
    std::optional<Matching> matching;
    std::optional<Scoring> scoring;
};

struct Pattern {  // src/pattern.rs:9-18
    std::string needle;
    bool negated = false;
    PatternConfig config;

    Pattern() = default;
    Pattern(const char* n) : needle(n) {}  // `impl From<&str> for Pattern`: matched literally, no syntax
    Pattern(std::string n, PatternConfig c = {}) : needle(std::move(n)), config(std::move(c)) {}
    Pattern with_negated(bool v) const { Pattern p = *this; p.negated = v; return p; }
    Pattern max_typos(std::optional<uint16_t> v) const { Pattern p = *this; p.config.max_typos = v; return p; }
    Pattern matching(std::optional<Matching> v) const { Pattern p = *this; p.config.matching = v; return p; }
    Pattern casing(std::optional<CaseMatching> v) const { Pattern p = *this; p.config.casing = v; return p; }

    // `Pattern::parse_query(query)` (src/pattern.rs:186-222): `^foo` prefix, `foo$` suffix, `^foo$` exact, `'foo` substring, `!foo`
    // negated (substring unless anchored), backslash escapes, atoms with an empty needle dropped
    static std::vector<Pattern> parse_query(std::string_view query) {
        fzb_pattern* arr = nullptr;
        size_t n = 0;
        check(fzb_parse_query((const uint8_t*)query.data(), query.size(), &arr, &n));
        std::vector<Pattern> out;
        for (size_t i = 0; i < n; i++) {
            Pattern p(std::string((const char*)arr[i].needle_utf8, arr[i].needle_len));
            p.negated = arr[i].negated != 0;
            if (arr[i].matching >= 0) p.config.matching = (Matching)arr[i].matching;
            out.push_back(std::move(p));
        }
        fzb_patterns_free(arr, n);
        return out;
    }
};

// The haystack list `match_list(&haystacks)` borrows: packed and uploaded once, it stays resident in HBM across queries.
class Corpus {
  public:
    template <typename Strings>  // any range of things convertible to std::string_view
    explicit Corpus(const Strings& haystacks) {
        std::string bytes;
        std::vector<uint64_t> ends;
        for (const auto& h : haystacks) {
            const std::string_view v(h);
            bytes.append(v.data(), v.size());
            ends.push_back(bytes.size());
        }
        fzb_corpus* c = nullptr;
        check(fzb_corpus_upload((const uint8_t*)bytes.data(), ends.data(), ends.size(), &c));
        h_.reset(c);
    }
    size_t len() const { return fzb_corpus_len(h_.get()); }
    const fzb_corpus* raw() const { return h_.get(); }

  private:
    struct Del { void operator()(fzb_corpus* c) const { fzb_corpus_free(c); } };
    std::unique_ptr<fzb_corpus, Del> h_;
};

// The list cut into one contiguous shard per GPU (fzb_corpus_upload_sharded): the devices of a node in the role of
// `match_list_parallel`'s worker threads (src/matcher/parallel.rs:18-89).  `gpus == 0`: every visible device.
class ShardedCorpus {
  public:
    template <typename Strings>
    explicit ShardedCorpus(const Strings& haystacks, int gpus = 0, bool by_bytes = false, bool oversubscribe = false) {
        std::string bytes;
        std::vector<uint64_t> ends;
        for (const auto& h : haystacks) {
            const std::string_view v(h);
            bytes.append(v.data(), v.size());
            ends.push_back(bytes.size());
        }
        if (gpus == 0) check(fzb_device_count(&gpus));
        fzb_sharded_corpus* c = nullptr;
        check(fzb_corpus_upload_sharded((const uint8_t*)bytes.data(), ends.data(), ends.size(), gpus,
                                        (by_bytes ? FZB_SHARD_BY_BYTES : 0) | (oversubscribe ? FZB_SHARD_OVERSUBSCRIBE : 0), &c));
        h_.reset(c);
    }
    size_t len() const { return fzb_sharded_corpus_len(h_.get()); }
    int shards() const { return fzb_sharded_corpus_shards(h_.get()); }
    const fzb_sharded_corpus* raw() const { return h_.get(); }

  private:
    struct Del { void operator()(fzb_sharded_corpus* c) const { fzb_sharded_corpus_free(c); } };
    std::unique_ptr<fzb_sharded_corpus, Del> h_;
};

// One process per GPU (fzb_shard_comm, csrc/host_rccl.hip): the ranks in the role of `match_list_parallel`'s worker threads, the runs moved by
// RCCL below the C ABI.  `ShardComm::unique_id()` on rank 0, its 128 bytes to the other ranks by whatever started them, then the constructor
// on every rank (collective) with the rank's GPU current.
class ShardComm {
  public:
    using Id = std::array<uint8_t, FZB_RCCL_ID_BYTES>;
    static Id unique_id() {
        Id id{};
        check(fzb_rccl_unique_id(id.data()));
        return id;
    }
    ShardComm(const Id& id, int rank, int world) {
        fzb_shard_comm* c = nullptr;
        check(fzb_shard_comm_create(id.data(), rank, world, &c));
        h_.reset(c);
    }
    int rank() const { return fzb_shard_comm_rank(h_.get()); }
    int world() const { return fzb_shard_comm_world(h_.get()); }
    fzb_shard_comm* raw() const { return h_.get(); }

  private:
    struct Del { void operator()(fzb_shard_comm* c) const { fzb_shard_comm_free(c); } };
    std::unique_ptr<fzb_shard_comm, Del> h_;
};

class Matcher {  // src/matcher/mod.rs:77-222
  public:
    // `Matcher::new(pattern, &config)`
    Matcher(const Pattern& pattern, const Config& config = {}) : Matcher(std::vector<Pattern>{pattern}, config) {}
    // `Matcher::from_patterns(&patterns, &config)`
    Matcher(const std::vector<Pattern>& patterns, const Config& config) : config_(config), patterns_(patterns) { build(); }
    static Matcher from_patterns(const std::vector<Pattern>& patterns, const Config& config = {}) { return Matcher(patterns, config); }
    // `Matcher::from_query(query, &config)`
    static Matcher from_query(std::string_view query, const Config& config = {}) { return Matcher(Pattern::parse_query(query), config); }

    const Config& config() const { return config_; }
    const std::vector<Pattern>& patterns() const { return patterns_; }

    // `set_pattern` / `set_config` (src/matcher/mod.rs:143-176).  A single plain pattern keeps its device workspace.
    void set_pattern(const Pattern& p) {
        if (single_ && plain(p) && patterns_.size() == 1 && plain(patterns_[0])) {
            check(fzb_matcher_set_pattern(single_.get(), (const uint8_t*)p.needle.data(), p.needle.size()));
            patterns_ = {p};
            return;
        }
        patterns_ = {p};
        build();
    }
    void set_config(const Config& c) {
        config_ = c;
        if (single_) {
            const fzb_config r = resolve(patterns_[0]);
            check(fzb_matcher_set_config(single_.get(), &r));
        } else {
            build();
        }
    }

    // `match_list(&haystacks)`: ordered per config.sort
    std::vector<Match> match_list(const Corpus& corpus) {
        fzb_match* out = nullptr;
        size_t n = 0;
        if (single_) check(fzb_match_list(single_.get(), corpus.raw(), &out, &n));
        else check(fzb_multi_match_list(multi_.get(), corpus.raw(), &out, &n));
        return take(out, n);
    }
    template <typename Strings>
    std::vector<Match> match_list(const Strings& haystacks) { return match_list(Corpus(haystacks)); }

    // `match_list_indices(&haystacks)` (src/matcher/mod.rs:234-275; multi-pattern: match_one_indices_multi, multi.rs:56-82).
    // `selection`: corpus indices standing in for the haystack list (the top of a match_list result); empty optional = the whole corpus.
    std::vector<MatchIndices> match_list_indices(const Corpus& corpus, const std::optional<std::vector<uint32_t>>& selection = std::nullopt) {
        if (selection && selection->empty()) return {};
        fzb_match_indices* out = nullptr;
        uint32_t* pos = nullptr;
        size_t n = 0;
        const uint32_t* sel = selection ? selection->data() : nullptr;
        const size_t nsel = selection ? selection->size() : 0;
        if (single_) check(fzb_match_list_indices(single_.get(), corpus.raw(), sel, nsel, &out, &n, &pos));
        else check(fzb_multi_match_list_indices(multi_.get(), corpus.raw(), sel, nsel, &out, &n, &pos));
        std::vector<MatchIndices> v(n);
        for (size_t i = 0; i < n; i++)
            v[i] = MatchIndices{out[i].score, out[i].index, out[i].exact != 0, std::vector<uint32_t>(pos + out[i].positions_begin, pos + out[i].positions_begin + out[i].positions_len)};
        fzb_match_indices_free(out, pos);
        return v;
    }
    template <typename Strings>
    std::vector<MatchIndices> match_list_indices(const Strings& haystacks) { return match_list_indices(Corpus(haystacks)); }

    // The per-item side of the interface (src/matcher/mod.rs:277-371), served by one batched device pass in list order:
    // `match_iter(haystacks)`: the matches in haystack order whatever config.sort says; `match_one(haystack, index)`
    std::vector<Match> match_iter(const Corpus& corpus, uint32_t index_offset = 0) {
        fzb_match* out = nullptr;
        size_t n = 0;
        const size_t count = fzb_corpus_len(corpus.raw());
        if (single_) check(fzb_match_list_into(single_.get(), corpus.raw(), 0, count, index_offset, &out, &n));
        else check(fzb_multi_match_list_into(multi_.get(), corpus.raw(), 0, count, index_offset, &out, &n));
        return take(out, n);
    }
    template <typename Strings>
    std::vector<Match> match_iter(const Strings& haystacks) { return match_iter(Corpus(haystacks)); }
    std::optional<Match> match_one(const std::string& haystack, uint32_t index) {
        auto v = match_iter(Corpus(std::vector<std::string>{haystack}), index);
        return v.empty() ? std::nullopt : std::optional<Match>(v[0]);
    }
    // `match_iter_indices(haystacks)` / `match_one_indices(haystack, index)`
    std::vector<MatchIndices> match_iter_indices(const Corpus& corpus, uint32_t index_offset = 0) {
        fzb_match_indices* out = nullptr;
        uint32_t* pos = nullptr;
        size_t n = 0;
        if (single_) check(fzb_match_list_indices_into(single_.get(), corpus.raw(), nullptr, 0, index_offset, &out, &n, &pos));
        else check(fzb_multi_match_list_indices_into(multi_.get(), corpus.raw(), nullptr, 0, index_offset, &out, &n, &pos));
        std::vector<MatchIndices> v(n);
        for (size_t i = 0; i < n; i++)
            v[i] = MatchIndices{out[i].score, out[i].index, out[i].exact != 0, std::vector<uint32_t>(pos + out[i].positions_begin, pos + out[i].positions_begin + out[i].positions_len)};
        fzb_match_indices_free(out, pos);
        return v;
    }
    template <typename Strings>
    std::vector<MatchIndices> match_iter_indices(const Strings& haystacks) { return match_iter_indices(Corpus(haystacks)); }
    std::optional<MatchIndices> match_one_indices(const std::string& haystack, uint32_t index) {
        auto v = match_iter_indices(Corpus(std::vector<std::string>{haystack}), index);
        return v.empty() ? std::nullopt : std::optional<MatchIndices>(v[0]);
    }

    // `match_list_parallel(&haystacks, threads)`: same result for every thread count; threads == 0 panics like the reference
    std::vector<Match> match_list_parallel(const Corpus& corpus, size_t threads) {
        if (threads == 0) throw Panic("threads must be positive");
        return match_list(corpus);
    }
    template <typename Strings>
    std::vector<Match> match_list_parallel(const Strings& haystacks, size_t threads) {
        if (threads == 0) throw Panic("threads must be positive");
        return match_list(Corpus(haystacks));
    }

    // `match_list_parallel` with one DEVICE per worker (fzb_match_list_parallel_sharded): per shard pipeline + device sort on its GPU,
    // k-way merge of the runs on this thread; the same list `match_list` returns.  Single-pattern matchers (the multi-pattern
    // composition has no sharded entry point yet).
    std::vector<Match> match_list_parallel(const ShardedCorpus& corpus) {
        if (!single_) throw Error(FZB_ERR_INVALID, "match_list_parallel over a sharded corpus needs a single-pattern matcher");
        fzb_match* out = nullptr;
        size_t n = 0;
        check(fzb_match_list_parallel_sharded(single_.get(), corpus.raw(), &out, &n));
        return take(out, n);
    }
    // `match_list_parallel` with one PROCESS per worker (fzb_match_list_parallel_rccl, collective): this rank's share of the list (`shard`,
    // first global index `index_offset`); the whole list's result on rank 0 (`to_all`: on every rank), empty elsewhere.
    std::vector<Match> match_list_parallel(const Corpus& shard, uint32_t index_offset, ShardComm& comm, bool to_all = false) {
        if (!single_) throw Error(FZB_ERR_INVALID, "match_list_parallel over a communicator needs a single-pattern matcher");
        fzb_match* out = nullptr;
        size_t n = 0;
        check(fzb_match_list_parallel_rccl(single_.get(), shard.raw(), index_offset, comm.raw(), to_all ? FZB_GATHER_ALL : FZB_GATHER_ROOT, &out, &n));
        return take(out, n);
    }
    // how the runs of that call reached the root device (fzb_matcher_shard_report: gather form, peer access per shard)
    std::string shard_report() const { return single_ ? std::string(fzb_matcher_shard_report(single_.get())) : std::string(); }

  private:
    static bool plain(const Pattern& p) {
        return !p.negated && !p.config.max_typos && !p.config.casing && !p.config.unicode && !p.config.matching && !p.config.scoring;
    }
    fzb_config resolve(const Pattern& p) const {  // PatternConfig::resolve (src/pattern.rs:250-262)
        Config c = config_;
        if (p.config.max_typos) c.max_typos_ = p.config.max_typos;
        if (p.config.casing) c.casing_ = *p.config.casing;
        if (p.config.unicode) c.unicode_ = *p.config.unicode;
        if (p.config.matching) c.matching_ = *p.config.matching;
        if (p.config.scoring) c.scoring_ = *p.config.scoring;
        return c.raw();
    }
    void build() {  // build_patterns (src/matcher/mod.rs:178-190): one non-negated pattern -> the single-pattern matcher
        single_.reset();
        multi_.reset();
        size_t live = 0, last = 0;
        for (size_t i = 0; i < patterns_.size(); i++)
            if (!patterns_[i].needle.empty()) { live++; last = i; }
        if (live == 1 && !patterns_[last].negated) {
            const fzb_config r = resolve(patterns_[last]);
            fzb_matcher* m = nullptr;
            check(fzb_matcher_create(&r, (const uint8_t*)patterns_[last].needle.data(), patterns_[last].needle.size(), &m));
            single_.reset(m);
            if (patterns_.size() != 1) patterns_ = {patterns_[last]};
            return;
        }
        std::vector<fzb_pattern> raw(patterns_.size());
        for (size_t i = 0; i < patterns_.size(); i++) {
            const Pattern& p = patterns_[i];
            fzb_pattern& r = raw[i];
            r = fzb_pattern{};
            r.needle_utf8 = (const uint8_t*)p.needle.data();
            r.needle_len = p.needle.size();
            r.negated = p.negated;
            r.has_max_typos = p.config.max_typos.has_value();
            r.max_typos = p.config.max_typos.value_or(0);
            r.casing = p.config.casing ? (int32_t)*p.config.casing : -1;
            r.unicode = p.config.unicode ? (int32_t)*p.config.unicode : -1;
            r.matching = p.config.matching ? (int32_t)*p.config.matching : -1;
            r.has_scoring = p.config.scoring.has_value();
            if (p.config.scoring) r.scoring = p.config.scoring->raw();
        }
        const fzb_config c = config_.raw();
        fzb_multi_matcher* mm = nullptr;
        check(fzb_multi_matcher_create(&c, raw.data(), raw.size(), &mm));
        multi_.reset(mm);
    }
    static std::vector<Match> take(fzb_match* out, size_t n) {
        std::vector<Match> r(n);
        for (size_t i = 0; i < n; i++) r[i] = Match{out[i].index, out[i].score, out[i].exact != 0};
        fzb_matches_free(out);
        return r;
    }
    struct DelS { void operator()(fzb_matcher* m) const { fzb_matcher_free(m); } };
    struct DelM { void operator()(fzb_multi_matcher* m) const { fzb_multi_matcher_free(m); } };
    Config config_;
    std::vector<Pattern> patterns_;
    std::unique_ptr<fzb_matcher, DelS> single_;
    std::unique_ptr<fzb_multi_matcher, DelM> multi_;
};

// `iter::FuzzyMatchExt` (src/matcher/iter.rs:35-126): `haystacks.iter().fuzzy_match(needle, &config)`
template <typename Strings>
inline std::vector<Match> fuzzy_match(const Strings& haystacks, const std::string& needle, const Config& config = Config()) {
    return Matcher(needle.c_str(), config).match_iter(haystacks);
}
template <typename Strings>
inline std::vector<MatchIndices> fuzzy_match_indices(const Strings& haystacks, const std::string& needle, const Config& config = Config()) {
    return Matcher(needle.c_str(), config).match_iter_indices(haystacks);
}

}  // namespace frizbee
