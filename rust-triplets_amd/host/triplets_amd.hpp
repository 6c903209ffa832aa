// C++ host API over the C ABI of libtriplets_amd.so.  Type and member names follow the reference
// crate so that code written against it reads the same:
//   DenoiserConfig, ChunkingStrategy            src/config.rs:12-119
//   DenoiserPreprocessor::process               src/preprocessor/backends/denoiser_preprocessor.rs:291-307
//   WhitespaceTokenizer::tokenize/token_count   src/tokenizer.rs:55-65
//   SlidingWindowChunker::materialize           src/chunking/sliding_window.rs:11-117
//   RecordChunk / ChunkView                     src/data.rs:130-174
// plus the batched entry point the reference does not have (BatchPreprocessor).
// Errors: the reference's path has no error channel; here a failing GPU call throws triplets_amd::Error
// (there is no CPU fallback).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <deque>
#include <exception>
#include <map>
#include <mutex>
#include <memory>
#include <optional>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/triplets_amd.h"

namespace triplets_amd {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc) {
    if (rc < 0) throw Error(rc, std::string("triplets_amd: ") + trp_last_error());
}

struct DenoiserConfig {
    bool enabled = false;
    float max_digit_ratio = 0.35f;
    bool strip_markdown = true;
};

struct ChunkingStrategy {
    size_t max_window_tokens = 1024;
    std::vector<size_t> overlap_tokens = {64};
    float summary_fallback_weight = 0.35f;
    size_t summary_fallback_tokens = 512;
    float chunk_weight_floor = 0.1f;
    // The preprocessors of the strategy, in registration order (ChunkingStrategy::register_preprocessor,
    // src/config.rs:105-113).  The GPU path implements the reference's one TextPreprocessor, the denoiser:
    // `denoiser` is the first one registered, `more_denoisers` those after it.
    std::optional<DenoiserConfig> denoiser;
    std::vector<DenoiserConfig> more_denoisers;
    void register_denoiser(const DenoiserConfig& c) {
        if (!denoiser) denoiser = c; else more_denoisers.push_back(c);
    }
};

struct ChunkView {
    enum Kind { Window = 0, SummaryFallback = 1 } kind = Window;
    size_t index = 0, overlap = 0, span = 0;   // Window
    std::string strategy;                      // SummaryFallback ("head")
    float weight = 0.0f;                       // SummaryFallback
};

struct RecordChunk {
    std::string record_id;
    size_t section_idx = 0;
    ChunkView view;
    std::string text;
    size_t tokens_estimate = 0;
};

class Context {
public:
    explicit Context(int device = 0) { check(trp_create(device, &h_)); }
    ~Context() { trp_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    trp_ctx* raw() const { return h_; }

private:
    trp_ctx* h_ = nullptr;
};

namespace detail {
inline std::vector<uint32_t> overlaps_of(const ChunkingStrategy& st) {
    std::vector<uint32_t> v;
    for (size_t o : st.overlap_tokens) v.push_back(o > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o);
    return v;
}
inline trp_config make_config(const ChunkingStrategy& st, const std::vector<uint32_t>& ov) {
    trp_config c;
    std::memset(&c, 0, sizeof c);
    c.denoise_enabled = st.denoiser && st.denoiser->enabled;
    c.strip_markdown = st.denoiser ? st.denoiser->strip_markdown : true;
    c.max_digit_ratio = st.denoiser ? st.denoiser->max_digit_ratio : 0.35f;
    c.max_window_tokens = st.max_window_tokens > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)st.max_window_tokens;
    c.overlaps = ov.data();
    c.n_overlaps = (uint32_t)ov.size();
    c.summary_fallback_tokens =
        st.summary_fallback_tokens > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)st.summary_fallback_tokens;
    return c;
}
inline std::vector<trp_denoiser> more_of(const ChunkingStrategy& st) {
    std::vector<trp_denoiser> v;
    for (const DenoiserConfig& d : st.more_denoisers) v.push_back(trp_denoiser{d.enabled, d.strip_markdown, d.max_digit_ratio});
    return v;
}
}  // namespace detail

// Host-side view of one batch result (owns the arrays).
class BatchResult {
public:
    BatchResult() { std::memset(&r_, 0, sizeof r_); std::memset(&ct_, 0, sizeof ct_); }
    ~BatchResult() { trp_result_free(&r_); trp_chunk_text_free(&ct_); }
    BatchResult(const BatchResult&) = delete;
    BatchResult& operator=(const BatchResult&) = delete;
    BatchResult(BatchResult&& o) noexcept : r_(o.r_), ct_(o.ct_), weight_(o.weight_) {
        std::memset(&o.r_, 0, sizeof o.r_);
        std::memset(&o.ct_, 0, sizeof o.ct_);
    }
    BatchResult& operator=(BatchResult&& o) noexcept {
        if (this != &o) {
            trp_result_free(&r_);
            trp_chunk_text_free(&ct_);
            r_ = o.r_;
            ct_ = o.ct_;
            weight_ = o.weight_;
            std::memset(&o.r_, 0, sizeof o.r_);
            std::memset(&o.ct_, 0, sizeof o.ct_);
        }
        return *this;
    }
    // the text of chunk i as the GPU rendered it (results made with TRP_OUT_CHUNK_TEXT)
    bool has_chunk_text() const { return ct_.offsets != nullptr; }
    std::string_view chunk_text(size_t i) const {
        return {reinterpret_cast<const char*>(ct_.text) + ct_.offsets[i], (size_t)(ct_.offsets[i + 1] - ct_.offsets[i])};
    }

    const trp_result& raw() const { return r_; }
    size_t records() const { return r_.n_records; }
    bool dropped(size_t r) const { return r_.dropped[r] != 0; }
    std::string_view text(size_t r) const {
        return {reinterpret_cast<const char*>(r_.denoised) + r_.denoised_offsets[r],
                (size_t)(r_.denoised_offsets[r + 1] - r_.denoised_offsets[r])};
    }
    size_t token_count(size_t r) const { return r_.tok_offsets[r + 1] - r_.tok_offsets[r]; }
    std::string_view token(size_t r, size_t k) const {
        size_t i = r_.tok_offsets[r] + k;
        return {reinterpret_cast<const char*>(r_.denoised) + r_.tok_start[i], r_.tok_len[i]};
    }
    // The Vec<RecordChunk> the reference builds for section r (sliding_window.rs:40-115).
    std::vector<RecordChunk> chunks(size_t r, const std::string& record_id, size_t section_idx) const {
        std::vector<RecordChunk> out;
        const size_t total = token_count(r);
        for (uint64_t i = r_.chunk_offsets[r]; i < r_.chunk_offsets[r + 1]; i++) {
            const trp_chunk& c = r_.chunks[i];
            RecordChunk rc;
            rc.record_id = record_id;
            rc.section_idx = section_idx;
            rc.tokens_estimate = c.tok_end - c.tok_begin;
            if (c.kind == 0) {
                rc.view.kind = ChunkView::Window;
                rc.view.index = c.index; rc.view.overlap = c.overlap; rc.view.span = c.span;
            } else {
                rc.view.kind = ChunkView::SummaryFallback;
                rc.view.strategy = "head";
                rc.view.weight = weight_;
            }
            if (has_chunk_text()) {
                rc.text = std::string(chunk_text(i));   // rendered on the GPU: neither the text nor the tokens came back
            } else if (c.kind == 0 && c.tok_begin == 0 && c.tok_end == total && c.span == total) {
                rc.text = std::string(text(r));   // single window: preprocessed text verbatim
            } else {
                for (uint32_t k = c.tok_begin; k < c.tok_end; k++) {
                    if (k > c.tok_begin) rc.text.push_back(' ');
                    rc.text.append(token(r, k));
                }
            }
            out.push_back(std::move(rc));
        }
        return out;
    }

private:
    friend class BatchPreprocessor;
    trp_result r_;
    trp_chunk_text ct_;
    float weight_ = 0.35f;
};

// Batched path: many sections per call.
class BatchPreprocessor {
public:
    explicit BatchPreprocessor(Context& ctx) : ctx_(ctx) {}
    BatchResult run(const std::vector<std::string_view>& sections, const ChunkingStrategy& st) {
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offs(1, 0);
        size_t total = 0;
        for (auto s : sections) total += s.size();
        bytes.reserve(total);
        for (auto s : sections) {
            bytes.insert(bytes.end(), s.begin(), s.end());
            offs.push_back(bytes.size());
        }
        return run_packed(bytes.data(), offs.data(), sections.size(), st);
    }
    // Only the arrays named by `outputs` (TRP_OUT_* flags) travel back; the per-record arrays always do.
    BatchResult run_select(const std::vector<std::string_view>& sections, const ChunkingStrategy& st, uint32_t outputs) {
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offs(1, 0);
        for (auto s : sections) {
            bytes.insert(bytes.end(), s.begin(), s.end());
            offs.push_back(bytes.size());
        }
        auto ov = detail::overlaps_of(st);
        trp_config c = detail::make_config(st, ov);
        BatchResult out;
        out.weight_ = st.summary_fallback_weight;
        auto more = detail::more_of(st);
        check(trp_preprocess_batch_fold(ctx_.raw(), bytes.data(), offs.data(), sections.size(), &c, more.data(),
                                        (uint32_t)more.size(), outputs, &out.r_,
                                        (outputs & TRP_OUT_CHUNK_TEXT) ? &out.ct_ : nullptr));
        return out;
    }
    BatchResult run_packed(const uint8_t* bytes, const uint64_t* offsets, size_t n, const ChunkingStrategy& st) {
        auto ov = detail::overlaps_of(st);
        trp_config c = detail::make_config(st, ov);
        BatchResult out;
        out.weight_ = st.summary_fallback_weight;
        auto more = detail::more_of(st);
        check(trp_preprocess_batch_fold(ctx_.raw(), bytes, offsets, n, &c, more.data(), (uint32_t)more.size(), TRP_OUT_ALL,
                                        &out.r_, nullptr));
        return out;
    }

private:
    Context& ctx_;
};

// One batch over several GPUs of this process: one Context and one host thread per entry of `devices` (HIP device
// ordinals; an ordinal may be listed more than once).  The batch is cut into byte-balanced contiguous record
// ranges, every shard runs the whole path on its own context at the same time as the others (records are
// independent, src/chunking/algorithm.rs:7-13: nothing is exchanged between GPUs), and the result answers per
// global record index.
class ShardedResult {
public:
    size_t records() const { return first_.empty() ? 0 : first_.back(); }
    size_t shards() const { return parts_.size(); }
    const BatchResult& shard(size_t s) const { return parts_[s]; }
    // (shard, record within the shard) of global record r
    std::pair<size_t, size_t> locate(size_t r) const {
        const size_t s = (size_t)(std::upper_bound(first_.begin(), first_.end(), r) - first_.begin()) - 1;
        return {s, r - first_[s]};
    }
    bool dropped(size_t r) const { auto [s, i] = locate(r); return parts_[s].dropped(i); }
    std::string_view text(size_t r) const { auto [s, i] = locate(r); return parts_[s].text(i); }
    size_t token_count(size_t r) const { auto [s, i] = locate(r); return parts_[s].token_count(i); }
    std::vector<RecordChunk> chunks(size_t r, const std::string& record_id, size_t section_idx) const {
        auto [s, i] = locate(r);
        return parts_[s].chunks(i, record_id, section_idx);
    }

private:
    friend class ShardedPreprocessor;
    std::vector<BatchResult> parts_;
    std::vector<size_t> first_;   // first_[s] = first global record of shard s; one more entry: the record count
};

class ShardedPreprocessor {
public:
    explicit ShardedPreprocessor(const std::vector<int>& devices) {
        if (devices.empty()) throw Error(TRP_E_ARG, "triplets_amd: ShardedPreprocessor needs at least one device");
        for (int d : devices) ctxs_.emplace_back(new Context(d));
    }
    // every GPU the runtime shows
    static std::vector<int> all_devices() {
        std::vector<int> v;
        for (int d = 0; d < trp_device_count(); d++) v.push_back(d);
        return v;
    }
    size_t shards() const { return ctxs_.size(); }

    ShardedResult run(const std::vector<std::string_view>& sections, const ChunkingStrategy& st) {
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> offs(1, 0);
        for (auto s : sections) {
            bytes.insert(bytes.end(), s.begin(), s.end());
            offs.push_back(bytes.size());
        }
        return run_packed(bytes.data(), offs.data(), sections.size(), st);
    }
    ShardedResult run_packed(const uint8_t* bytes, const uint64_t* offsets, size_t n, const ChunkingStrategy& st) {
        const size_t G = ctxs_.size();
        for (const DenoiserConfig& d : st.more_denoisers)
            if (d.enabled) throw Error(TRP_E_ARG, "triplets_amd: a strategy with several denoisers runs on BatchPreprocessor");
        ShardedResult out;
        out.first_.assign(G + 1, n);
        out.first_[0] = 0;
        const uint64_t N = offsets[n];
        for (size_t k = 1; k < G; k++) {   // shard k starts at the first record whose offset is >= k * N / G
            const uint64_t target = (uint64_t)((unsigned __int128)N * k / G);
            size_t r = (size_t)(std::lower_bound(offsets, offsets + n + 1, target) - offsets);
            out.first_[k] = std::max(std::min(r, n), out.first_[k - 1]);
        }
        out.parts_.resize(G);
        std::vector<std::exception_ptr> errs(G);
        std::vector<std::thread> pool;
        for (size_t k = 0; k < G; k++)
            pool.emplace_back([&, k]() {
                try {
                    const size_t r0 = out.first_[k], r1 = out.first_[k + 1];
                    std::vector<uint64_t> rebased(r1 - r0 + 1);
                    for (size_t i = 0; i <= r1 - r0; i++) rebased[i] = offsets[r0 + i] - offsets[r0];
                    BatchPreprocessor bp(*ctxs_[k]);
                    out.parts_[k] = bp.run_packed(bytes + offsets[r0], rebased.data(), r1 - r0, st);
                } catch (...) {
                    errs[k] = std::current_exception();
                }
            });
        for (auto& t : pool) t.join();
        for (auto& e : errs)
            if (e) std::rethrow_exception(e);
        return out;
    }

private:
    std::vector<std::unique_ptr<Context>> ctxs_;
};

// ---- the steps right after the path ----------------------------------------------------------------------
namespace detail {
inline std::vector<std::string> take_texts(trp_chunk_text& ct, size_t n) {
    std::vector<std::string> out;
    out.reserve(n);
    for (size_t i = 0; i < n; i++)
        out.emplace_back(reinterpret_cast<const char*>(ct.text) + ct.offsets[i], (size_t)(ct.offsets[i + 1] - ct.offsets[i]));
    trp_chunk_text_free(&ct);
    return out;
}
inline void pack(const std::vector<std::string_view>& texts, std::vector<uint8_t>& bytes, std::vector<uint64_t>& offs) {
    offs.assign(1, 0);
    for (auto s : texts) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        offs.push_back(bytes.size());
    }
}
}  // namespace detail

// record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532) for many records in one GPU batch: headings and
// section texts in order, non-blank parts followed by the platform newline, cut to max_tokens tokens joined by one
// space when there are more (0 = no limit), the record id when nothing is left.
struct Bm25Section {
    std::optional<std::string_view> heading;
    std::string_view text;
};
struct Bm25Record {
    std::string_view id;
    std::vector<Bm25Section> sections;
};
inline std::vector<std::string> record_bm25_texts(Context& ctx, const std::vector<Bm25Record>& records, size_t max_tokens,
                                                  const std::string& newline = "\n") {
    std::vector<std::string_view> parts;
    std::vector<uint64_t> doc(1, 0);
    for (const auto& r : records) {
        for (const auto& s : r.sections) {
            if (s.heading) parts.push_back(*s.heading);
            parts.push_back(s.text);
        }
        doc.push_back(parts.size());
    }
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offs;
    detail::pack(parts, bytes, offs);
    trp_chunk_text ct;
    check(trp_bm25_texts(ctx.raw(), bytes.data(), offs.data(), parts.size(), doc.data(), records.size(),
                         max_tokens > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_tokens, newline.c_str(), &ct));
    auto out = detail::take_texts(ct, records.size());
    for (size_t i = 0; i < out.size(); i++)
        if (out[i].empty()) out[i] = std::string(records[i].id);   // bm25_backend.rs:517-519
    return out;
}
// The BM25 query cap (bm25_backend.rs:219-225): every text cut to its first `limit` tokens joined by one space, or
// left as it is when it has no more than that.
inline std::vector<std::string> cap_tokens(Context& ctx, const std::vector<std::string_view>& texts, size_t limit) {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offs;
    detail::pack(texts, bytes, offs);
    trp_chunk_text ct;
    check(trp_head_tokens_text(ctx.raw(), bytes.data(), offs.data(), texts.size(),
                               limit > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)limit, &ct));
    return detail::take_texts(ct, texts.size());
}
// Bodies of selected chunks of a batch cut to a token budget (the body half of decorate_chunk,
// src/sampler/mod.rs:1497-1504): the sections are preprocessed and chunked with `st`, chunk chunk_index[i] of the
// batch (in the batch's chunk order) is cut to budget[i] tokens, verbatim when the budget covers it.
inline std::vector<std::string> chunk_bodies(Context& ctx, const std::vector<std::string_view>& sections,
                                             const ChunkingStrategy& st, const std::vector<uint32_t>& chunk_index,
                                             const std::vector<uint32_t>& budget) {
    if (chunk_index.size() != budget.size()) throw Error(TRP_E_ARG, "triplets_amd: chunk_index and budget differ in length");
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offs;
    detail::pack(sections, bytes, offs);
    auto ov = detail::overlaps_of(st);
    trp_config c = detail::make_config(st, ov);
    trp_chunk_text ct;
    check(trp_chunk_bodies(ctx.raw(), bytes.data(), offs.data(), sections.size(), &c, chunk_index.data(), budget.data(),
                           chunk_index.size(), &ct));
    return detail::take_texts(ct, chunk_index.size());
}
// decorate_chunk (src/sampler/mod.rs:1470-1508; rng-free twin :1582-1620) for many chunks at once.  Every item is a
// chunk's text and token count (RecordChunk::text / tokens_estimate) with the prefix the host's KvpPrefixSampler
// produced for it (nullopt = the sampler gave none: the chunk is left as it is, :1481-1483).  Returns text and
// tokens_estimate.
struct DecorateItem {
    std::string_view chunk_text;
    std::optional<std::string_view> prefix;
};
struct Decorated {
    std::string text;
    size_t tokens_estimate;
};
inline std::vector<Decorated> decorate_chunks(Context& ctx, const std::vector<DecorateItem>& items, size_t max_window_tokens,
                                              const std::string& newline = "\n") {
    // one batch: the chunk texts, then the prefixes; token counts from the batch, cuts rendered on the GPU
    const size_t n = items.size();
    // (a prefix travels with one extra token behind it: a cut to at most its own token count is then always a real
    // cut, rendered as tokens joined by one space - what :1490-1494 does even when nothing is dropped)
    std::vector<std::string> guarded(n);
    for (size_t i = 0; i < n; i++)
        if (items[i].prefix) guarded[i] = std::string(*items[i].prefix) + " .";
    std::vector<std::string_view> texts;
    for (const auto& it : items) texts.push_back(it.chunk_text);
    for (size_t i = 0; i < n; i++) texts.push_back(guarded[i]);
    ChunkingStrategy plain;                       // no denoiser: the texts are tokenised as they are
    plain.max_window_tokens = 0xFFFFFFFFull;      // one window per non-empty text: chunk k = k-th text with a token
    plain.overlap_tokens = {0};
    plain.summary_fallback_tokens = 0;
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> offs;
    detail::pack(texts, bytes, offs);
    auto ov = detail::overlaps_of(plain);
    trp_config c = detail::make_config(plain, ov);
    BatchResult counts;
    {
        BatchPreprocessor bp(ctx);
        counts = bp.run_select(texts, plain, 0);   // per-record arrays only: token and chunk offsets
    }
    const trp_result& cr = counts.raw();
    std::vector<uint32_t> index, budget;
    std::vector<size_t> slot(n, (size_t)-1);   // which rendered text item i uses
    const uint64_t mw = max_window_tokens;
    std::vector<int> mode(n, 0);               // 0 unchanged, 1 prefix + body (maybe cut), 2 prefix tokens only
    for (size_t i = 0; i < n; i++) {
        if (!items[i].prefix) continue;                                      // :1481-1483
        const uint64_t pt = counts.token_count(n + i) - 1, bt = counts.token_count(i), total = pt + bt;
        if (mw > 0 && total > mw && pt >= mw) {                              // :1488-1495
            mode[i] = 2;
            slot[i] = index.size();
            index.push_back((uint32_t)cr.chunk_offsets[n + i]);
            budget.push_back((uint32_t)mw);
        } else {
            mode[i] = 1;
            const bool cut = mw > 0 && total > mw;                           // :1496-1502
            if (bt) {
                slot[i] = index.size();
                index.push_back((uint32_t)cr.chunk_offsets[i]);
                budget.push_back(cut ? (uint32_t)(mw - pt) : 0xFFFFFFFFu);
            }
        }
    }
    std::vector<std::string> rendered;
    if (!index.empty()) {
        trp_chunk_text ct;
        check(trp_chunk_bodies(ctx.raw(), bytes.data(), offs.data(), texts.size(), &c, index.data(), budget.data(), index.size(), &ct));
        rendered = detail::take_texts(ct, index.size());
    }
    std::vector<Decorated> out(n);
    for (size_t i = 0; i < n; i++) {
        const uint64_t pt = items[i].prefix ? counts.token_count(n + i) - 1 : 0, bt = counts.token_count(i), total = pt + bt;
        const std::string body = slot[i] != (size_t)-1 ? rendered[slot[i]] : std::string();
        if (mode[i] == 0) out[i] = {std::string(items[i].chunk_text), (size_t)bt};
        else if (mode[i] == 2) out[i] = {body, (size_t)mw};
        else {
            const bool cut = mw > 0 && total > mw;
            // a body without a token is its own text (all whitespace or empty), as in the reference's format!
            const std::string b = bt ? body : std::string(items[i].chunk_text);
            out[i] = {std::string(*items[i].prefix) + newline + b, (size_t)(cut ? mw : total)};
        }
    }
    return out;
}

// Per-call mirrors of the reference's public types (each call is a batch of one).
class DenoiserPreprocessor {
public:
    DenoiserPreprocessor(Context& ctx, DenoiserConfig config) : ctx_(ctx), config_(config) {}
    std::optional<std::string> process(std::string_view text) const {
        ChunkingStrategy st;
        st.denoiser = config_;
        auto ov = detail::overlaps_of(st);
        trp_config c = detail::make_config(st, ov);
        uint8_t* out = nullptr;
        size_t len = 0;
        int rc = trp_denoise(ctx_.raw(), reinterpret_cast<const uint8_t*>(text.data()), text.size(), &c, &out, &len);
        check(rc);
        if (rc == 0) return std::nullopt;
        std::string s(reinterpret_cast<char*>(out), len);
        trp_free(out);
        return s;
    }

private:
    Context& ctx_;
    DenoiserConfig config_;
};

class WhitespaceTokenizer {
public:
    explicit WhitespaceTokenizer(Context& ctx) : ctx_(ctx) {}
    std::vector<std::string_view> tokenize(std::string_view text) const {
        uint32_t *starts = nullptr, *lens = nullptr;
        size_t n = 0;
        check(trp_tokenize(ctx_.raw(), reinterpret_cast<const uint8_t*>(text.data()), text.size(), &starts, &lens, &n));
        std::vector<std::string_view> out;
        out.reserve(n);
        for (size_t i = 0; i < n; i++) out.push_back(text.substr(starts[i], lens[i]));
        trp_free(starts);
        trp_free(lens);
        return out;
    }
    size_t token_count(std::string_view text) const {
        size_t n = 0;
        check(trp_token_count(ctx_.raw(), reinterpret_cast<const uint8_t*>(text.data()), text.size(), &n));
        return n;
    }

private:
    Context& ctx_;
};

// normalize_inline_whitespace (src/utils.rs:18-33): whitespace runs -> one space, trimmed.
inline std::string normalize_inline_whitespace(Context& ctx, std::string_view text) {
    uint8_t* out = nullptr;
    size_t len = 0;
    check(trp_normalize_inline_whitespace(ctx.raw(), reinterpret_cast<const uint8_t*>(text.data()), text.size(), &out, &len));
    std::string s(reinterpret_cast<char*>(out), len);
    trp_free(out);
    return s;
}

class SlidingWindowChunker {
public:
    explicit SlidingWindowChunker(Context& ctx) : ctx_(ctx) {}
    std::vector<RecordChunk> materialize(const ChunkingStrategy& st, const std::string& record_id, size_t section_idx,
                                         std::string_view section_text) const {
        BatchPreprocessor bp(ctx_);
        BatchResult r = bp.run({section_text}, st);
        return r.chunks(0, record_id, section_idx);
    }

private:
    Context& ctx_;
};

// Chunk table filled by GPU batches and served per (record id, section index): what a sampler needs to call
// a GPU-backed chunker once per chunk selection.  The reference calls ChunkingAlgorithm::materialize per
// selection, concurrently from its worker threads (`Send + Sync`, src/chunking/algorithm.rs:5,
// src/sampler/mod.rs:2510-2537), and tokenises the RAW text of every section when it refreshes its record
// cache (record_has_long_anchor_or_context_section, src/sampler/mod.rs:1803-1812) - the place to call prefill.
//
// The reference's materialize is a pure function of the text it is handed (sliding_window.rs:18), and its record
// cache is a rolling one (src/ingestion.rs:18).  Both hold here: an entry remembers the length and a 64-bit hash
// of the text it was made from, a lookup with another text is a miss that replaces it (a record re-ingested under
// its old id never gets its old chunks), and the table keeps at most `budget_bytes` of chunk text, dropping the
// entries that were inserted first when it grows beyond that.  invalidate() drops a record by hand.
//
// Thread safety: every member may be called from any number of threads.  Lookups take a shared lock.  The GPU
// context is used by one thread at a time; sections that miss the table while another thread is on the GPU
// are queued and computed together, as one batch, by whichever thread gets the GPU next.
// The table is keyed by the strategy as well (window / overlap / fallback sizes and the denoiser settings):
// a changed ChunkingStrategy never sees chunks made for another one.
class ChunkCache {
public:
    struct SectionRef {
        std::string record_id;
        size_t section_idx;
        std::string_view text;
    };
    static constexpr size_t DEFAULT_BUDGET = (size_t)1 << 30;
    explicit ChunkCache(Context& ctx, size_t budget_bytes = DEFAULT_BUDGET) : ctx_(ctx), budget_(budget_bytes) {}

    // One batch for the chunks (with the strategy's denoiser, if any) and, when a denoiser is registered, one
    // pass-through batch over the same sections for the raw token counts.
    void prefill(const ChunkingStrategy& st, const std::vector<SectionRef>& sections) {
        std::lock_guard<std::mutex> gpu(gpu_mu_);
        compute(st, sections, nullptr);
    }
    // The chunks of one section (SlidingWindowChunker::materialize, sliding_window.rs:11-117).
    std::vector<RecordChunk> materialize(const ChunkingStrategy& st, const std::string& record_id, size_t section_idx,
                                         std::string_view section_text) {
        const Key key{strategy_hash(st), record_id, section_idx};
        const uint64_t th = text_hash(section_text);
        {
            std::shared_lock<std::shared_mutex> rd(table_mu_);
            auto it = table_.find(key);
            if (it != table_.end() && it->second.text_len == section_text.size() && it->second.text_hash == th)
                return it->second.chunks;
        }
        // miss (or the text under this id has changed): join the queue; the first thread to get the GPU computes
        // everything queued for this strategy
        Pending me{&st, SectionRef{record_id, section_idx, section_text}, key, false, nullptr, {}};
        {
            std::lock_guard<std::mutex> q(queue_mu_);
            queue_.push_back(&me);
        }
        {
            std::lock_guard<std::mutex> gpu(gpu_mu_);
            if (!me.done) drain();
        }
        if (me.error) std::rethrow_exception(me.error);
        return std::move(me.result);   // handed over by the batch itself: the entry may already have been evicted
    }
    // WhitespaceTokenizer.token_count(&section.text) of the RAW section text - what the reference's sampler
    // compares with max_window_tokens (src/sampler/mod.rs:1810) - for a section that is in the table.
    std::optional<size_t> token_count(const std::string& record_id, size_t section_idx) const {
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        auto it = raw_tokens_.find({record_id, section_idx});
        if (it == raw_tokens_.end()) return std::nullopt;
        return it->second.tokens;
    }
    // Token count of the preprocessed text (what the windows of materialize are cut from).
    std::optional<size_t> preprocessed_token_count(const ChunkingStrategy& st, const std::string& record_id,
                                                   size_t section_idx) const {
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        auto it = table_.find(Key{strategy_hash(st), record_id, section_idx});
        if (it == table_.end()) return std::nullopt;
        return it->second.tokens;
    }
    size_t size() const {
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        return table_.size();
    }
    size_t bytes() const {   // what the budget is compared with: chunk text + record ids of the entries
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        return bytes_;
    }
    size_t budget() const { return budget_; }
    void clear() {
        std::unique_lock<std::shared_mutex> wr(table_mu_);
        table_.clear();
        raw_tokens_.clear();
        order_.clear();
        bytes_ = 0;
    }
    // Drops every entry of one record (all sections, all strategies).  Returns the number of entries dropped.
    size_t invalidate(const std::string& record_id) {
        std::unique_lock<std::shared_mutex> wr(table_mu_);
        size_t n = 0;
        for (auto it = table_.begin(); it != table_.end();) {
            if (it->first.record_id == record_id) { bytes_ -= it->second.cost; it = table_.erase(it); n++; }
            else ++it;
        }
        for (auto it = raw_tokens_.lower_bound({record_id, 0}); it != raw_tokens_.end() && it->first.first == record_id;)
            it = raw_tokens_.erase(it);
        return n;
    }
    size_t gpu_batches() const { return batches_.load(); }   // batches run so far (for tests and tuning)

    // FNV-1a over everything of the strategy that changes the result.
    static uint64_t strategy_hash(const ChunkingStrategy& st) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](uint64_t v) {
            for (int i = 0; i < 8; i++) { h ^= (v >> (8 * i)) & 0xFFu; h *= 1099511628211ull; }
        };
        mix(st.max_window_tokens);
        mix(st.overlap_tokens.size());
        for (size_t o : st.overlap_tokens) mix(o);
        mix(st.summary_fallback_tokens);
        uint32_t w, r = 0;
        std::memcpy(&w, &st.summary_fallback_weight, 4);
        mix(w);
        const bool on = st.denoiser && st.denoiser->enabled;
        mix(on);
        if (on) {
            std::memcpy(&r, &st.denoiser->max_digit_ratio, 4);
            mix(r);
            mix(st.denoiser->strip_markdown);
        }
        for (const DenoiserConfig& d : st.more_denoisers) {   // (a disabled one changes nothing)
            if (!d.enabled) continue;
            std::memcpy(&r, &d.max_digit_ratio, 4);
            mix(0x100u | (d.strip_markdown ? 1u : 0u));
            mix(r);
        }
        return h;
    }
    // 64-bit hash of a section text (eight bytes per step, FNV-style multiply, final avalanche).
    static uint64_t text_hash(std::string_view t) {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)t.size();
        size_t i = 0;
        for (; i + 8 <= t.size(); i += 8) {
            uint64_t v;
            std::memcpy(&v, t.data() + i, 8);
            h = (h ^ v) * 0x100000001B3ull;
            h ^= h >> 29;
        }
        uint64_t v = 0;
        if (i < t.size()) std::memcpy(&v, t.data() + i, t.size() - i);
        h = (h ^ v) * 0x100000001B3ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        return h;
    }

private:
    struct Key {
        uint64_t strategy;
        std::string record_id;
        size_t section_idx;
        bool operator<(const Key& o) const {
            return std::tie(strategy, record_id, section_idx) < std::tie(o.strategy, o.record_id, o.section_idx);
        }
    };
    struct Entry {
        std::vector<RecordChunk> chunks;
        size_t tokens = 0;
        uint64_t text_hash = 0;
        size_t text_len = 0;
        size_t cost = 0;       // bytes counted against the budget
        uint64_t serial = 0;   // insertion number (the rolling order)
    };
    struct RawCount {
        size_t tokens;
    };
    struct Pending {
        const ChunkingStrategy* st;
        SectionRef ref;
        Key key;
        bool done;
        std::exception_ptr error;
        std::vector<RecordChunk> result;
    };

    // gpu_mu_ held.  Everything queued right now, one batch per strategy.
    void drain() {
        std::vector<Pending*> work;
        {
            std::lock_guard<std::mutex> q(queue_mu_);
            work.swap(queue_);
        }
        while (!work.empty()) {
            const uint64_t h = work.front()->key.strategy;
            std::vector<Pending*> same, rest;
            for (Pending* p : work) (p->key.strategy == h ? same : rest).push_back(p);
            std::vector<SectionRef> refs;
            for (Pending* p : same) refs.push_back(p->ref);
            std::exception_ptr err;
            std::vector<std::vector<RecordChunk>> made;
            try {
                compute(*same.front()->st, refs, &made);
            } catch (...) {
                err = std::current_exception();
            }
            for (size_t i = 0; i < same.size(); i++) {
                if (!err) same[i]->result = std::move(made[i]);
                same[i]->error = err;
                same[i]->done = true;
            }
            work.swap(rest);
        }
    }
    static size_t cost_of(const Entry& e, const std::string& record_id) {
        size_t n = sizeof(Entry) + record_id.size();
        for (const auto& c : e.chunks) n += sizeof(RecordChunk) + c.text.size() + c.record_id.size();
        return n;
    }
    // gpu_mu_ held.  `handed` (optional) receives a copy of every section's chunks.
    void compute(const ChunkingStrategy& st, const std::vector<SectionRef>& sections,
                 std::vector<std::vector<RecordChunk>>* handed) {
        if (handed) handed->assign(sections.size(), {});
        if (sections.empty()) return;
        std::vector<std::string_view> texts;
        texts.reserve(sections.size());
        for (const auto& s : sections) texts.push_back(s.text);
        BatchPreprocessor bp(ctx_);
        // descriptors + the chunk texts rendered on the GPU: 1.03 bytes per input byte over the link instead of the
        // 2.44 of every array (the preprocessed text and the token table stay on the device)
        BatchResult r = bp.run_select(texts, st, TRP_OUT_CHUNKS | TRP_OUT_CHUNK_TEXT);
        batches_++;
        std::vector<size_t> raw(sections.size());
        bool denoising = st.denoiser && st.denoiser->enabled;
        for (const DenoiserConfig& d : st.more_denoisers) denoising = denoising || d.enabled;
        if (denoising) {
            ChunkingStrategy plain = st;
            plain.denoiser.reset();
            plain.more_denoisers.clear();
            BatchResult t = bp.run_select(texts, plain, 0);   // per-record arrays only: the token offsets
            batches_++;
            for (size_t i = 0; i < sections.size(); i++) raw[i] = t.token_count(i);
        } else {
            for (size_t i = 0; i < sections.size(); i++) raw[i] = r.token_count(i);
        }
        const uint64_t h = strategy_hash(st);
        std::vector<Entry> made(sections.size());
        for (size_t i = 0; i < sections.size(); i++) {
            made[i].chunks = r.chunks(i, sections[i].record_id, sections[i].section_idx);
            made[i].tokens = r.token_count(i);
            made[i].text_hash = text_hash(sections[i].text);
            made[i].text_len = sections[i].text.size();
            made[i].cost = cost_of(made[i], sections[i].record_id);
            if (handed) (*handed)[i] = made[i].chunks;
        }
        std::unique_lock<std::shared_mutex> wr(table_mu_);
        for (size_t i = 0; i < sections.size(); i++) {
            Key key{h, sections[i].record_id, sections[i].section_idx};
            auto it = table_.find(key);
            if (it != table_.end()) { bytes_ -= it->second.cost; table_.erase(it); }
            made[i].serial = ++serial_;
            bytes_ += made[i].cost;
            order_.emplace_back(made[i].serial, key);
            table_.emplace(std::move(key), std::move(made[i]));
            raw_tokens_[{sections[i].record_id, sections[i].section_idx}] = RawCount{raw[i]};
        }
        // the rolling part: the oldest entries go while the table is over its budget (the newest one always stays)
        while (bytes_ > budget_ && order_.size() > 1) {
            auto [serial, key] = std::move(order_.front());
            order_.pop_front();
            auto it = table_.find(key);
            if (it == table_.end() || it->second.serial != serial) continue;   // replaced or invalidated meanwhile
            bytes_ -= it->second.cost;
            table_.erase(it);
            raw_tokens_.erase({key.record_id, key.section_idx});   // the raw count goes with the entry: both are re-made by a miss
        }
        while (!order_.empty()) {   // forget order entries whose table entry is gone
            auto it = table_.find(order_.front().second);
            if (it != table_.end() && it->second.serial == order_.front().first) break;
            order_.pop_front();
        }
    }

    Context& ctx_;
    size_t budget_;
    mutable std::shared_mutex table_mu_;
    std::mutex gpu_mu_, queue_mu_;
    std::vector<Pending*> queue_;
    std::map<Key, Entry> table_;
    std::map<std::pair<std::string, size_t>, RawCount> raw_tokens_;
    std::deque<std::pair<uint64_t, Key>> order_;
    size_t bytes_ = 0;
    uint64_t serial_ = 0;
    std::atomic<size_t> batches_{0};
};

}  // namespace triplets_amd
