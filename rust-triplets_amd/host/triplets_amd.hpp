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
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
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
    // The one preprocessor the GPU path implements; set by register_denoiser.
    std::optional<DenoiserConfig> denoiser;
    void register_denoiser(const DenoiserConfig& c) { denoiser = c; }
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
}  // namespace detail

// Host-side view of one batch result (owns the arrays).
class BatchResult {
public:
    BatchResult() { std::memset(&r_, 0, sizeof r_); }
    ~BatchResult() { trp_result_free(&r_); }
    BatchResult(const BatchResult&) = delete;
    BatchResult& operator=(const BatchResult&) = delete;
    BatchResult(BatchResult&& o) noexcept : r_(o.r_), weight_(o.weight_) { std::memset(&o.r_, 0, sizeof o.r_); }

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
            if (c.kind == 0 && c.tok_begin == 0 && c.tok_end == total && c.span == total) {
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
    BatchResult run_packed(const uint8_t* bytes, const uint64_t* offsets, size_t n, const ChunkingStrategy& st) {
        auto ov = detail::overlaps_of(st);
        trp_config c = detail::make_config(st, ov);
        BatchResult out;
        out.weight_ = st.summary_fallback_weight;
        check(trp_preprocess_batch(ctx_.raw(), bytes, offsets, n, &c, &out.r_));
        return out;
    }

private:
    Context& ctx_;
};

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

// Chunk table filled by one GPU batch and served per (record id, section index): what a sampler needs to
// call a GPU-backed chunker once per chunk selection (the reference calls ChunkingAlgorithm::materialize
// per selection, src/sampler/mod.rs:2510-2537, and re-tokenises every section when it refreshes its
// record cache, :1803-1836 - the place to call prefill).  token_count answers what
// record_has_long_anchor_or_context_section (:1765-1773) asks without tokenising again.
class ChunkCache {
public:
    struct SectionRef {
        std::string record_id;
        size_t section_idx;
        std::string_view text;
    };
    explicit ChunkCache(Context& ctx) : ctx_(ctx) {}

    void prefill(const ChunkingStrategy& st, const std::vector<SectionRef>& sections) {
        std::vector<std::string_view> texts;
        texts.reserve(sections.size());
        for (const auto& s : sections) texts.push_back(s.text);
        BatchPreprocessor bp(ctx_);
        BatchResult r = bp.run(texts, st);
        for (size_t i = 0; i < sections.size(); i++) {
            Entry e;
            e.chunks = r.chunks(i, sections[i].record_id, sections[i].section_idx);
            e.tokens = r.token_count(i);
            table_[{sections[i].record_id, sections[i].section_idx}] = std::move(e);
        }
    }
    // The chunks of one section; a section that was not prefilled is computed as a batch of one.
    std::vector<RecordChunk> materialize(const ChunkingStrategy& st, const std::string& record_id, size_t section_idx,
                                         std::string_view section_text) {
        auto it = table_.find({record_id, section_idx});
        if (it != table_.end()) return it->second.chunks;
        prefill(st, {SectionRef{record_id, section_idx, section_text}});
        return table_[{record_id, section_idx}].chunks;
    }
    std::optional<size_t> token_count(const std::string& record_id, size_t section_idx) const {
        auto it = table_.find({record_id, section_idx});
        if (it == table_.end()) return std::nullopt;
        return it->second.tokens;
    }
    size_t size() const { return table_.size(); }
    void clear() { table_.clear(); }

private:
    struct Entry {
        std::vector<RecordChunk> chunks;
        size_t tokens = 0;
    };
    Context& ctx_;
    std::map<std::pair<std::string, size_t>, Entry> table_;
};

}  // namespace triplets_amd
