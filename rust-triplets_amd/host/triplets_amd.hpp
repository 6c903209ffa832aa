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
#include <atomic>
#include <exception>
#include <map>
#include <mutex>
#include <optional>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <string_view>
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
        check(trp_preprocess_batch_select(ctx_.raw(), bytes.data(), offs.data(), sections.size(), &c, outputs, &out.r_, nullptr));
        return out;
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

// Chunk table filled by GPU batches and served per (record id, section index): what a sampler needs to call
// a GPU-backed chunker once per chunk selection.  The reference calls ChunkingAlgorithm::materialize per
// selection, concurrently from its worker threads (`Send + Sync`, src/chunking/algorithm.rs:5,
// src/sampler/mod.rs:2510-2537), and tokenises the RAW text of every section when it refreshes its record
// cache (record_has_long_anchor_or_context_section, src/sampler/mod.rs:1803-1812) - the place to call prefill.
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
    explicit ChunkCache(Context& ctx) : ctx_(ctx) {}

    // One batch for the chunks (with the strategy's denoiser, if any) and, when a denoiser is registered, one
    // pass-through batch over the same sections for the raw token counts.
    void prefill(const ChunkingStrategy& st, const std::vector<SectionRef>& sections) {
        std::lock_guard<std::mutex> gpu(gpu_mu_);
        compute(st, sections);
    }
    // The chunks of one section (SlidingWindowChunker::materialize, sliding_window.rs:11-117).
    std::vector<RecordChunk> materialize(const ChunkingStrategy& st, const std::string& record_id, size_t section_idx,
                                         std::string_view section_text) {
        const Key key{strategy_hash(st), record_id, section_idx};
        {
            std::shared_lock<std::shared_mutex> rd(table_mu_);
            auto it = table_.find(key);
            if (it != table_.end()) return it->second.chunks;
        }
        // miss: join the queue; the first thread to get the GPU computes everything queued for this strategy
        Pending me{&st, SectionRef{record_id, section_idx, section_text}, key, false, nullptr};
        {
            std::lock_guard<std::mutex> q(queue_mu_);
            queue_.push_back(&me);
        }
        {
            std::lock_guard<std::mutex> gpu(gpu_mu_);
            if (!me.done) drain();
        }
        if (me.error) std::rethrow_exception(me.error);
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        return table_.at(key).chunks;
    }
    // WhitespaceTokenizer.token_count(&section.text) of the RAW section text - what the reference's sampler
    // compares with max_window_tokens (src/sampler/mod.rs:1810) - for a section that was prefilled.
    std::optional<size_t> token_count(const std::string& record_id, size_t section_idx) const {
        std::shared_lock<std::shared_mutex> rd(table_mu_);
        auto it = raw_tokens_.find({record_id, section_idx});
        if (it == raw_tokens_.end()) return std::nullopt;
        return it->second;
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
    void clear() {
        std::unique_lock<std::shared_mutex> wr(table_mu_);
        table_.clear();
        raw_tokens_.clear();
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
    };
    struct Pending {
        const ChunkingStrategy* st;
        SectionRef ref;
        Key key;
        bool done;
        std::exception_ptr error;
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
            try {
                compute(*same.front()->st, refs);
            } catch (...) {
                err = std::current_exception();
            }
            for (Pending* p : same) { p->error = err; p->done = true; }
            work.swap(rest);
        }
    }
    // gpu_mu_ held.
    void compute(const ChunkingStrategy& st, const std::vector<SectionRef>& sections) {
        if (sections.empty()) return;
        std::vector<std::string_view> texts;
        texts.reserve(sections.size());
        for (const auto& s : sections) texts.push_back(s.text);
        BatchPreprocessor bp(ctx_);
        BatchResult r = bp.run(texts, st);
        batches_++;
        std::vector<size_t> raw(sections.size());
        if (st.denoiser && st.denoiser->enabled) {
            ChunkingStrategy plain = st;
            plain.denoiser.reset();
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
        }
        std::unique_lock<std::shared_mutex> wr(table_mu_);
        for (size_t i = 0; i < sections.size(); i++) {
            table_[Key{h, sections[i].record_id, sections[i].section_idx}] = std::move(made[i]);
            raw_tokens_[{sections[i].record_id, sections[i].section_idx}] = raw[i];
        }
    }

    Context& ctx_;
    mutable std::shared_mutex table_mu_;
    std::mutex gpu_mu_, queue_mu_;
    std::vector<Pending*> queue_;
    std::map<Key, Entry> table_;
    std::map<std::pair<std::string, size_t>, size_t> raw_tokens_;
    std::atomic<size_t> batches_{0};
};

}  // namespace triplets_amd
