// Exercises the C++ host API on the chunker vector of the reference's own test
// (src/chunking/sliding_window.rs:155-172) and a denoiser vector (denoiser_preprocessor.rs:806).
// Exit status 0 = all checks passed, 2 = no GPU context could be created, 1 = a check failed.
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "triplets_amd.hpp"

using namespace triplets_amd;

#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main() {
    try {
        Context ctx(0);
        ChunkingStrategy st;
        st.max_window_tokens = 4;
        st.overlap_tokens = {1};
        st.summary_fallback_tokens = 2;
        SlidingWindowChunker chunker(ctx);
        auto chunks = chunker.materialize(st, "r1", 0, "one two three four five six seven");
        EXPECT(chunks.size() == 3);
        EXPECT(chunks[0].text == "one two three four" && chunks[0].view.kind == ChunkView::Window);
        EXPECT(chunks[1].text == "four five six seven" && chunks[1].view.index == 1 && chunks[1].view.overlap == 1);
        EXPECT(chunks[2].view.kind == ChunkView::SummaryFallback && chunks[2].text == "one two");
        EXPECT(chunks[2].view.strategy == "head" && chunks[2].tokens_estimate == 2);

        WhitespaceTokenizer tok(ctx);
        auto toks = tok.tokenize("  alpha\tbeta\xE3\x80\x80gamma \n");
        EXPECT(toks.size() == 3 && toks[0] == "alpha" && toks[1] == "beta" && toks[2] == "gamma");
        EXPECT(tok.token_count("") == 0);

        DenoiserConfig dc;
        dc.enabled = true;
        DenoiserPreprocessor den(ctx, dc);
        auto kept = den.process("plain prose stays\n12 34 56 78\n| a | b |\n|---|---|");
        EXPECT(kept.has_value() && *kept == "plain prose stays\na b");
        EXPECT(!den.process("123 456\n---").has_value());

        ChunkingStrategy sd;
        sd.register_denoiser(dc);
        auto none = chunker.materialize(sd, "r2", 1, "1234 5678\n99");
        EXPECT(none.empty());
        EXPECT(normalize_inline_whitespace(ctx, "Alpha\r\n\r\n  Beta\tGamma") == "Alpha Beta Gamma");
        EXPECT(normalize_inline_whitespace(ctx, " \n ") == "");

        ChunkCache cache(ctx);
        cache.prefill(st, {{"r1", 0, "one two three four five six seven"}, {"r1", 1, "a b"}, {"r9", 0, ""}});
        EXPECT(cache.size() == 3 && cache.token_count("r1", 0).value() == 7 && cache.token_count("r9", 0).value() == 0);
        {
            // token_count is the count of the RAW section text (src/sampler/mod.rs:1810), also when a denoiser
            // shortens the text the windows are cut from; a changed strategy never sees another one's chunks
            ChunkingStrategy sdn = st;
            sdn.register_denoiser(dc);
            const char* noisy = "alpha beta 1 2\n12 34 56 78\n| a | b |";
            cache.prefill(sdn, {{"rd", 0, noisy}});
            EXPECT(cache.token_count("rd", 0).value() == 13);
            EXPECT(cache.preprocessed_token_count(sdn, "rd", 0).value() == 6);     // "alpha beta 1 2\na b"
            EXPECT(!cache.preprocessed_token_count(st, "rd", 0).has_value());
            auto with = cache.materialize(sdn, "rd", 0, noisy), without = cache.materialize(st, "rd", 0, noisy);
            EXPECT(with.size() == 3 && with[0].text == "alpha beta 1 2" && with[1].text == "2 a b");
            EXPECT(without.size() == 5 && without[0].text == "alpha beta 1 2");
            EXPECT(cache.preprocessed_token_count(st, "rd", 0).value() == 13);
        }
        {
            // many threads, as the reference's rayon workers call materialize: every answer equals the one a
            // fresh single-threaded chunker gives, and concurrent misses share batches
            const int n_threads = 16, n_sections = 400;
            std::vector<std::string> texts;
            for (int i = 0; i < n_sections; i++) {
                std::string t;
                for (int k = 0; k < 3 + (i * 7) % 23; k++) t += "w" + std::to_string(i) + "_" + std::to_string(k) + (k % 5 == 4 ? "\n" : " ");
                texts.push_back(t);
            }
            ChunkCache shared(ctx);
            std::vector<int> bad(n_threads, 0);
            std::vector<std::thread> pool;
            for (int t = 0; t < n_threads; t++)
                pool.emplace_back([&, t]() {
                    try {
                        for (int rep = 0; rep < 3; rep++)
                            for (int i = (t * 37) % n_sections, n = 0; n < n_sections; n++, i = (i + 1) % n_sections) {
                                auto got = shared.materialize(st, "rec" + std::to_string(i), (size_t)i, texts[i]);
                                const size_t T = 3 + (i * 7) % 23;
                                const size_t want = T <= 4 ? 1 : 1 + (T - 4 + 2) / 3 + 1;
                                if (got.size() != want || got[0].record_id != "rec" + std::to_string(i) || got[0].section_idx != (size_t)i ||
                                    got[0].text.rfind("w" + std::to_string(i) + "_0", 0) != 0)
                                    bad[t]++;
                                auto tc = shared.token_count("rec" + std::to_string(i), (size_t)i);
                                if (!tc || *tc != T) bad[t]++;
                            }
                    } catch (...) {
                        bad[t] = 1000000;
                    }
                });
            for (auto& th : pool) th.join();
            for (int t = 0; t < n_threads; t++) EXPECT(bad[t] == 0);
            EXPECT(shared.size() == (size_t)n_sections);
            EXPECT(shared.gpu_batches() < (size_t)n_sections);   // misses under contention were batched
            SlidingWindowChunker fresh(ctx);
            for (int i = 0; i < n_sections; i += 57) {
                auto a = shared.materialize(st, "rec" + std::to_string(i), (size_t)i, texts[i]);
                auto b = fresh.materialize(st, "rec" + std::to_string(i), (size_t)i, texts[i]);
                EXPECT(a.size() == b.size());
                for (size_t k = 0; k < a.size(); k++)
                    EXPECT(a[k].text == b[k].text && a[k].tokens_estimate == b[k].tokens_estimate && a[k].view.kind == b[k].view.kind);
            }
        }
        auto c0 = cache.materialize(st, "r1", 0, "one two three four five six seven");
        EXPECT(c0.size() == 3 && c0[1].text == "four five six seven" && c0[0].record_id == "r1");
        {
            // the result is a function of the text handed in (sliding_window.rs:18): the same id with another text
            // is recomputed, never served from the entry of the old text
            const size_t before = cache.gpu_batches();
            auto again = cache.materialize(st, "r1", 0, "one two three four five six seven");
            EXPECT(cache.gpu_batches() == before && again.size() == 3);                       // a hit
            auto changed = cache.materialize(st, "r1", 0, "alpha beta");
            EXPECT(cache.gpu_batches() == before + 1);                                        // a miss
            EXPECT(changed.size() == 1 && changed[0].text == "alpha beta" && changed[0].tokens_estimate == 2);
            EXPECT(cache.token_count("r1", 0).value() == 2);
            auto changed2 = cache.materialize(st, "r1", 0, "alpha beta");
            EXPECT(cache.gpu_batches() == before + 1 && changed2.size() == 1 && changed2[0].text == "alpha beta");
            // same length, other bytes
            auto changed3 = cache.materialize(st, "r1", 0, "alpha bet4");
            EXPECT(cache.gpu_batches() == before + 2 && changed3[0].text == "alpha bet4");
            EXPECT(cache.invalidate("r1") == 2 && !cache.token_count("r1", 0).has_value() && !cache.token_count("r1", 1).has_value());
            cache.prefill(st, {{"r1", 0, "one two three four five six seven"}, {"r1", 1, "a b"}});
        }
        {
            // a byte budget: ten times more sections than fit go through, the table stays under its budget and the
            // entries that were inserted last are the ones still there
            ChunkCache small(ctx, 64 * 1024);
            std::vector<std::string> ids, texts;
            for (int i = 0; i < 4000; i++) {
                ids.push_back("s" + std::to_string(i));
                std::string t;
                for (int k = 0; k < 12; k++) t += "tok" + std::to_string(i) + "x" + std::to_string(k) + " ";
                texts.push_back(t);
            }
            for (int base = 0; base < 4000; base += 500) {
                std::vector<ChunkCache::SectionRef> refs;
                for (int i = base; i < base + 500; i++) refs.push_back({ids[i], 0, texts[i]});
                small.prefill(st, refs);
                EXPECT(small.bytes() <= small.budget());
            }
            EXPECT(small.size() > 10 && small.size() < 400);
            EXPECT(small.preprocessed_token_count(st, "s3999", 0).has_value() && !small.preprocessed_token_count(st, "s0", 0).has_value());
            EXPECT(!small.token_count("s0", 0).has_value() && small.token_count("s3999", 0).value() == 12);
            auto back = small.materialize(st, "s0", 0, texts[0]);   // evicted: computed again, correct
            EXPECT(back.size() == 5 && back[0].text.rfind("tok0x0", 0) == 0 && small.bytes() <= small.budget());
        }
        {
            // BM25 document texts, the query cap and decorate_chunk, with the reference's own cases
            // (src/sampler/tests.rs:10700, :10637, :566, :600)
            std::vector<Bm25Record> recs = {
                {"bm25_text_zero", {{std::string_view("Heading"), "anchor body text with many tokens here and there and everywhere"},
                                    {std::nullopt, "context section body words more and more content"}}},
                {"only-blank", {{std::string_view("  "), " \t "}}},
                {"no-sections", {}}};
            auto all = record_bm25_texts(ctx, recs, 0);
            EXPECT(all.size() == 3 && all[0] == "Heading\nanchor body text with many tokens here and there and everywhere\n"
                                                "context section body words more and more content\n");
            EXPECT(all[1] == "only-blank" && all[2] == "no-sections");
            auto five = record_bm25_texts(ctx, recs, 5, "\r\n");
            EXPECT(five[0] == "Heading anchor body text with" && five[1] == "only-blank");
            std::string q;
            for (int i = 0; i < 74; i++) q += (i ? " word" : "word") + std::to_string(i);
            auto capped = cap_tokens(ctx, {q, " a  b ", ""}, 64);
            EXPECT(tok.token_count(capped[0]) == 64 && capped[0].rfind("word0 word1 ", 0) == 0 && capped[1] == " a  b " && capped[2].empty());
            auto dec = decorate_chunks(ctx, {{"one two three", std::string_view("meta: long=a b c d e f g h i")},
                                             {"one two three", std::string_view("meta: source=unit")},
                                             {"one two three", std::nullopt},
                                             {"one  two", std::string_view("p  q")},
                                             {"one two three", std::string_view("a  b c d e")}}, 5);
            EXPECT(dec[0].tokens_estimate == 5 && dec[0].text == "meta: long=a b c d");                 // prefix alone fills the window
            EXPECT(dec[2].text == "one two three" && dec[2].tokens_estimate == 3);
            EXPECT(dec[3].text == "p  q\none  two" && dec[3].tokens_estimate == 4);                    // fits: both verbatim
            EXPECT(dec[4].text == "a b c d e" && dec[4].tokens_estimate == 5);                          // exactly the window: joined
            auto dec4 = decorate_chunks(ctx, {{"one two three", std::string_view("meta: source=unit")}}, 4);
            EXPECT(dec4[0].text == "meta: source=unit\none two" && dec4[0].tokens_estimate == 4);
            auto bodies = chunk_bodies(ctx, {"one two three four five six seven", "a b"}, st, {0, 1, 2, 3}, {2, 100, 1, 1});
            EXPECT(bodies.size() == 4 && bodies[0] == "one two" && bodies[1] == "four five six seven" && bodies[2] == "one" && bodies[3] == "a");
        }
        {
            // one batch over several contexts of this process (here: the one GPU listed three times)
            ShardedPreprocessor sp({0, 0, 0});
            std::vector<std::string> texts;
            for (int i = 0; i < 300; i++) {
                std::string t;
                for (int k = 0; k < 1 + (i * 5) % 17; k++) t += "w" + std::to_string(i) + "_" + std::to_string(k) + " ";
                texts.push_back(t);
            }
            std::vector<std::string_view> views(texts.begin(), texts.end());
            ShardedResult sr = sp.run(views, st);
            BatchPreprocessor bp(ctx);
            BatchResult whole = bp.run(views, st);
            EXPECT(sr.records() == 300 && sr.shards() == 3);
            for (size_t r = 0; r < 300; r++) {
                EXPECT(sr.token_count(r) == whole.token_count(r) && sr.text(r) == whole.text(r) && sr.dropped(r) == whole.dropped(r));
                auto a = sr.chunks(r, "id", r), b = whole.chunks(r, "id", r);
                EXPECT(a.size() == b.size());
                for (size_t k = 0; k < a.size(); k++) EXPECT(a[k].text == b[k].text && a[k].view.index == b[k].view.index);
            }
            EXPECT(ShardedPreprocessor::all_devices().size() >= 1);
        }
        auto c1 = cache.materialize(st, "r1", 1, "a b");
        EXPECT(c1.size() == 1 && c1[0].text == "a b" && c1[0].view.span == 2 && c1[0].section_idx == 1);
        EXPECT(cache.materialize(st, "r9", 0, "").empty());
        auto miss = cache.materialize(st, "r2", 4, "x y z");   // not prefilled: a batch of one
        EXPECT(miss.size() == 1 && miss[0].tokens_estimate == 3 && cache.size() == 6);
        // a lookup that follows the miss is a hit
        const size_t nb = cache.gpu_batches();
        EXPECT(cache.materialize(st, "r2", 4, "x y z").size() == 1 && cache.gpu_batches() == nb);
        {
            // Two registered denoisers (config.rs:105-113): the text is folded through them in order and the
            // section is gone as soon as one returns None (sliding_window.rs:20-34).  The first keeps the line
            // "ab 12 34" (two of six letters and digits are letters: ratio 0.67 is not above 0.9) and the second, with
            // the default ratio, strips its digit tokens; a section the first one empties stays dropped; a disabled
            // second one changes nothing.
            ChunkingStrategy two;
            two.register_denoiser({.enabled = true, .max_digit_ratio = 0.9f});
            two.register_denoiser({.enabled = true});
            BatchPreprocessor bp(ctx);
            auto r2 = bp.run({"words stay\nab 12 34", "123 456\n---", "plain"}, two);
            EXPECT(r2.text(0) == "words stay\nab" && !r2.dropped(0));
            EXPECT(r2.dropped(1) && r2.chunks(1, "x", 1).empty());
            EXPECT(r2.text(2) == "plain");
            ChunkingStrategy one;
            one.register_denoiser({.enabled = true, .max_digit_ratio = 0.9f});
            one.register_denoiser({.enabled = false});
            auto r1 = bp.run({"words stay\nab 12 34"}, one);
            EXPECT(r1.text(0) == "words stay\nab 12 34");
            ChunkCache folded(ctx);
            auto fc = folded.materialize(two, "f", 0, "words stay\nab 12 34");
            EXPECT(fc.size() == 1 && fc[0].text == "words stay\nab" && folded.token_count("f", 0).value() == 5);
        }
        std::puts("host_check: ok");
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == TRP_E_HIP ? 2 : 1;
    }
}
