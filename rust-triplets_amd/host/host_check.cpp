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
        auto c0 = cache.materialize(st, "r1", 0, "ignored: served from the table");
        EXPECT(c0.size() == 3 && c0[1].text == "four five six seven" && c0[0].record_id == "r1");
        auto c1 = cache.materialize(st, "r1", 1, "a b");
        EXPECT(c1.size() == 1 && c1[0].text == "a b" && c1[0].view.span == 2 && c1[0].section_idx == 1);
        EXPECT(cache.materialize(st, "r9", 0, "").empty());
        auto miss = cache.materialize(st, "r2", 4, "x y z");   // not prefilled: a batch of one
        EXPECT(miss.size() == 1 && miss[0].tokens_estimate == 3 && cache.size() == 6);
        std::puts("host_check: ok");
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == TRP_E_HIP ? 2 : 1;
    }
}
