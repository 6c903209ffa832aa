// Exercises the C++ host API on the chunker vector of the reference's own test
// (src/chunking/sliding_window.rs:155-172) and a denoiser vector (denoiser_preprocessor.rs:806).
// Exit status 0 = all checks passed, 2 = no GPU context could be created, 1 = a check failed.
#include <cstdio>

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
        auto c0 = cache.materialize(st, "r1", 0, "ignored: served from the table");
        EXPECT(c0.size() == 3 && c0[1].text == "four five six seven" && c0[0].record_id == "r1");
        auto c1 = cache.materialize(st, "r1", 1, "a b");
        EXPECT(c1.size() == 1 && c1[0].text == "a b" && c1[0].view.span == 2 && c1[0].section_idx == 1);
        EXPECT(cache.materialize(st, "r9", 0, "").empty());
        auto miss = cache.materialize(st, "r2", 4, "x y z");   // not prefilled: a batch of one
        EXPECT(miss.size() == 1 && miss[0].tokens_estimate == 3 && cache.size() == 4);
        std::puts("host_check: ok");
        return 0;
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == TRP_E_HIP ? 2 : 1;
    }
}
