/*
 * CPU oracle for the denoise -> tokenise -> chunk path of the `triplets` crate.
 *
 * THIS IS TEST INFRASTRUCTURE.  It restates the reference algorithm in plain C so
 * that the HIP implementation can be checked against it.  Nothing in the shipped
 * library links, loads or calls it; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * Parity status: pinned by the reference's own known-answer tests (transcribed as
 * data in tests/golden/), except for CR / CRLF handling, which goes through the
 * third-party crate `line-ending` 1.5.1 (not vendored in the reference tree); its
 * documented behaviour ("\r\n" -> "\n", then "\r" -> "\n") is restated here and is
 * "parity unpinned" because no reference test feeds a '\r' to the denoiser.
 *
 * Reference files followed (paths relative to the reference checkout):
 *   src/preprocessor/backends/denoiser_preprocessor.rs:21-268   denoise_text + helpers
 *   src/tokenizer.rs:55-65                                       WhitespaceTokenizer
 *   src/chunking/sliding_window.rs:11-117                        SlidingWindowChunker
 *   src/config.rs:34-42, 60-71                                   parameter defaults
 *   src/sampler/mod.rs:1484-1506                                 decorate_chunk (text after the prefix)
 *   src/sampler/backends/bm25_backend.rs:219-225, 503-532        BM25 query cap, record_bm25_text
 */
#ifndef TRP_ORACLE_H
#define TRP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors DenoiserConfig + the chunking part of ChunkingStrategy (src/config.rs:12-58). */
typedef struct {
    uint8_t denoise_enabled;
    uint8_t strip_markdown;
    float max_digit_ratio;
    uint32_t max_window_tokens;
    const uint32_t* overlaps;
    uint32_t n_overlaps;
    uint32_t summary_fallback_tokens;
} orc_config;

/* One chunk view (src/data.rs:157-174) without the owned text: token range is
 * relative to the first token of its record. */
typedef struct {
    uint32_t record;
    uint32_t kind; /* 0 = Window, 1 = SummaryFallback */
    uint32_t index;
    uint32_t overlap;
    uint32_t span;
    uint32_t tok_begin;
    uint32_t tok_end;
    uint32_t reserved;
} orc_chunk;

typedef struct {
    uint8_t* denoised;          /* concatenated preprocessed text of all records */
    uint64_t* denoised_offsets; /* R+1 */
    uint8_t* dropped;           /* R: 1 when the preprocessor returned None */
    uint32_t* tok_start;        /* byte offset into `denoised` */
    uint32_t* tok_len;          /* byte length */
    uint64_t* tok_offsets;      /* R+1 */
    orc_chunk* chunks;
    uint64_t* chunk_offsets; /* R+1 */
    uint64_t n_records, n_denoised, n_tokens, n_chunks;
} orc_result;

/* denoise_text (denoiser_preprocessor.rs:221-268).  Returns 1 for Some(text) and
 * stores a malloc'd buffer in *out (may be NULL when *out_len == 0), 0 for None. */
int orc_denoise(const uint8_t* text, size_t len, int enabled, float max_digit_ratio,
                int strip_markdown, uint8_t** out, size_t* out_len);

/* WhitespaceTokenizer::tokenize (tokenizer.rs:57-59).  Writes up to `cap` (start,len)
 * pairs and returns the total number of tokens. */
size_t orc_tokenize(const uint8_t* text, size_t len, uint32_t* starts, uint32_t* lens, size_t cap);

/* Chunk views that SlidingWindowChunker::materialize (sliding_window.rs:40-115) emits for
 * a section with `n_tokens` tokens.  Writes up to `cap` and returns the total count. */
size_t orc_chunk_views(uint64_t n_tokens, const orc_config* cfg, uint32_t record, orc_chunk* out,
                       size_t cap);

/* Whole path over a packed batch (bytes + offsets[R+1]).  Returns 0 on success. */
int orc_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
              const orc_config* cfg, orc_result* out);
/* Same, records split over `n_threads` host threads by a byte-balanced static partition. */
int orc_batch_mt(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                 const orc_config* cfg, int n_threads, orc_result* out);
void orc_result_free(orc_result* r);

/* Chunk text as the reference renders it (sliding_window.rs:44,67,95-100).  Returns the
 * byte length; writes at most cap bytes. */
size_t orc_chunk_text(const orc_result* r, uint64_t chunk_idx, uint8_t* out, size_t cap);

/* Character predicates (for table tests). */
int orc_is_alphabetic(uint32_t cp);
int orc_is_whitespace(uint32_t cp);

/* Test hooks for the reference's private helpers. `out` must hold `len` bytes. */
int orc_is_markdown_table_line(const uint8_t* line, size_t len);
int orc_is_markdown_table_separator(const uint8_t* line, size_t len);
size_t orc_strip_table_pipes(const uint8_t* line, size_t len, uint8_t* out);

/* normalize_inline_whitespace (src/utils.rs:18-33): the File/CSV sources apply it before
 * the path; used to restate the example_dataset configuration. */
size_t orc_normalize_inline_whitespace(const uint8_t* text, size_t len, uint8_t* out);

/* The steps right after the path (they re-use its tokenizer).  Each returns the byte length and stores a
 * malloc'd buffer in *out (NULL when the length is 0).
 * orc_decorate_chunk: body of decorate_chunk for a sampled prefix (src/sampler/mod.rs:1484-1506).
 * orc_record_bm25_text: src/sampler/backends/bm25_backend.rs:503-532.
 * orc_cap_tokens: the BM25 query cap (bm25_backend.rs:219-225). */
size_t orc_decorate_chunk(const uint8_t* text, size_t text_len, const uint8_t* prefix, size_t prefix_len,
                          uint32_t max_window_tokens, const uint8_t* newline, size_t newline_len,
                          uint8_t** out, uint64_t* tokens_estimate);
size_t orc_record_bm25_text(const uint8_t* const* headings, const size_t* heading_len, const uint8_t* has_heading,
                            const uint8_t* const* texts, const size_t* text_len, size_t n_sections,
                            const uint8_t* record_id, size_t id_len, size_t max_tokens,
                            const uint8_t* newline, size_t newline_len, uint8_t** out);
size_t orc_cap_tokens(const uint8_t* text, size_t len, size_t limit, uint8_t** out);

#ifdef __cplusplus
}
#endif
#endif
