/*
 * triplets_amd - C ABI of the MI355X (gfx950) implementation of the `triplets` crate's per-record
 * text preprocessing path: OCR/markdown denoising -> whitespace tokenisation -> sliding-window
 * chunking, over MANY records packed into one contiguous byte buffer plus an offsets array.
 *
 * Nothing like this batched entry point exists in the reference (it processes one section per
 * call, on the CPU); each declaration below cites the reference interface whose semantics it
 * reproduces bit for bit.  Paths are relative to the reference checkout.
 *
 * Conventions
 *   - every function returns an int status: TRP_OK (0) or a negative TRP_E_* code; the text of the
 *     last error of the calling thread is available from trp_last_error().
 *   - text must be valid UTF-8 (the reference takes Rust `&str`); record boundaries never split a
 *     character.  Malformed input never faults but its classification is unspecified.
 *   - a batch holds at most TRP_MAX_BATCH_BYTES bytes and 2^32-2 records; callers split larger
 *     corpora by record range (records are independent).
 *   - a trp_ctx is bound to one GPU and one HIP stream and must not be used from two threads at the
 *     same time; create one context per thread (they are cheap and independent).
 */
#ifndef TRIPLETS_AMD_H
#define TRIPLETS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRP_OK 0
#define TRP_E_ARG (-1)      /* invalid argument */
#define TRP_E_HIP (-2)      /* a HIP runtime call failed (no GPU, out of memory, ...) */
#define TRP_E_TOO_LARGE (-3) /* batch exceeds TRP_MAX_BATCH_BYTES or the record limit */

#define TRP_MAX_BATCH_BYTES 0xFFFF0000ull
#define TRP_MAX_OVERLAPS 16

typedef struct trp_ctx trp_ctx;

/* DenoiserConfig (src/config.rs:12-42) and the chunking fields of ChunkingStrategy
 * (src/config.rs:45-71).  Defaults of the reference: enabled=false, max_digit_ratio=0.35,
 * strip_markdown=true, max_window_tokens=1024, overlap_tokens=[64], summary_fallback_tokens=512. */
typedef struct {
    uint8_t denoise_enabled;
    uint8_t strip_markdown;
    float max_digit_ratio;
    uint32_t max_window_tokens;
    const uint32_t* overlaps; /* host pointer, n_overlaps entries (<= TRP_MAX_OVERLAPS) */
    uint32_t n_overlaps;
    uint32_t summary_fallback_tokens;
} trp_config;

/* One RecordChunk view (src/data.rs:130-174, built at src/chunking/sliding_window.rs:45-111) minus
 * its owned strings.  kind 0 = ChunkView::Window{index,overlap,span}, kind 1 =
 * ChunkView::SummaryFallback{strategy:"head"}.  The chunk covers tokens [tok_begin, tok_end) of its
 * record (relative to the record's first token); tokens_estimate = tok_end - tok_begin. */
typedef struct {
    uint32_t record;
    uint32_t kind;
    uint32_t index;
    uint32_t overlap;
    uint32_t span;
    uint32_t tok_begin;
    uint32_t tok_end;
    uint32_t reserved;
} trp_chunk;

/* Result of one batch.  All arrays are host memory for trp_preprocess_batch and device memory for
 * trp_preprocess_batch_device.
 *   denoised / denoised_offsets[R+1]: the preprocessed text of record r is
 *       denoised[denoised_offsets[r] .. denoised_offsets[r+1]).  With the denoiser disabled the text
 *       is the input itself: `denoised` then aliases the caller's input buffer (device variant) or
 *       is a copy of it (host variant) and denoised_offsets equals the input offsets.
 *   dropped[R]: 1 where the preprocessor returned None (the section yields no chunks).
 *   tok_start / tok_len, tok_offsets[R+1]: token k of record r is the byte range
 *       denoised[tok_start[tok_offsets[r]+k] .. +tok_len[...]) - what
 *       WhitespaceTokenizer::tokenize returns as &str slices.
 *   chunks, chunk_offsets[R+1]: the Vec<RecordChunk> of record r, in the reference's order. */
typedef struct {
    uint8_t* denoised;
    uint64_t* denoised_offsets;
    uint8_t* dropped;
    uint32_t* tok_start;
    uint32_t* tok_len;
    uint64_t* tok_offsets;
    trp_chunk* chunks;
    uint64_t* chunk_offsets;
    uint64_t n_records, n_denoised, n_tokens, n_chunks;
} trp_result;

/* Duration of one kernel of the last profiled call (see trp_set_profiling). */
typedef struct {
    char name[32];
    float ms;
} trp_kernel_time;

/* Create / destroy a context on GPU `device` (a HIP device ordinal).  Fails with TRP_E_HIP when no
 * usable gfx950 device is present: there is no CPU fallback. */
int trp_create(int device, trp_ctx** out);
void trp_destroy(trp_ctx* ctx);
/* Number of GPUs the HIP runtime shows to this process (0 when there is none or the runtime fails).  For hosts
 * that spread a batch over all GPUs by record range, one context and one thread per device: records are
 * independent (ChunkingAlgorithm::materialize takes one section, src/chunking/algorithm.rs:7-13), so the shards
 * share nothing. */
int trp_device_count(void);
const char* trp_last_error(void);
/* The hipStream_t the context launches on (as void*), for callers that order their own work. */
void* trp_stream(trp_ctx* ctx);
/* Ordering against the caller's own GPU work.  The reference needs none of this: its path is a synchronous
 * CPU call (SlidingWindowChunker::materialize, src/chunking/sliding_window.rs:11); a device-resident
 * replacement has to say on which stream it runs.
 * trp_create_on_stream: like trp_create, but every kernel and copy of the context is enqueued on the
 *   caller's `hip_stream` (a hipStream_t; NULL is the legacy default stream), so device inputs produced on
 *   that stream need no synchronisation before a call, and later work on it sees the results.  The stream
 *   stays the caller's: it must outlive the context and is not destroyed by trp_destroy.
 * trp_wait_event: makes all work the context enqueues from now on wait for `hip_event` (a hipEvent_t the
 *   caller recorded on the stream that produces the next inputs); the host does not block. */
int trp_create_on_stream(int device, void* hip_stream, trp_ctx** out);
int trp_wait_event(trp_ctx* ctx, void* hip_event);

/* Batched path over host memory: copies the batch to the GPU, runs the kernels and copies every
 * result array back into buffers owned by the library (release them with trp_result_free).
 * Replaces, for all records at once: the preprocessor fold + tokenizer + window loop of
 * SlidingWindowChunker::materialize (src/chunking/sliding_window.rs:11-117), i.e. per record
 * DenoiserPreprocessor::process / denoise_text (src/preprocessor/backends/denoiser_preprocessor.rs:
 * 221-268, 303-306), WhitespaceTokenizer::tokenize (src/tokenizer.rs:57-59) and the window and
 * summary-fallback arithmetic (sliding_window.rs:40-115).
 * offsets has n_records+1 entries, offsets[0] = 0, non-decreasing; record r is
 * bytes[offsets[r] .. offsets[r+1]). */
int trp_preprocess_batch(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets,
                         uint64_t n_records, const trp_config* cfg, trp_result* out);
void trp_result_free(trp_result* r);
/* The same call for consumers that do not need every array back: `outputs` selects what is copied to the
 * host (the link from the GPU is the limit of this entry point: all arrays together are about 2.4 bytes per
 * input byte).  The per-record arrays (the three offset arrays, dropped[]) and the totals always come back;
 * an array that was not selected is NULL in `out`.  TRP_OUT_CHUNK_TEXT renders every chunk's text on the GPU
 * (RecordChunk::text, sliding_window.rs:44, :67, :95-100) and returns it in `chunk_text`: what a sampler
 * that serves ChunkingAlgorithm::materialize from a table needs is chunks + chunk text, without the
 * preprocessed text and the token table.  Release with trp_result_free and trp_chunk_text_free. */
#define TRP_OUT_TEXT 1u       /* denoised */
#define TRP_OUT_TOKENS 2u     /* tok_start, tok_len */
#define TRP_OUT_CHUNKS 4u     /* chunks */
#define TRP_OUT_CHUNK_TEXT 8u /* chunk_text (must be non-NULL) */
#define TRP_OUT_ALL 7u        /* what trp_preprocess_batch returns */
typedef struct {
    uint8_t* text;     /* text of all chunks, concatenated in chunk order */
    uint64_t* offsets; /* n_chunks + 1 */
    uint64_t n_bytes;
} trp_chunk_text;
int trp_preprocess_batch_select(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets,
                                uint64_t n_records, const trp_config* cfg, uint32_t outputs, trp_result* out,
                                trp_chunk_text* chunk_text);
void trp_chunk_text_free(trp_chunk_text* t);
/* Large host result arrays are page-locked blocks that trp_result_free / trp_free return to a
 * process-wide pool for the next call (up to 16 GiB stays pooled).  trp_trim_host_pool unpins and frees
 * the blocks that are not in use.  No counterpart in the reference (its results are heap Strings). */
void trp_trim_host_pool(void);

/* A strategy with more than one registered preprocessor (ChunkingStrategy::register_preprocessor,
 * src/config.rs:105-113): SlidingWindowChunker::materialize folds the section text through them in registration
 * order and returns no chunks as soon as one returns None (src/chunking/sliding_window.rs:20-34).  The reference
 * ships one TextPreprocessor, the denoiser, so the fold taken over here is a list of denoisers: `cfg`'s own
 * first, then `more[0 .. n_more)`; a disabled one returns its text unchanged (denoiser_preprocessor.rs:222-224).
 * The text of each stage is the input of the next, on the GPU; `out` describes the last stage (denoised = the
 * fully folded text, dropped = some stage returned None), tokenised and chunked with `cfg`'s chunking fields.
 * `outputs` / `chunk_text` as in trp_preprocess_batch_select (TRP_OUT_ALL and NULL: every array, no chunk text).
 * Otherwise as trp_preprocess_batch / trp_preprocess_batch_device. */
typedef struct {
    uint8_t denoise_enabled;
    uint8_t strip_markdown;
    float max_digit_ratio;
} trp_denoiser;
int trp_preprocess_batch_fold(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                              const trp_config* cfg, const trp_denoiser* more, uint32_t n_more, uint32_t outputs,
                              trp_result* out, trp_chunk_text* chunk_text);
int trp_preprocess_batch_device_fold(trp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                     uint64_t n_records, uint64_t n_bytes, const trp_config* cfg,
                                     const trp_denoiser* more, uint32_t n_more, trp_result* out);

/* Same path with the batch already resident in GPU memory (16-byte aligned `d_bytes`, device
 * `d_offsets`).  The result arrays are device pointers into buffers owned by the context; they stay
 * valid until the next call on this context.  The call returns after the kernels have completed
 * (the totals in `out` are read back).  The kernels run on the context's stream (trp_stream): inputs
 * produced on another stream must be ordered before the call, either by creating the context on the
 * producer's stream (trp_create_on_stream) or with an event (trp_wait_event); a context made by
 * trp_create runs on a private non-blocking stream that waits for nothing by itself.
 * The offsets (same contract as above, with offsets[n_records] = n_bytes) are checked by the first kernel, as
 * the host cannot see them: a batch whose offsets break the contract fails with TRP_E_ARG once its kernels
 * have run (its result arrays hold nothing of use); no kernel reads or writes outside its arrays because of
 * them. */
int trp_preprocess_batch_device(trp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                uint64_t n_records, uint64_t n_bytes, const trp_config* cfg,
                                trp_result* out);

/* The same call in two halves, for callers that keep several batches in flight the way the reference's
 * BatchPrefetcher produces batches ahead of their consumer (src/sampler/mod.rs:159-248):
 * trp_submit_batch_device enqueues the whole path on the context's stream and returns at once;
 * trp_wait_batch waits for it and fills `out` exactly as trp_preprocess_batch_device does.  One batch
 * per context may be in flight; contexts run concurrently with each other (one stream each).  The input
 * buffers must stay untouched until trp_wait_batch returns. */
int trp_submit_batch_device(trp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                            uint64_t n_records, uint64_t n_bytes, const trp_config* cfg);
int trp_wait_batch(trp_ctx* ctx, trp_result* out);

/* trp_preprocess_batch_device for callers that cannot allocate GPU memory at the time of the call (tight
 * serving loops, harnesses that forbid hipMalloc, processes whose device memory is fully budgeted).  It has no
 * counterpart in the reference.  It is a blocking call like trp_preprocess_batch_device: it synchronises the
 * context's stream and reads the batch totals on the host, so it cannot be recorded into a HIP graph (stream
 * capture rejects the synchronise).  The call allocates and frees nothing, on the device or the host: every working array lives in a
 * fixed scratch array that is part of the library's code object (about 25 MiB on each GPU the library is loaded
 * on, held until the process ends), and every result array is the caller's device memory, described by
 * trp_out_buffers:
 *   denoised (16-byte aligned, denoised_cap >= n_bytes bytes, all of which the kernels may use as working space:
 *   the text is its first n_denoised bytes afterwards; unused with the denoiser disabled, where the text is the
 *   input itself), denoised_offsets / tok_offsets / chunk_offsets (n_records + 1 entries each), dropped
 *   (n_records), tok_start / tok_len (tok_cap >= trp_token_bound(n_bytes, n_records) entries each), chunks
 *   (chunk_cap entries).
 * Limits: n_bytes <= TRP_NOALLOC_MAX_BYTES and n_records <= TRP_NOALLOC_MAX_RECORDS (TRP_E_TOO_LARGE beyond;
 * split by record range); TRP_E_TOO_LARGE also when the batch has more chunks than chunk_cap (out->n_chunks
 * then says how many) or more digit-heavy lines left to the work list than one per 64 bytes of text.  Results
 * are identical to trp_preprocess_batch_device; `out` points into the caller's buffers.  The call returns after
 * the kernels have completed, like trp_preprocess_batch_device.  All calls on one GPU share that GPU's scratch
 * array, from whatever context or thread: calls that overlap in time on one GPU race and must be serialised by
 * the caller.  The other entry points do not use the array.  The batch becomes "the last device-resident batch" of
 * the context: the text renderers below (which do allocate their own output) read it through the caller's
 * buffers, which must therefore stay untouched until the next batch call on the context. */
#define TRP_NOALLOC_MAX_BYTES (32ull << 20)
#define TRP_NOALLOC_MAX_RECORDS (1ull << 19)
typedef struct {
    uint8_t* denoised;
    uint64_t denoised_cap;
    uint64_t* denoised_offsets;
    uint8_t* dropped;
    uint32_t* tok_start;
    uint32_t* tok_len;
    uint64_t tok_cap;
    uint64_t* tok_offsets;
    trp_chunk* chunks;
    uint64_t chunk_cap;
    uint64_t* chunk_offsets;
} trp_out_buffers;
uint64_t trp_token_bound(uint64_t n_bytes, uint64_t n_records);
int trp_preprocess_batch_device_noalloc(trp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                                        uint64_t n_records, uint64_t n_bytes, const trp_config* cfg,
                                        const trp_out_buffers* buffers, trp_result* out);

/* Copy `bytes` bytes of a device array of the last result into host memory (ordered on the context's
 * stream, returns when the copy is complete). */
int trp_copy_to_host(trp_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* Chunk text of the last device-resident batch, rendered as the reference does
 * (sliding_window.rs:44, :67, :95-100): tokens joined by one space, or the preprocessed text verbatim
 * for a record with a single window.  Produces device arrays: text bytes of all chunks concatenated
 * and text_offsets[n_chunks+1]. */
int trp_materialize_chunk_text_device(trp_ctx* ctx, uint8_t** d_text, uint64_t** d_text_offsets,
                                      uint64_t* n_text_bytes);

/* "First k tokens" text of every record of the last device-resident batch: the record's preprocessed text
 * verbatim when it has at most max_tokens tokens, otherwise its first max_tokens tokens joined by one
 * space.  With max_tokens = 64 over a batch of query texts this is the BM25 query cap
 * (src/sampler/backends/bm25_backend.rs:219-225).  Produces device arrays like
 * trp_materialize_chunk_text_device: text bytes and text_offsets[R+1].  (A limit of 0 selects no tokens; the
 * reference's "0 = no limit" convention of record_bm25_text is implemented by trp_bm25_texts_device.) */
int trp_head_tokens_text_device(trp_ctx* ctx, uint32_t max_tokens, uint8_t** d_text,
                                uint64_t** d_text_offsets, uint64_t* n_text_bytes);

/* The body of decorate_chunk (src/sampler/mod.rs:1484-1506, rng-free twin :1596-1618) for n selected chunks of
 * the last device-resident batch.  chunk_index[i] (host array) names a chunk of the batch, budget[i] (host
 * array) the number of body tokens that fit behind the sampled prefix: max_window_tokens - prefix_tokens when
 * prefix_tokens + body_tokens > max_window_tokens > 0 (:1497-1498), any value >= the chunk's token count
 * otherwise.  Text i is the chunk's text verbatim when the budget covers it (:1504), else its first budget[i]
 * tokens joined by one space (:1498-1500); the caller puts prefix + platform_newline() in front (the prefix
 * is sampled by the host's RNG, src/kvp.rs) and sets tokens_estimate (:1495, :1501, :1505).  Produces device
 * arrays: text bytes and text_offsets[n+1]. */
int trp_chunk_bodies_device(trp_ctx* ctx, const uint32_t* chunk_index, const uint32_t* budget, uint64_t n,
                            uint8_t** d_text, uint64_t** d_text_offsets, uint64_t* n_text_bytes);

/* record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532) for n_docs documents over the last
 * device-resident batch, whose records are the headings and section texts of the documents in order (a None
 * heading is simply left out of the batch): document i is records [doc_offsets[i], doc_offsets[i+1]) (host
 * array, n_docs+1 entries).  Text i = every non-blank record followed by `newline` (:505-516; blank = no token,
 * i.e. trim().is_empty()), cut to its first max_tokens tokens joined by one space when it has more than
 * max_tokens > 0 of them (:523-531); max_tokens = 0 means no limit (:520-521).  A document without any token
 * yields the empty text: the caller substitutes the record id (:517-519).  `newline` is platform_newline()
 * ("\n" or "\r\n").  Produces device arrays: text bytes and text_offsets[n_docs+1]. */
int trp_bm25_texts_device(trp_ctx* ctx, const uint64_t* doc_offsets, uint64_t n_docs, uint32_t max_tokens,
                          const char* newline, uint8_t** d_text, uint64_t** d_text_offsets,
                          uint64_t* n_text_bytes);

/* The two steps above for callers whose texts live in host memory (the reference's call sites do: decorate_chunk at
 * src/sampler/mod.rs:1470-1508 / :1582-1620 works on a RecordChunk's String, record_bm25_text at
 * src/sampler/backends/bm25_backend.rs:503-532 on a DataRecord's sections).  Each call copies the packed batch to
 * the GPU, runs it as ONE device batch (at most TRP_MAX_BATCH_BYTES; no sub-batching), renders the requested texts
 * there and returns them in library-owned host memory (release with trp_chunk_text_free): out->offsets has n + 1
 * (n_docs + 1) entries.
 * trp_head_tokens_text: trp_head_tokens_text_device over the batch's records as they are (no denoiser): text r is
 *   record r when it has at most max_tokens tokens, else its first max_tokens tokens joined by one space - the
 *   BM25 query cap with max_tokens = 64 (src/sampler/backends/bm25_backend.rs:219-225).
 * trp_bm25_texts: the records of the batch are the headings and section texts of the documents in order, as for
 *   trp_bm25_texts_device; the texts are tokenised as they are (no denoiser: record_bm25_text applies none).
 * trp_chunk_bodies: the batch is preprocessed and chunked with `cfg` exactly as trp_preprocess_batch does, then
 *   chunk chunk_index[i] of that result is cut to budget[i] tokens as for trp_chunk_bodies_device.  (chunk_index
 *   refers to the batch's chunk order: a caller that does not know it yet calls trp_preprocess_batch_select with
 *   TRP_OUT_CHUNKS first, or uses the device variants, which keep the batch on the GPU between the two steps.) */
int trp_head_tokens_text(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                         uint32_t max_tokens, trp_chunk_text* out);
int trp_bm25_texts(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                   const uint64_t* doc_offsets, uint64_t n_docs, uint32_t max_tokens, const char* newline,
                   trp_chunk_text* out);
int trp_chunk_bodies(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                     const trp_config* cfg, const uint32_t* chunk_index, const uint32_t* budget, uint64_t n,
                     trp_chunk_text* out);

/* normalize_inline_whitespace (src/utils.rs:18-33) for every record of a batch: each run of White_Space
 * characters becomes one ' ' and the result is trimmed.  The File and CSV sources apply it to every text
 * they ingest (src/source/backends/file_source.rs:260, csv_source.rs:280-310), i.e. it is the step just
 * before the preprocessing path.  Same calling conventions and result layout as trp_preprocess_batch /
 * trp_preprocess_batch_device: `denoised` holds the normalised text, the token table its tokens, the
 * chunk descriptors are those of the reference's default chunking, dropped[] is all zero (the function
 * never returns None; an all-whitespace record becomes the empty string). */
int trp_normalize_batch(trp_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                        trp_result* out);
int trp_normalize_batch_device(trp_ctx* ctx, const uint8_t* d_bytes, const uint64_t* d_offsets,
                               uint64_t n_records, uint64_t n_bytes, trp_result* out);
/* One text (a batch of one): malloc'd copy in *out (free with trp_free). */
int trp_normalize_inline_whitespace(trp_ctx* ctx, const uint8_t* text, size_t len, uint8_t** out,
                                    size_t* out_len);

/* Single-record conveniences mirroring the reference's public functions (each runs a batch of one).
 * trp_denoise: denoise_text (denoiser_preprocessor.rs:221): returns 1 for Some(text) with a malloc'd
 *   copy in *out (free with trp_free), 0 for None, negative on error.
 * trp_tokenize: WhitespaceTokenizer::tokenize (tokenizer.rs:57): (start,len) pairs into `text`.
 * trp_token_count: WhitespaceTokenizer::token_count (tokenizer.rs:62).
 * trp_chunk_section: SlidingWindowChunker::materialize (sliding_window.rs:11) for one section. */
int trp_denoise(trp_ctx* ctx, const uint8_t* text, size_t len, const trp_config* cfg, uint8_t** out,
                size_t* out_len);
int trp_tokenize(trp_ctx* ctx, const uint8_t* text, size_t len, uint32_t** starts, uint32_t** lens,
                 size_t* n_tokens);
int trp_token_count(trp_ctx* ctx, const uint8_t* text, size_t len, size_t* n_tokens);
int trp_chunk_section(trp_ctx* ctx, const uint8_t* text, size_t len, const trp_config* cfg,
                      trp_result* out);
void trp_free(void* p);

/* Per-kernel timing with HIP events on the context's stream.  When on, the next batch call records
 * an event pair around every kernel; trp_get_profile returns the durations of that call (number of
 * entries, at most cap). */
int trp_set_profiling(trp_ctx* ctx, int on);
int trp_get_profile(trp_ctx* ctx, trp_kernel_time* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
