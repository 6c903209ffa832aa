//! Raw declarations of include/triplets_amd.h.  Field order and widths must match the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const TRP_OK: c_int = 0;
pub const TRP_MAX_BATCH_BYTES: u64 = 0xFFFF_0000;
pub const TRP_OUT_TEXT: u32 = 1;
pub const TRP_OUT_TOKENS: u32 = 2;
pub const TRP_OUT_CHUNKS: u32 = 4;
pub const TRP_OUT_CHUNK_TEXT: u32 = 8;
pub const TRP_OUT_ALL: u32 = 7;

#[repr(C)]
pub struct trp_ctx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct trp_config {
    pub denoise_enabled: u8,
    pub strip_markdown: u8,
    pub max_digit_ratio: f32,
    pub max_window_tokens: u32,
    pub overlaps: *const u32,
    pub n_overlaps: u32,
    pub summary_fallback_tokens: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct trp_chunk {
    pub record: u32,
    pub kind: u32,
    pub index: u32,
    pub overlap: u32,
    pub span: u32,
    pub tok_begin: u32,
    pub tok_end: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct trp_result {
    pub denoised: *mut u8,
    pub denoised_offsets: *mut u64,
    pub dropped: *mut u8,
    pub tok_start: *mut u32,
    pub tok_len: *mut u32,
    pub tok_offsets: *mut u64,
    pub chunks: *mut trp_chunk,
    pub chunk_offsets: *mut u64,
    pub n_records: u64,
    pub n_denoised: u64,
    pub n_tokens: u64,
    pub n_chunks: u64,
}

/// Texts rendered on the GPU, in library-owned host memory (release with `trp_chunk_text_free`).
#[repr(C)]
pub struct trp_chunk_text {
    pub text: *mut u8,
    pub offsets: *mut u64,
    pub n_bytes: u64,
}

/// One more DenoiserPreprocessor of a strategy's fold (see trp_preprocess_batch_fold).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct trp_denoiser {
    pub denoise_enabled: u8,
    pub strip_markdown: u8,
    pub max_digit_ratio: f32,
}

extern "C" {
    pub fn trp_create(device: c_int, out: *mut *mut trp_ctx) -> c_int;
    pub fn trp_device_count() -> c_int;
    pub fn trp_destroy(ctx: *mut trp_ctx);
    pub fn trp_last_error() -> *const c_char;
    pub fn trp_preprocess_batch(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        cfg: *const trp_config,
        out: *mut trp_result,
    ) -> c_int;
    /// Same call, returning only the arrays named by `outputs` (per-record arrays always come back).
    pub fn trp_preprocess_batch_select(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        cfg: *const trp_config,
        outputs: u32,
        out: *mut trp_result,
        chunk_text: *mut trp_chunk_text,
    ) -> c_int;
    /// The same for a strategy with several registered denoisers: `cfg`'s own first, then `more` in order
    /// (ChunkingStrategy::register_preprocessor, src/config.rs:105-113; the fold of
    /// src/chunking/sliding_window.rs:20-34).
    pub fn trp_preprocess_batch_fold(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        cfg: *const trp_config,
        more: *const trp_denoiser,
        n_more: u32,
        outputs: u32,
        out: *mut trp_result,
        chunk_text: *mut trp_chunk_text,
    ) -> c_int;
    pub fn trp_result_free(r: *mut trp_result);
    pub fn trp_chunk_text_free(t: *mut trp_chunk_text);
    /// record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532) over a packed host batch.
    pub fn trp_bm25_texts(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        doc_offsets: *const u64,
        n_docs: u64,
        max_tokens: u32,
        newline: *const c_char,
        out: *mut trp_chunk_text,
    ) -> c_int;
    /// The BM25 query cap (bm25_backend.rs:219-225): every record cut to its first `max_tokens` tokens.
    pub fn trp_head_tokens_text(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        max_tokens: u32,
        out: *mut trp_chunk_text,
    ) -> c_int;
    /// Bodies of selected chunks cut to a token budget (decorate_chunk, src/sampler/mod.rs:1497-1504).
    pub fn trp_chunk_bodies(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        cfg: *const trp_config,
        chunk_index: *const u32,
        budget: *const u32,
        n: u64,
        out: *mut trp_chunk_text,
    ) -> c_int;
    pub fn trp_denoise(
        ctx: *mut trp_ctx,
        text: *const u8,
        len: usize,
        cfg: *const trp_config,
        out: *mut *mut u8,
        out_len: *mut usize,
    ) -> c_int;
    pub fn trp_token_count(ctx: *mut trp_ctx, text: *const u8, len: usize, n: *mut usize) -> c_int;
    pub fn trp_normalize_batch(
        ctx: *mut trp_ctx,
        bytes: *const u8,
        offsets: *const u64,
        n_records: u64,
        out: *mut trp_result,
    ) -> c_int;
    pub fn trp_normalize_inline_whitespace(
        ctx: *mut trp_ctx,
        text: *const u8,
        len: usize,
        out: *mut *mut u8,
        out_len: *mut usize,
    ) -> c_int;
    pub fn trp_free(p: *mut c_void);
}
