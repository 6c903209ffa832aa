//! GPU-backed implementations of the two plug-in traits of the `triplets` crate:
//!
//! * [`GpuSlidingWindowChunker`] implements `triplets::ChunkingAlgorithm` (the trait the sampler is
//!   given through `TripletSampler::new_with_chunker`).  The sampler asks for the chunks of ONE section
//!   at a time, many times per batch, so the chunker answers from a table that [`GpuSlidingWindowChunker::
//!   prefill`] computes for a whole set of records in one GPU batch; a section that is not in the table
//!   is computed as a batch of one.
//! * [`GpuDenoiserPreprocessor`] implements `triplets::TextPreprocessor` for callers that only want
//!   the denoiser.
//!
//! The library never falls back to a CPU implementation: when the GPU call fails the chunker returns no
//! chunks (the sampler then tries another section, exactly as for a section the denoiser dropped) and
//! the error text is reported through `log_error`.
pub mod ffi;

use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::{Mutex, RwLock};

use triplets::config::{ChunkingStrategy, DenoiserConfig};
use triplets::data::{ChunkView, DataRecord, RecordChunk, RecordSection};
use triplets::{ChunkingAlgorithm, TextPreprocessor};

fn last_error() -> String {
    unsafe {
        let p = ffi::trp_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

fn log_error(what: &str) {
    eprintln!("triplets-amd: {what}: {}", last_error());
}

/// One GPU context (one device, one stream).  The C library wants one caller at a time per context.
pub struct Context {
    raw: Mutex<*mut ffi::trp_ctx>,
}
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Context {
    pub fn new(device: i32) -> Result<Self, String> {
        let mut raw = std::ptr::null_mut();
        let rc = unsafe { ffi::trp_create(device, &mut raw) };
        if rc != ffi::TRP_OK {
            return Err(last_error());
        }
        Ok(Self { raw: Mutex::new(raw) })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        let raw = *self.raw.lock().unwrap();
        unsafe { ffi::trp_destroy(raw) };
    }
}

/// Chunk descriptors and preprocessed text of a batch of sections, in host memory.
struct Batch {
    raw: ffi::trp_result,
}
unsafe impl Send for Batch {}

impl Drop for Batch {
    fn drop(&mut self) {
        unsafe { ffi::trp_result_free(&mut self.raw) };
    }
}

impl Batch {
    fn run(ctx: &Context, texts: &[&str], denoiser: Option<&DenoiserConfig>, st: &ChunkingStrategy) -> Option<Batch> {
        let mut bytes = Vec::with_capacity(texts.iter().map(|t| t.len()).sum());
        let mut offsets = Vec::with_capacity(texts.len() + 1);
        offsets.push(0u64);
        for t in texts {
            bytes.extend_from_slice(t.as_bytes());
            offsets.push(bytes.len() as u64);
        }
        let overlaps: Vec<u32> = st.overlap_tokens.iter().map(|&o| o.min(u32::MAX as usize) as u32).collect();
        let cfg = ffi::trp_config {
            denoise_enabled: denoiser.map_or(0, |d| d.enabled as u8),
            strip_markdown: denoiser.map_or(1, |d| d.strip_markdown as u8),
            max_digit_ratio: denoiser.map_or(0.35, |d| d.max_digit_ratio),
            max_window_tokens: st.max_window_tokens.min(u32::MAX as usize) as u32,
            overlaps: overlaps.as_ptr(),
            n_overlaps: overlaps.len() as u32,
            summary_fallback_tokens: st.summary_fallback_tokens.min(u32::MAX as usize) as u32,
        };
        let mut raw: ffi::trp_result = unsafe { std::mem::zeroed() };
        let guard = ctx.raw.lock().unwrap();
        let rc = unsafe {
            ffi::trp_preprocess_batch(*guard, bytes.as_ptr(), offsets.as_ptr(), texts.len() as u64, &cfg, &mut raw)
        };
        drop(guard);
        if rc != ffi::TRP_OK {
            log_error("trp_preprocess_batch");
            return None;
        }
        Some(Batch { raw })
    }

    fn slice<T>(p: *const T, n: usize) -> &'static [T] {
        if n == 0 { &[] } else { unsafe { std::slice::from_raw_parts(p, n) } }
    }

    /// Chunk prototypes (view, text, tokens_estimate) of section `r` of the batch.
    fn chunks_of(&self, r: usize, summary_weight: f32) -> Vec<(ChunkView, String, usize)> {
        let res = &self.raw;
        let n = res.n_records as usize;
        let den = Self::slice(res.denoised, res.n_denoised as usize);
        let den_off = Self::slice(res.denoised_offsets, n + 1);
        let tok_off = Self::slice(res.tok_offsets, n + 1);
        let tok_start = Self::slice(res.tok_start, res.n_tokens as usize);
        let tok_len = Self::slice(res.tok_len, res.n_tokens as usize);
        let ch_off = Self::slice(res.chunk_offsets, n + 1);
        let chunks = Self::slice(res.chunks, res.n_chunks as usize);
        let t0 = tok_off[r] as usize;
        let total = (tok_off[r + 1] - tok_off[r]) as usize;
        let mut out = Vec::new();
        for c in &chunks[ch_off[r] as usize..ch_off[r + 1] as usize] {
            let (b, e) = (c.tok_begin as usize, c.tok_end as usize);
            // One window that covers the whole section carries the preprocessed text verbatim; every
            // other chunk is its tokens joined by single spaces.
            let text = if c.kind == 0 && b == 0 && e == total && c.span as usize == total {
                let s = &den[den_off[r] as usize..den_off[r + 1] as usize];
                String::from_utf8_lossy(s).into_owned()
            } else {
                let mut s = String::new();
                for k in t0 + b..t0 + e {
                    if k > t0 + b {
                        s.push(' ');
                    }
                    let at = tok_start[k] as usize;
                    s.push_str(&String::from_utf8_lossy(&den[at..at + tok_len[k] as usize]));
                }
                s
            };
            let view = if c.kind == 0 {
                ChunkView::Window { index: c.index as usize, overlap: c.overlap as usize, span: c.span as usize }
            } else {
                ChunkView::SummaryFallback { strategy: "head".into(), weight: summary_weight }
            };
            out.push((view, text, e - b));
        }
        out
    }
}

type Key = (String, usize);

/// `ChunkingAlgorithm` served from GPU batches.  Build it with the denoiser settings it should apply
/// (do NOT also register a `DenoiserPreprocessor` on the strategy: the GPU path already denoises).
pub struct GpuSlidingWindowChunker {
    ctx: Context,
    denoiser: Option<DenoiserConfig>,
    table: RwLock<HashMap<Key, Vec<(ChunkView, String, usize)>>>,
}

impl GpuSlidingWindowChunker {
    pub fn new(device: i32, denoiser: Option<DenoiserConfig>) -> Result<Self, String> {
        Ok(Self { ctx: Context::new(device)?, denoiser, table: RwLock::new(HashMap::new()) })
    }

    /// Runs every section of `records` through the GPU in one batch and remembers the chunks.
    /// Natural call site: where the sampler refreshes its record cache at ingest time.
    pub fn prefill(&self, strategy: &ChunkingStrategy, records: &[DataRecord]) {
        let mut keys = Vec::new();
        let mut texts = Vec::new();
        for rec in records {
            for (i, s) in rec.sections.iter().enumerate() {
                keys.push((rec.id.to_string(), i));
                texts.push(s.text.as_str());
            }
        }
        if let Some(batch) = Batch::run(&self.ctx, &texts, self.denoiser.as_ref(), strategy) {
            let mut table = self.table.write().unwrap();
            for (r, key) in keys.into_iter().enumerate() {
                table.insert(key, batch.chunks_of(r, strategy.summary_fallback_weight));
            }
        }
    }

    pub fn clear(&self) {
        self.table.write().unwrap().clear();
    }
}

impl ChunkingAlgorithm for GpuSlidingWindowChunker {
    fn materialize(
        &self,
        strategy: &ChunkingStrategy,
        record: &DataRecord,
        section_idx: usize,
        section: &RecordSection,
    ) -> Vec<RecordChunk> {
        let key = (record.id.to_string(), section_idx);
        let protos = match self.table.read().unwrap().get(&key) {
            Some(p) => p.clone(),
            None => match Batch::run(&self.ctx, &[section.text.as_str()], self.denoiser.as_ref(), strategy) {
                Some(b) => b.chunks_of(0, strategy.summary_fallback_weight),
                None => return Vec::new(),
            },
        };
        protos
            .into_iter()
            .map(|(view, text, tokens_estimate)| RecordChunk {
                record_id: record.id.clone(),
                section_idx,
                view,
                text,
                tokens_estimate,
                quality: record.quality,
                kvp_meta: Default::default(),
            })
            .collect()
    }
}

/// `TextPreprocessor` that runs the denoiser on the GPU (a batch of one per call; prefer the chunker
/// above, which batches).
pub struct GpuDenoiserPreprocessor {
    ctx: Context,
    config: DenoiserConfig,
}

impl GpuDenoiserPreprocessor {
    pub fn new(device: i32, config: DenoiserConfig) -> Result<Self, String> {
        Ok(Self { ctx: Context::new(device)?, config })
    }
}

impl TextPreprocessor for GpuDenoiserPreprocessor {
    fn process(&self, text: &str) -> Option<String> {
        let none = 0u32;
        let cfg = ffi::trp_config {
            denoise_enabled: self.config.enabled as u8,
            strip_markdown: self.config.strip_markdown as u8,
            max_digit_ratio: self.config.max_digit_ratio,
            max_window_tokens: 1024,
            overlaps: &none,
            n_overlaps: 0,
            summary_fallback_tokens: 0,
        };
        let mut out: *mut u8 = std::ptr::null_mut();
        let mut len = 0usize;
        let guard = self.ctx.raw.lock().unwrap();
        let rc = unsafe { ffi::trp_denoise(*guard, text.as_ptr(), text.len(), &cfg, &mut out, &mut len) };
        drop(guard);
        if rc < 0 {
            log_error("trp_denoise");
            return None;
        }
        if rc == 0 {
            return None;
        }
        let s = unsafe { String::from_utf8_lossy(std::slice::from_raw_parts(out, len)).into_owned() };
        unsafe { ffi::trp_free(out as *mut _) };
        Some(s)
    }
}
