//! GPU-backed implementations of the two plug-in traits of the `triplets` crate:
//!
//! * [`GpuSlidingWindowChunker`] implements `triplets::ChunkingAlgorithm` (the trait the sampler is
//!   given through `TripletSampler::new_with_chunker`).  The sampler asks for the chunks of ONE section
//!   at a time, many times per batch and from several worker threads, so the chunker answers from a
//!   table that [`GpuSlidingWindowChunker::prefill`] computes for a whole set of records in one GPU
//!   batch; sections that miss the table are queued and computed together by whichever thread gets
//!   the GPU next.
//! * [`GpuDenoiserPreprocessor`] implements `triplets::TextPreprocessor` for callers that only want
//!   the denoiser.
//!
//! The library never falls back to a CPU implementation.  The traits have no error channel, so a GPU
//! failure is kept as a sticky error ([`GpuSlidingWindowChunker::take_error`]) and reported through
//! `log_error`; `prefill` returns it as a `Result`.  NOTE: this crate has not been compiled (the build
//! environment has no Rust toolchain); the tested host sides are the C++ and Python ones.
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
    /// `outputs`: TRP_OUT_* selection (0 = per-record arrays only, enough for token counts).
    fn run(
        ctx: &Context,
        texts: &[&str],
        denoiser: Option<&DenoiserConfig>,
        st: &ChunkingStrategy,
        outputs: u32,
    ) -> Result<Batch, String> {
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
            ffi::trp_preprocess_batch_select(
                *guard,
                bytes.as_ptr(),
                offsets.as_ptr(),
                texts.len() as u64,
                &cfg,
                outputs,
                &mut raw,
                std::ptr::null_mut(),
            )
        };
        drop(guard);
        if rc != ffi::TRP_OK {
            let e = last_error();
            log_error("trp_preprocess_batch_select");
            return Err(e);
        }
        Ok(Batch { raw })
    }

    fn token_count(&self, r: usize) -> usize {
        let off = Self::slice(self.raw.tok_offsets, self.raw.n_records as usize + 1);
        (off[r + 1] - off[r]) as usize
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

type Proto = (ChunkView, String, usize);
/// (strategy fingerprint, record id, section index): chunks made for one `ChunkingStrategy` are never
/// served for another.
type Key = (u64, String, usize);

/// FNV-1a over everything of the strategy and the denoiser settings that changes the result.
fn strategy_fingerprint(st: &ChunkingStrategy, den: Option<&DenoiserConfig>) -> u64 {
    let mut h: u64 = 0xcbf2_9ce4_8422_2325;
    let mut mix = |v: u64| {
        for b in v.to_le_bytes() {
            h ^= b as u64;
            h = h.wrapping_mul(0x0000_0100_0000_01b3);
        }
    };
    mix(st.max_window_tokens as u64);
    mix(st.overlap_tokens.len() as u64);
    for &o in &st.overlap_tokens {
        mix(o as u64);
    }
    mix(st.summary_fallback_tokens as u64);
    mix(st.summary_fallback_weight.to_bits() as u64);
    let on = den.map_or(false, |d| d.enabled);
    mix(on as u64);
    if let Some(d) = den.filter(|d| d.enabled) {
        mix(d.max_digit_ratio.to_bits() as u64);
        mix(d.strip_markdown as u64);
    }
    h
}

struct Miss {
    key: Key,
    text: *const str, // the caller's section text; the caller is blocked until `done`
    strategy: *const ChunkingStrategy,
    done: bool,
}

/// `ChunkingAlgorithm` served from GPU batches.  Build it with the denoiser settings it should apply
/// (do NOT also register a `DenoiserPreprocessor` on the strategy: the GPU path already denoises).
///
/// A strategy that carries registered preprocessors (`strategy.preprocessors()` non-empty) cannot be
/// served: the GPU path implements the denoiser only, and running arbitrary host preprocessors would
/// need the reference's CPU chunker.  Such a call records an error, logs it and returns no chunks;
/// give such samplers the reference's own `SlidingWindowChunker`.
pub struct GpuSlidingWindowChunker {
    ctx: Context,
    denoiser: Option<DenoiserConfig>,
    table: RwLock<HashMap<Key, Vec<Proto>>>,
    raw_tokens: RwLock<HashMap<(String, usize), usize>>,
    queue: Mutex<Vec<*mut Miss>>,
    gpu: Mutex<()>,
    error: Mutex<Option<String>>,
}
unsafe impl Send for GpuSlidingWindowChunker {}
unsafe impl Sync for GpuSlidingWindowChunker {}

impl GpuSlidingWindowChunker {
    pub fn new(device: i32, denoiser: Option<DenoiserConfig>) -> Result<Self, String> {
        Ok(Self {
            ctx: Context::new(device)?,
            denoiser,
            table: RwLock::new(HashMap::new()),
            raw_tokens: RwLock::new(HashMap::new()),
            queue: Mutex::new(Vec::new()),
            gpu: Mutex::new(()),
            error: Mutex::new(None),
        })
    }

    /// The first GPU failure since the last call, if any (the traits cannot return it).
    pub fn take_error(&self) -> Option<String> {
        self.error.lock().unwrap().take()
    }

    fn fail(&self, e: String) -> String {
        let mut slot = self.error.lock().unwrap();
        if slot.is_none() {
            *slot = Some(e.clone());
        }
        e
    }

    /// Runs every section of `records` through the GPU and remembers the chunks and the token count of
    /// the RAW section text (`WhitespaceTokenizer.token_count(&section.text)`, what
    /// `record_has_long_anchor_or_context_section` asks, src/sampler/mod.rs:1803-1812).  Inputs larger
    /// than one batch are split by record range.  Natural call site: where the sampler refreshes its
    /// record cache at ingest time.
    pub fn prefill(&self, strategy: &ChunkingStrategy, records: &[DataRecord]) -> Result<(), String> {
        if !strategy.preprocessors().is_empty() {
            return Err(self.fail("strategy has registered preprocessors: not served by the GPU chunker".into()));
        }
        let mut keys: Vec<(String, usize)> = Vec::new();
        let mut texts: Vec<&str> = Vec::new();
        for rec in records {
            for (i, s) in rec.sections.iter().enumerate() {
                keys.push((rec.id.to_string(), i));
                texts.push(s.text.as_str());
            }
        }
        let _gpu = self.gpu.lock().unwrap();
        let mut at = 0;
        while at < texts.len() {
            // as many sections as fit one batch (a single section above the limit fails in the library)
            let (mut end, mut bytes) = (at, 0u64);
            while end < texts.len() && (end == at || bytes + texts[end].len() as u64 <= ffi::TRP_MAX_BATCH_BYTES) {
                bytes += texts[end].len() as u64;
                end += 1;
            }
            self.compute(strategy, &keys[at..end], &texts[at..end]).map_err(|e| self.fail(e))?;
            at = end;
        }
        Ok(())
    }

    /// `gpu` held.  One batch for the chunks and, with a denoiser, one pass-through batch for the raw counts.
    fn compute(&self, strategy: &ChunkingStrategy, keys: &[(String, usize)], texts: &[&str]) -> Result<(), String> {
        let batch = Batch::run(&self.ctx, texts, self.denoiser.as_ref(), strategy, ffi::TRP_OUT_ALL)?;
        let denoising = self.denoiser.as_ref().map_or(false, |d| d.enabled);
        let raw: Vec<usize> = if denoising {
            let plain = Batch::run(&self.ctx, texts, None, strategy, 0)?;
            (0..texts.len()).map(|r| plain.token_count(r)).collect()
        } else {
            (0..texts.len()).map(|r| batch.token_count(r)).collect()
        };
        let fp = strategy_fingerprint(strategy, self.denoiser.as_ref());
        let protos: Vec<Vec<Proto>> =
            (0..texts.len()).map(|r| batch.chunks_of(r, strategy.summary_fallback_weight)).collect();
        let mut table = self.table.write().unwrap();
        let mut counts = self.raw_tokens.write().unwrap();
        for ((key, p), n) in keys.iter().zip(protos).zip(raw) {
            table.insert((fp, key.0.clone(), key.1), p);
            counts.insert(key.clone(), n);
        }
        Ok(())
    }

    /// Token count of the raw text of a prefilled section.
    pub fn token_count(&self, record_id: &str, section_idx: usize) -> Option<usize> {
        self.raw_tokens.read().unwrap().get(&(record_id.to_string(), section_idx)).copied()
    }

    pub fn clear(&self) {
        self.table.write().unwrap().clear();
        self.raw_tokens.write().unwrap().clear();
    }

    /// `gpu` held: everything queued right now, one batch per strategy fingerprint.
    fn drain(&self) {
        let mut work: Vec<*mut Miss> = std::mem::take(&mut *self.queue.lock().unwrap());
        while let Some(&first) = work.first() {
            let fp = unsafe { (*first).key.0 };
            let (same, rest): (Vec<_>, Vec<_>) = work.into_iter().partition(|&m| unsafe { (*m).key.0 == fp });
            let keys: Vec<(String, usize)> = same.iter().map(|&m| unsafe { ((*m).key.1.clone(), (*m).key.2) }).collect();
            let texts: Vec<&str> = same.iter().map(|&m| unsafe { &*(*m).text }).collect();
            let strategy = unsafe { &*(*same[0]).strategy };
            if let Err(e) = self.compute(strategy, &keys, &texts) {
                self.fail(e);
            }
            for &m in &same {
                unsafe { (*m).done = true };
            }
            work = rest;
        }
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
        if !strategy.preprocessors().is_empty() {
            self.fail("strategy has registered preprocessors: not served by the GPU chunker".into());
            eprintln!("triplets-amd: strategy with registered preprocessors given to the GPU chunker; no chunks");
            return Vec::new();
        }
        let key: Key = (strategy_fingerprint(strategy, self.denoiser.as_ref()), record.id.to_string(), section_idx);
        let mut protos = self.table.read().unwrap().get(&key).cloned();
        if protos.is_none() {
            // miss: queue it; the first thread to get the GPU computes everything queued
            let mut me = Miss { key: key.clone(), text: section.text.as_str(), strategy, done: false };
            self.queue.lock().unwrap().push(&mut me);
            {
                let _gpu = self.gpu.lock().unwrap();
                if !me.done {
                    self.drain();
                }
            }
            protos = self.table.read().unwrap().get(&key).cloned();
        }
        // a GPU failure leaves the section out of the table: the error is kept (take_error) and logged
        protos
            .unwrap_or_default()
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
