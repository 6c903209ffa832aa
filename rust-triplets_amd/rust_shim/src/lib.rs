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

use std::collections::{HashMap, VecDeque};
use std::ffi::CStr;
use std::sync::{Arc, Mutex, RwLock};

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

/// Chunk text rendered on the GPU (library-owned host memory).
struct ChunkText {
    raw: ffi::trp_chunk_text,
}
unsafe impl Send for ChunkText {}
impl Drop for ChunkText {
    fn drop(&mut self) {
        unsafe { ffi::trp_chunk_text_free(&mut self.raw) };
    }
}

impl Batch {
    /// `outputs`: TRP_OUT_* selection (0 = per-record arrays only, enough for token counts).
    fn run(
        ctx: &Context,
        texts: &[&str],
        denoiser: Option<&DenoiserConfig>,
        more: &[DenoiserConfig],
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
        let cfg = Self::config(denoiser, st, &overlaps);
        let more = Self::more(more);
        let mut raw: ffi::trp_result = unsafe { std::mem::zeroed() };
        let guard = ctx.raw.lock().unwrap();
        let rc = unsafe {
            ffi::trp_preprocess_batch_fold(
                *guard,
                bytes.as_ptr(),
                offsets.as_ptr(),
                texts.len() as u64,
                &cfg,
                more.as_ptr(),
                more.len() as u32,
                outputs,
                &mut raw,
                std::ptr::null_mut::<ffi::trp_chunk_text>(),
            )
        };
        drop(guard);
        if rc != ffi::TRP_OK {
            let e = last_error();
            log_error("trp_preprocess_batch_fold");
            return Err(e);
        }
        Ok(Batch { raw })
    }

    /// The denoisers registered after the first one, as the C ABI takes them.
    fn more(more: &[DenoiserConfig]) -> Vec<ffi::trp_denoiser> {
        more.iter()
            .map(|d| ffi::trp_denoiser {
                denoise_enabled: d.enabled as u8,
                strip_markdown: d.strip_markdown as u8,
                max_digit_ratio: d.max_digit_ratio,
            })
            .collect()
    }

    fn config(denoiser: Option<&DenoiserConfig>, st: &ChunkingStrategy, overlaps: &[u32]) -> ffi::trp_config {
        ffi::trp_config {
            denoise_enabled: denoiser.map_or(0, |d| d.enabled as u8),
            strip_markdown: denoiser.map_or(1, |d| d.strip_markdown as u8),
            max_digit_ratio: denoiser.map_or(0.35, |d| d.max_digit_ratio),
            max_window_tokens: st.max_window_tokens.min(u32::MAX as usize) as u32,
            overlaps: overlaps.as_ptr(),
            n_overlaps: overlaps.len() as u32,
            summary_fallback_tokens: st.summary_fallback_tokens.min(u32::MAX as usize) as u32,
        }
    }

    /// Chunk descriptors + the text of every chunk rendered on the GPU (what a `materialize` table needs): the
    /// preprocessed text and the token table stay on the device.
    fn run_with_chunk_text(
        ctx: &Context,
        texts: &[&str],
        denoiser: Option<&DenoiserConfig>,
        more: &[DenoiserConfig],
        st: &ChunkingStrategy,
    ) -> Result<(Batch, ChunkText), String> {
        let (bytes, offsets) = pack(texts);
        let overlaps: Vec<u32> = st.overlap_tokens.iter().map(|&o| o.min(u32::MAX as usize) as u32).collect();
        let cfg = Self::config(denoiser, st, &overlaps);
        let mut raw: ffi::trp_result = unsafe { std::mem::zeroed() };
        let mut ct: ffi::trp_chunk_text = unsafe { std::mem::zeroed() };
        let more = Self::more(more);
        let guard = ctx.raw.lock().unwrap_or_else(|p| p.into_inner());
        let rc = unsafe {
            ffi::trp_preprocess_batch_fold(
                *guard,
                bytes.as_ptr(),
                offsets.as_ptr(),
                texts.len() as u64,
                &cfg,
                more.as_ptr(),
                more.len() as u32,
                ffi::TRP_OUT_CHUNKS | ffi::TRP_OUT_CHUNK_TEXT,
                &mut raw,
                &mut ct,
            )
        };
        drop(guard);
        if rc != ffi::TRP_OK {
            let e = last_error();
            log_error("trp_preprocess_batch_fold");
            return Err(e);
        }
        Ok((Batch { raw }, ChunkText { raw: ct }))
    }

    /// Chunk prototypes of section `r` from the descriptors and the GPU-rendered chunk text.
    fn chunks_with_text(&self, r: usize, ct: &ChunkText, summary_weight: f32) -> Vec<Proto> {
        let res = &self.raw;
        let n = res.n_records as usize;
        let ch_off = Self::slice(res.chunk_offsets as *const u64, n + 1);
        let chunks = Self::slice(res.chunks as *const ffi::trp_chunk, res.n_chunks as usize);
        let t_off = Self::slice(ct.raw.offsets as *const u64, res.n_chunks as usize + 1);
        let text = Self::slice(ct.raw.text as *const u8, ct.raw.n_bytes as usize);
        (ch_off[r] as usize..ch_off[r + 1] as usize)
            .map(|i| {
                let c = &chunks[i];
                let view = if c.kind == 0 {
                    ChunkView::Window { index: c.index as usize, overlap: c.overlap as usize, span: c.span as usize }
                } else {
                    ChunkView::SummaryFallback { strategy: "head".into(), weight: summary_weight }
                };
                let s = String::from_utf8_lossy(&text[t_off[i] as usize..t_off[i + 1] as usize]).into_owned();
                (view, s, (c.tok_end - c.tok_begin) as usize)
            })
            .collect()
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
fn strategy_fingerprint(st: &ChunkingStrategy, den: Option<&DenoiserConfig>, more: &[DenoiserConfig]) -> u64 {
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
    for d in more.iter().filter(|d| d.enabled) {
        mix(0x100 | d.strip_markdown as u64);
        mix(d.max_digit_ratio.to_bits() as u64);
    }
    h
}

/// 64-bit hash of a section text (eight bytes per step, final avalanche): an entry is only served for the
/// text it was made from.
fn text_hash(t: &str) -> u64 {
    let b = t.as_bytes();
    let mut h: u64 = 0x9E37_79B9_7F4A_7C15 ^ b.len() as u64;
    let mut chunks = b.chunks_exact(8);
    for c in &mut chunks {
        let v = u64::from_le_bytes(c.try_into().unwrap());
        h = (h ^ v).wrapping_mul(0x0000_0100_0000_01B3);
        h ^= h >> 29;
    }
    let mut tail = [0u8; 8];
    let rest = chunks.remainder();
    tail[..rest.len()].copy_from_slice(rest);
    h = (h ^ u64::from_le_bytes(tail)).wrapping_mul(0x0000_0100_0000_01B3);
    h ^= h >> 32;
    h = h.wrapping_mul(0xD6E8_FEB8_6659_FD93);
    h ^ (h >> 32)
}

struct Entry {
    protos: Arc<Vec<Proto>>,
    text_hash: u64,
    text_len: usize,
    cost: usize,
    serial: u64,
}

#[derive(Default)]
struct Table {
    map: HashMap<Key, Entry>,
    raw_tokens: HashMap<(String, usize), usize>,
    order: VecDeque<(u64, Key)>, // insertion order: the rolling part (src/ingestion.rs:18)
    bytes: usize,
    serial: u64,
}

/// A section that missed the table, shared between the thread that waits for it and the thread that computes it.
struct Miss {
    key: Key,
    text: String,
    strategy: ChunkingStrategy,
    slot: Mutex<Option<Result<Arc<Vec<Proto>>, String>>>,
}

/// `ChunkingAlgorithm` served from GPU batches.  Build it with the denoiser settings it should apply
/// (do NOT also register a `DenoiserPreprocessor` on the strategy: the GPU path already denoises).
///
/// The reference's `materialize` is a function of the section text it is handed
/// (src/chunking/sliding_window.rs:18): an entry remembers the length and a hash of the text it was made from,
/// and a lookup with another text under the same id is a miss that replaces it.  The table keeps at most
/// `budget_bytes` of chunk text and drops the entries inserted first beyond that; `invalidate` drops a record.
///
/// A strategy that carries registered preprocessors (`strategy.preprocessors()` non-empty) cannot be
/// served: the trait objects are opaque, the GPU path implements the denoiser only (one, or a fold of several:
/// `with_denoisers`), and running arbitrary host preprocessors would need the reference's CPU chunker.  Such a call records an error, logs it and returns no chunks;
/// give such samplers the reference's own `SlidingWindowChunker`.
pub struct GpuSlidingWindowChunker {
    ctx: Context,
    denoiser: Option<DenoiserConfig>,
    /// denoisers registered after the first one: the text is folded through all of them in order
    more: Vec<DenoiserConfig>,
    table: RwLock<Table>,
    budget_bytes: usize,
    queue: Mutex<Vec<Arc<Miss>>>,
    gpu: Mutex<()>,
    error: Mutex<Option<String>>,
}

pub const DEFAULT_BUDGET_BYTES: usize = 1 << 30;

impl GpuSlidingWindowChunker {
    pub fn new(device: i32, denoiser: Option<DenoiserConfig>) -> Result<Self, String> {
        Self::with_budget(device, denoiser, DEFAULT_BUDGET_BYTES)
    }

    /// A chunker for a strategy that would register several `DenoiserPreprocessor`s
    /// (`ChunkingStrategy::register_preprocessor`, src/config.rs:105-113): the section text is folded through
    /// them in this order on the GPU, and a section is dropped as soon as one returns `None`
    /// (src/chunking/sliding_window.rs:20-34).
    pub fn with_denoisers(device: i32, denoisers: Vec<DenoiserConfig>, budget_bytes: usize) -> Result<Self, String> {
        let mut it = denoisers.into_iter();
        let first = it.next();
        let mut me = Self::with_budget(device, first, budget_bytes)?;
        me.more = it.collect();
        Ok(me)
    }

    pub fn with_budget(device: i32, denoiser: Option<DenoiserConfig>, budget_bytes: usize) -> Result<Self, String> {
        Ok(Self {
            ctx: Context::new(device)?,
            denoiser,
            more: Vec::new(),
            table: RwLock::new(Table::default()),
            budget_bytes,
            queue: Mutex::new(Vec::new()),
            gpu: Mutex::new(()),
            error: Mutex::new(None),
        })
    }

    /// The first GPU failure since the last call, if any (the traits cannot return it).
    pub fn take_error(&self) -> Option<String> {
        self.error.lock().unwrap_or_else(|p| p.into_inner()).take()
    }

    fn fail(&self, e: String) -> String {
        let mut slot = self.error.lock().unwrap_or_else(|p| p.into_inner());
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
        let _gpu = self.gpu.lock().unwrap_or_else(|p| p.into_inner());
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

    /// `gpu` held.  One batch for the chunks - descriptors and chunk text rendered on the GPU, 1.03 bytes per
    /// input byte over the link instead of the 2.44 of all arrays - and, with a denoiser, one pass-through batch
    /// for the raw token counts.  Returns the prototypes it made, in the order of `keys`.
    fn compute(
        &self,
        strategy: &ChunkingStrategy,
        keys: &[(String, usize)],
        texts: &[&str],
    ) -> Result<Vec<Arc<Vec<Proto>>>, String> {
        let (batch, chunk_text) = Batch::run_with_chunk_text(&self.ctx, texts, self.denoiser.as_ref(), &self.more, strategy)?;
        let denoising = self.denoiser.as_ref().map_or(false, |d| d.enabled) || self.more.iter().any(|d| d.enabled);
        let raw: Vec<usize> = if denoising {
            let plain = Batch::run(&self.ctx, texts, None, &[], strategy, 0)?;
            (0..texts.len()).map(|r| plain.token_count(r)).collect()
        } else {
            (0..texts.len()).map(|r| batch.token_count(r)).collect()
        };
        let fp = strategy_fingerprint(strategy, self.denoiser.as_ref(), &self.more);
        let protos: Vec<Arc<Vec<Proto>>> = (0..texts.len())
            .map(|r| Arc::new(batch.chunks_with_text(r, &chunk_text, strategy.summary_fallback_weight)))
            .collect();
        let mut t = self.table.write().unwrap_or_else(|p| p.into_inner());
        for (((key, p), n), text) in keys.iter().zip(protos.iter()).zip(raw).zip(texts) {
            let k: Key = (fp, key.0.clone(), key.1);
            if let Some(old) = t.map.remove(&k) {
                t.bytes -= old.cost;
            }
            let cost = std::mem::size_of::<Entry>()
                + key.0.len()
                + p.iter().map(|c| std::mem::size_of::<Proto>() + c.1.len()).sum::<usize>();
            t.serial += 1;
            let serial = t.serial;
            t.bytes += cost;
            t.order.push_back((serial, k.clone()));
            t.map.insert(k, Entry { protos: p.clone(), text_hash: text_hash(text), text_len: text.len(), cost, serial });
            t.raw_tokens.insert(key.clone(), n);
        }
        // the rolling part: the oldest entries go while the table is over its budget (the newest one stays)
        while t.bytes > self.budget_bytes && t.order.len() > 1 {
            let (serial, k) = t.order.pop_front().unwrap();
            let stale = t.map.get(&k).map_or(true, |e| e.serial != serial);
            if stale {
                continue; // replaced or invalidated meanwhile
            }
            let e = t.map.remove(&k).unwrap();
            t.bytes -= e.cost;
            t.raw_tokens.remove(&(k.1, k.2));
        }
        Ok(protos)
    }

    /// Token count of the raw text of a section that is in the table.
    pub fn token_count(&self, record_id: &str, section_idx: usize) -> Option<usize> {
        let t = self.table.read().unwrap_or_else(|p| p.into_inner());
        t.raw_tokens.get(&(record_id.to_string(), section_idx)).copied()
    }

    /// Bytes of chunk text (and bookkeeping) the table holds; never more than the budget after an insertion.
    pub fn bytes(&self) -> usize {
        self.table.read().unwrap_or_else(|p| p.into_inner()).bytes
    }

    /// Drops every entry of one record (all sections, all strategies).
    pub fn invalidate(&self, record_id: &str) -> usize {
        let mut t = self.table.write().unwrap_or_else(|p| p.into_inner());
        let gone: Vec<Key> = t.map.keys().filter(|k| k.1 == record_id).cloned().collect();
        for k in &gone {
            if let Some(e) = t.map.remove(k) {
                t.bytes -= e.cost;
            }
        }
        t.raw_tokens.retain(|k, _| k.0 != record_id);
        gone.len()
    }

    pub fn clear(&self) {
        let mut t = self.table.write().unwrap_or_else(|p| p.into_inner());
        *t = Table::default();
    }

    /// `gpu` held: everything queued right now, one batch per strategy fingerprint.  Every queued miss gets its
    /// answer (or the error) through its own slot, whether or not the entry is still in the table afterwards.
    fn drain(&self) {
        let mut work: Vec<Arc<Miss>> = std::mem::take(&mut *self.queue.lock().unwrap_or_else(|p| p.into_inner()));
        while let Some(first) = work.first() {
            let fp = first.key.0;
            let (same, rest): (Vec<_>, Vec<_>) = work.into_iter().partition(|m| m.key.0 == fp);
            let keys: Vec<(String, usize)> = same.iter().map(|m| (m.key.1.clone(), m.key.2)).collect();
            let texts: Vec<&str> = same.iter().map(|m| m.text.as_str()).collect();
            match self.compute(&same[0].strategy, &keys, &texts) {
                Ok(protos) => {
                    for (m, p) in same.iter().zip(protos) {
                        *m.slot.lock().unwrap_or_else(|e| e.into_inner()) = Some(Ok(p));
                    }
                }
                Err(e) => {
                    self.fail(e.clone());
                    for m in &same {
                        *m.slot.lock().unwrap_or_else(|p| p.into_inner()) = Some(Err(e.clone()));
                    }
                }
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
        let key: Key = (strategy_fingerprint(strategy, self.denoiser.as_ref(), &self.more), record.id.to_string(), section_idx);
        let th = text_hash(&section.text);
        let hit = {
            let t = self.table.read().unwrap_or_else(|p| p.into_inner());
            t.map
                .get(&key)
                .filter(|e| e.text_len == section.text.len() && e.text_hash == th)
                .map(|e| e.protos.clone())
        };
        let protos = match hit {
            Some(p) => Some(p),
            None => {
                // miss, or the text under this id has changed: queue it (the queue owns its copy of the text and
                // of the strategy, so a panic elsewhere leaves nothing dangling); the first thread to get the GPU
                // computes everything queued
                let me = Arc::new(Miss {
                    key: key.clone(),
                    text: section.text.clone(),
                    strategy: strategy.clone(),
                    slot: Mutex::new(None),
                });
                self.queue.lock().unwrap_or_else(|p| p.into_inner()).push(me.clone());
                {
                    let _gpu = self.gpu.lock().unwrap_or_else(|p| p.into_inner());
                    if me.slot.lock().unwrap_or_else(|p| p.into_inner()).is_none() {
                        self.drain();
                    }
                }
                // a GPU failure leaves no answer: the error is kept (take_error) and logged
                match me.slot.lock().unwrap_or_else(|p| p.into_inner()).take() {
                    Some(Ok(p)) => Some(p),
                    _ => None,
                }
            }
        };
        protos
            .map(|p| p.as_ref().clone())
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

// ---- the steps right after the path -----------------------------------------------------------------------

fn pack(texts: &[&str]) -> (Vec<u8>, Vec<u64>) {
    let mut bytes = Vec::with_capacity(texts.iter().map(|t| t.len()).sum());
    let mut offsets = Vec::with_capacity(texts.len() + 1);
    offsets.push(0u64);
    for t in texts {
        bytes.extend_from_slice(t.as_bytes());
        offsets.push(bytes.len() as u64);
    }
    (bytes, offsets)
}

fn take_texts(ct: &mut ffi::trp_chunk_text, n: usize) -> Vec<String> {
    let offs = Batch::slice(ct.offsets as *const u64, n + 1);
    let text = Batch::slice(ct.text as *const u8, ct.n_bytes as usize);
    let out = (0..n)
        .map(|i| String::from_utf8_lossy(&text[offs[i] as usize..offs[i + 1] as usize]).into_owned())
        .collect();
    unsafe { ffi::trp_chunk_text_free(ct) };
    out
}

/// `record_bm25_text` (src/sampler/backends/bm25_backend.rs:503-532) for many records in one GPU batch.
/// `records`: (record id, [(heading, section text)]) - what the reference reads from a `DataRecord`.
pub fn record_bm25_texts(
    ctx: &Context,
    records: &[(&str, Vec<(Option<&str>, &str)>)],
    max_tokens: usize,
    newline: &str,
) -> Result<Vec<String>, String> {
    let mut parts: Vec<&str> = Vec::new();
    let mut doc: Vec<u64> = vec![0];
    for (_, sections) in records {
        for (heading, text) in sections {
            if let Some(h) = heading {
                parts.push(h);
            }
            parts.push(text);
        }
        doc.push(parts.len() as u64);
    }
    let (bytes, offsets) = pack(&parts);
    let nl = std::ffi::CString::new(newline).map_err(|e| e.to_string())?;
    let mut ct: ffi::trp_chunk_text = unsafe { std::mem::zeroed() };
    let guard = ctx.raw.lock().unwrap_or_else(|p| p.into_inner());
    let rc = unsafe {
        ffi::trp_bm25_texts(
            *guard,
            bytes.as_ptr(),
            offsets.as_ptr(),
            parts.len() as u64,
            doc.as_ptr(),
            records.len() as u64,
            max_tokens.min(u32::MAX as usize) as u32,
            nl.as_ptr(),
            &mut ct,
        )
    };
    drop(guard);
    if rc != ffi::TRP_OK {
        return Err(last_error());
    }
    let mut out = take_texts(&mut ct, records.len());
    for (t, (id, _)) in out.iter_mut().zip(records) {
        if t.is_empty() {
            *t = (*id).to_string(); // bm25_backend.rs:517-519
        }
    }
    Ok(out)
}

/// The BM25 query cap (bm25_backend.rs:219-225): every text cut to its first `limit` tokens joined by one space,
/// or left as it is when it has no more than that.
pub fn cap_tokens(ctx: &Context, texts: &[&str], limit: usize) -> Result<Vec<String>, String> {
    let (bytes, offsets) = pack(texts);
    let mut ct: ffi::trp_chunk_text = unsafe { std::mem::zeroed() };
    let guard = ctx.raw.lock().unwrap_or_else(|p| p.into_inner());
    let rc = unsafe {
        ffi::trp_head_tokens_text(
            *guard,
            bytes.as_ptr(),
            offsets.as_ptr(),
            texts.len() as u64,
            limit.min(u32::MAX as usize) as u32,
            &mut ct,
        )
    };
    drop(guard);
    if rc != ffi::TRP_OK {
        return Err(last_error());
    }
    Ok(take_texts(&mut ct, texts.len()))
}

/// The body half of `decorate_chunk` (src/sampler/mod.rs:1497-1504) for many chunks in one GPU batch: the
/// sections are preprocessed and chunked with `strategy` (and the denoiser, if any), chunk `chunk_index[i]` of
/// the batch is cut to `budget[i]` tokens (`max_window_tokens - prefix_tokens`), verbatim when the budget covers
/// it.  The prefix comes from the sampler's RNG (`KvpPrefixSampler::sample`) and is put in front by the caller.
pub fn chunk_bodies(
    ctx: &Context,
    sections: &[&str],
    denoiser: Option<&DenoiserConfig>,
    strategy: &ChunkingStrategy,
    chunk_index: &[u32],
    budget: &[u32],
) -> Result<Vec<String>, String> {
    if chunk_index.len() != budget.len() {
        return Err("chunk_index and budget differ in length".into());
    }
    let (bytes, offsets) = pack(sections);
    let overlaps: Vec<u32> = strategy.overlap_tokens.iter().map(|&o| o.min(u32::MAX as usize) as u32).collect();
    let cfg = Batch::config(denoiser, strategy, &overlaps);
    let mut ct: ffi::trp_chunk_text = unsafe { std::mem::zeroed() };
    let guard = ctx.raw.lock().unwrap_or_else(|p| p.into_inner());
    let rc = unsafe {
        ffi::trp_chunk_bodies(
            *guard,
            bytes.as_ptr(),
            offsets.as_ptr(),
            sections.len() as u64,
            &cfg,
            chunk_index.as_ptr(),
            budget.as_ptr(),
            chunk_index.len() as u64,
            &mut ct,
        )
    };
    drop(guard);
    if rc != ffi::TRP_OK {
        return Err(last_error());
    }
    Ok(take_texts(&mut ct, chunk_index.len()))
}

/// One batch of sections over several GPUs of this process: one context and one thread per entry of `devices`,
/// byte-balanced contiguous ranges of sections, nothing exchanged between the GPUs.  Returns, per shard, the
/// first section it holds and its chunk prototypes per section.
pub fn preprocess_sharded(
    devices: &[i32],
    texts: &[&str],
    denoiser: Option<&DenoiserConfig>,
    strategy: &ChunkingStrategy,
) -> Result<Vec<(usize, Vec<Vec<Proto>>)>, String> {
    if devices.is_empty() {
        return Err("preprocess_sharded needs at least one device".into());
    }
    let total: u64 = texts.iter().map(|t| t.len() as u64).sum();
    let g = devices.len();
    let mut cuts = vec![0usize; g + 1];
    cuts[g] = texts.len();
    let (mut at, mut seen) = (0usize, 0u64);
    for k in 1..g {
        let target = total * k as u64 / g as u64;
        while at < texts.len() && seen < target {
            seen += texts[at].len() as u64;
            at += 1;
        }
        cuts[k] = at;
    }
    let ctxs: Vec<Context> = devices.iter().map(|&d| Context::new(d)).collect::<Result<_, _>>()?;
    std::thread::scope(|scope| {
        let handles: Vec<_> = (0..g)
            .map(|k| {
                let (ctx, part) = (&ctxs[k], &texts[cuts[k]..cuts[k + 1]]);
                scope.spawn(move || -> Result<Vec<Vec<Proto>>, String> {
                    let (batch, chunk_text) = Batch::run_with_chunk_text(ctx, part, denoiser, &[], strategy)?;
                    Ok((0..part.len())
                        .map(|r| batch.chunks_with_text(r, &chunk_text, strategy.summary_fallback_weight))
                        .collect())
                })
            })
            .collect();
        let mut out = Vec::with_capacity(g);
        for (k, h) in handles.into_iter().enumerate() {
            let protos = h.join().map_err(|_| "a shard thread panicked".to_string())??;
            out.push((cuts[k], protos));
        }
        Ok(out)
    })
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
