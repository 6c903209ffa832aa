"""Host-side mirror (Python, ctypes) of the reference's denoiser / tokenizer / chunker surface on top
of the C ABI of libtriplets_amd.so (HIP kernels for gfx950).

Names and argument meaning follow the reference crate (paths relative to its checkout):
  DenoiserConfig, ChunkingStrategy                       src/config.rs:12-119
  denoise_text, DenoiserPreprocessor.process             src/preprocessor/backends/denoiser_preprocessor.rs:221, :303
  WhitespaceTokenizer.tokenize / token_count             src/tokenizer.rs:55-65
  SlidingWindowChunker.materialize                       src/chunking/sliding_window.rs:11-117
plus the batched entry point `preprocess_batch` (many records in one packed buffer).

There is no CPU fallback: importing works anywhere, but every call needs the built shared library
and a GPU, and raises TripletsAmdError otherwise.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libtriplets_amd.so")
if os.environ.get("TRIPLETS_AMD_LIB"):   # another build of the same library (kernel experiments)
    LIB_PATH = os.environ["TRIPLETS_AMD_LIB"]

TRP_OK = 0


class TripletsAmdError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("denoise_enabled", C.c_uint8), ("strip_markdown", C.c_uint8),
                ("max_digit_ratio", C.c_float), ("max_window_tokens", C.c_uint32),
                ("overlaps", C.POINTER(C.c_uint32)), ("n_overlaps", C.c_uint32),
                ("summary_fallback_tokens", C.c_uint32)]


class _Chunk(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("record", "kind", "index", "overlap", "span", "tok_begin", "tok_end", "reserved")]


def _copy_out(ptr: int, nbytes: int, dtype) -> np.ndarray:
    """One copy of `nbytes` of library-owned host memory at `ptr` as a numpy array the caller owns."""
    return np.frombuffer((C.c_ubyte * nbytes).from_address(ptr), dtype=dtype).copy()


class _Owner:
    """Frees one library-owned host array when the last numpy view of it goes away."""
    __slots__ = ("_free", "_ptr")

    def __init__(self, free, ptr):
        self._free, self._ptr = free, ptr

    def __del__(self):
        try:
            self._free(self._ptr)
        except Exception:
            pass


_ADOPT_MIN = 64 << 10


def _take_out(lib, ptr: int, nbytes: int, dtype) -> Tuple[np.ndarray, bool]:
    """A library-owned host array as a numpy array: arrays of 64 KiB and more are handed over as they are (the
    array's base keeps the memory and gives it back with trp_free when collected; True is returned and the caller
    must not free `ptr`), smaller ones are copied."""
    if nbytes < _ADOPT_MIN:
        return _copy_out(ptr, nbytes, dtype), False
    buf = (C.c_ubyte * nbytes).from_address(ptr)
    buf._owner = _Owner(lib.trp_free, ptr)
    return np.frombuffer(buf, dtype=dtype), True


class _Result(C.Structure):
    _fields_ = [("denoised", C.c_void_p), ("denoised_offsets", C.c_void_p), ("dropped", C.c_void_p),
                ("tok_start", C.c_void_p), ("tok_len", C.c_void_p), ("tok_offsets", C.c_void_p),
                ("chunks", C.c_void_p), ("chunk_offsets", C.c_void_p),
                ("n_records", C.c_uint64), ("n_denoised", C.c_uint64), ("n_tokens", C.c_uint64),
                ("n_chunks", C.c_uint64)]


class _Denoiser(C.Structure):
    _fields_ = [("denoise_enabled", C.c_uint8), ("strip_markdown", C.c_uint8), ("max_digit_ratio", C.c_float)]


class _ChunkText(C.Structure):
    _fields_ = [("text", C.c_void_p), ("offsets", C.c_void_p), ("n_bytes", C.c_uint64)]


class _OutBuffers(C.Structure):
    _fields_ = [("denoised", C.c_void_p), ("denoised_cap", C.c_uint64), ("denoised_offsets", C.c_void_p),
                ("dropped", C.c_void_p), ("tok_start", C.c_void_p), ("tok_len", C.c_void_p),
                ("tok_cap", C.c_uint64), ("tok_offsets", C.c_void_p), ("chunks", C.c_void_p),
                ("chunk_cap", C.c_uint64), ("chunk_offsets", C.c_void_p)]


NOALLOC_MAX_BYTES, NOALLOC_MAX_RECORDS = 32 << 20, 1 << 19
OUT_TEXT, OUT_TOKENS, OUT_CHUNKS, OUT_CHUNK_TEXT, OUT_ALL = 1, 2, 4, 8, 7


class _KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_float)]


CHUNK_DTYPE = np.dtype([(n, "<u4") for n in
                        ("record", "kind", "index", "overlap", "span", "tok_begin", "tok_end",
                         "reserved")])

EXPORTS = ["trp_create", "trp_destroy", "trp_last_error", "trp_stream", "trp_preprocess_batch",
           "trp_result_free", "trp_preprocess_batch_device", "trp_copy_to_host",
           "trp_materialize_chunk_text_device", "trp_denoise", "trp_tokenize", "trp_token_count",
           "trp_chunk_section", "trp_free", "trp_set_profiling", "trp_get_profile", "trp_trim_host_pool", "trp_normalize_batch",
           "trp_normalize_batch_device", "trp_normalize_inline_whitespace", "trp_head_tokens_text_device",
           "trp_submit_batch_device", "trp_wait_batch", "trp_create_on_stream", "trp_wait_event", "trp_chunk_bodies_device", "trp_bm25_texts_device", "trp_preprocess_batch_select",
           "trp_chunk_text_free", "trp_token_bound", "trp_preprocess_batch_device_noalloc", "trp_device_count", "trp_bm25_texts", "trp_chunk_bodies", "trp_head_tokens_text",
           "trp_preprocess_batch_fold", "trp_preprocess_batch_device_fold"]

_lib = None


def load_library():
    """Load libtriplets_amd.so (built in-tree by `make -C rust-triplets_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (same SONAME).  Two HIP runtimes in one process cannot both
    # drive the GPU, so when torch is installed it is imported first and the library binds to the
    # runtime torch loaded; without torch the system runtime under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise TripletsAmdError("HIP library not built: %s is missing (run __graft_entry__.build())"
                               % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    if os.environ.get("TRIPLETS_AMD_LIB"):
        # an older build under test may lack the newest entry points: they then fail when called, not here
        class _Missing:
            def __init__(self, name):
                self.name = name

            def __call__(self, *a):
                raise TripletsAmdError("%s is not exported by %s" % (self.name, LIB_PATH))
        for name in EXPORTS:
            if not hasattr(L, name):
                setattr(L, name, _Missing(name))
    L.trp_last_error.restype = C.c_char_p
    L.trp_device_count.restype = C.c_int
    L.trp_device_count.argtypes = []
    L.trp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.trp_create_on_stream.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.trp_wait_event.argtypes = [C.c_void_p, C.c_void_p]
    L.trp_destroy.argtypes = [C.c_void_p]
    L.trp_stream.restype = C.c_void_p
    L.trp_stream.argtypes = [C.c_void_p]
    L.trp_preprocess_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.POINTER(_Config), C.POINTER(_Result)]
    L.trp_result_free.argtypes = [C.POINTER(_Result)]
    L.trp_preprocess_batch_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_Config),
                                            C.POINTER(_Denoiser), C.c_uint32, C.c_uint32, C.POINTER(_Result),
                                            C.POINTER(_ChunkText)]
    L.trp_preprocess_batch_device_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                   C.POINTER(_Config), C.POINTER(_Denoiser), C.c_uint32,
                                                   C.POINTER(_Result)]
    L.trp_preprocess_batch_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_Config),
                                              C.c_uint32, C.POINTER(_Result), C.POINTER(_ChunkText)]
    L.trp_chunk_text_free.argtypes = [C.POINTER(_ChunkText)]
    L.trp_preprocess_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.c_uint64, C.POINTER(_Config), C.POINTER(_Result)]
    L.trp_submit_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                          C.POINTER(_Config)]
    L.trp_wait_batch.argtypes = [C.c_void_p, C.POINTER(_Result)]
    L.trp_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.trp_materialize_chunk_text_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.trp_denoise.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(_Config),
                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.trp_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.trp_token_count.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.trp_chunk_section.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(_Config),
                                    C.POINTER(_Result)]
    L.trp_free.argtypes = [C.c_void_p]
    L.trp_token_bound.restype = C.c_uint64
    L.trp_token_bound.argtypes = [C.c_uint64, C.c_uint64]
    L.trp_preprocess_batch_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                      C.POINTER(_Config), C.POINTER(_OutBuffers), C.POINTER(_Result)]
    L.trp_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.trp_get_profile.argtypes = [C.c_void_p, C.POINTER(_KernelTime), C.c_int]
    L.trp_head_tokens_text_device.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.trp_chunk_bodies_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.trp_bm25_texts_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.trp_bm25_texts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.c_char_p, C.POINTER(_ChunkText)]
    L.trp_head_tokens_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(_ChunkText)]
    L.trp_chunk_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_Config), C.c_void_p,
                                   C.c_void_p, C.c_uint64, C.POINTER(_ChunkText)]
    L.trp_normalize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(_Result)]
    L.trp_normalize_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                             C.POINTER(_Result)]
    L.trp_normalize_inline_whitespace.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
                                                  C.POINTER(C.c_size_t)]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise TripletsAmdError("triplets_amd error %d: %s" % (rc, load_library().trp_last_error().decode()))
    return rc


# ---------------------------------------------------------------------------------------------------
# configuration types (src/config.rs)
# ---------------------------------------------------------------------------------------------------
@dataclass
class DenoiserConfig:
    enabled: bool = False
    max_digit_ratio: float = 0.35
    strip_markdown: bool = True


@dataclass
class ChunkingStrategy:
    max_window_tokens: int = 1024
    overlap_tokens: List[int] = field(default_factory=lambda: [64])
    summary_fallback_weight: float = 0.35
    summary_fallback_tokens: int = 512
    chunk_weight_floor: float = 0.1
    preprocessors: list = field(default_factory=list)

    def register_preprocessor(self, p):
        self.preprocessors.append(p)
        return self


_CFG_CACHE = {}


def _make_cfg(den: DenoiserConfig, st: ChunkingStrategy):
    """The C configuration of a (denoiser, strategy) pair; kept per distinct setting (the library only reads it),
    which takes two microseconds off every call of a loop over small batches."""
    ov = tuple(st.overlap_tokens)
    key = (bool(den.enabled), bool(den.strip_markdown), float(den.max_digit_ratio), int(st.max_window_tokens), ov,
           int(st.summary_fallback_tokens))
    if key[2] != key[2]:   # NaN never equals itself: not cached
        key = None
    cfg = _CFG_CACHE.get(key) if key is not None else None
    if cfg is None:
        arr = (C.c_uint32 * max(1, len(ov)))(*ov)
        cfg = _Config(int(den.enabled), int(den.strip_markdown), float(den.max_digit_ratio),
                      int(st.max_window_tokens), C.cast(arr, C.POINTER(C.c_uint32)), len(ov),
                      int(st.summary_fallback_tokens))
        cfg._keep = arr
        if key is not None and len(_CFG_CACHE) < 256:
            _CFG_CACHE[key] = cfg
    return cfg


# ---------------------------------------------------------------------------------------------------
# context and batched entry points
# ---------------------------------------------------------------------------------------------------
class BatchResult:
    """Host copy of one batch result (numpy arrays)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def record_text(self, r) -> bytes:
        return self.denoised[int(self.denoised_offsets[r]):int(self.denoised_offsets[r + 1])].tobytes()

    def tokens(self, r) -> List[bytes]:
        t0, t1 = int(self.tok_offsets[r]), int(self.tok_offsets[r + 1])
        return [self.denoised[int(s):int(s) + int(n)].tobytes()
                for s, n in zip(self.tok_start[t0:t1], self.tok_len[t0:t1])]

    def chunk_text(self, c) -> bytes:
        ch = self.chunks[c]
        rec = int(ch["record"])
        t0 = int(self.tok_offsets[rec])
        nt = int(self.tok_offsets[rec + 1]) - t0
        if (int(self.chunk_offsets[rec + 1] - self.chunk_offsets[rec]) == 1 and ch["kind"] == 0
                and int(ch["tok_end"]) - int(ch["tok_begin"]) == nt):
            return self.record_text(rec)
        parts = []
        for k in range(int(ch["tok_begin"]), int(ch["tok_end"])):
            s = int(self.tok_start[t0 + k])
            parts.append(self.denoised[s:s + int(self.tok_len[t0 + k])].tobytes())
        return b" ".join(parts)


# ---------------------------------------------------------------------------------------------------
# sharding by record range (records are independent: no exchange between shards)
# ---------------------------------------------------------------------------------------------------
def shard_ranges(offsets: np.ndarray, n_shards: int):
    """Byte-balanced contiguous record ranges [(r0, r1), ...], one per shard: shard k starts at the
    first record whose start offset is >= k*N/n_shards."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    R = len(offsets) - 1
    N = int(offsets[-1]) if R >= 0 and len(offsets) else 0
    cuts = [0]
    for k in range(1, n_shards):
        target = (N * k) // n_shards
        cuts.append(int(np.searchsorted(offsets[:R + 1], np.uint64(target), side="left")))
    cuts.append(R)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(n_shards)]


def shard_slice(data: np.ndarray, offsets: np.ndarray, r0: int, r1: int):
    """The packed sub-batch of records [r0, r1): (bytes view, offsets rebased to 0)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    b0, b1 = int(offsets[r0]), int(offsets[r1])
    return data[b0:b1], (offsets[r0:r1 + 1] - np.uint64(b0)).astype(np.uint64)


def concat_results(parts: Sequence["BatchResult"]) -> "BatchResult":
    """Result of the whole batch from the results of its shards, in shard order.  Offsets are shifted
    by the totals of the preceding shards; token starts become u64 because the concatenated text may
    exceed the single-batch limit."""
    den, doff, drop, ts, tl, toff, ch, coff = [], [np.zeros(1, np.uint64)], [], [], [], \
        [np.zeros(1, np.uint64)], [], [np.zeros(1, np.uint64)]
    base_d = base_t = base_c = base_r = 0
    for p in parts:
        den.append(p.denoised)
        doff.append(p.denoised_offsets[1:] + np.uint64(base_d))
        drop.append(p.dropped)
        ts.append(p.tok_start.astype(np.uint64) + np.uint64(base_d))
        tl.append(p.tok_len)
        toff.append(p.tok_offsets[1:] + np.uint64(base_t))
        c = p.chunks.copy()
        c["record"] += np.uint32(base_r)
        ch.append(c)
        coff.append(p.chunk_offsets[1:] + np.uint64(base_c))
        base_d += int(p.n_denoised)
        base_t += int(p.n_tokens)
        base_c += int(p.n_chunks)
        base_r += int(p.n_records)

    def cat(xs, dt):
        return np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dt)

    return BatchResult(denoised=cat(den, np.uint8), denoised_offsets=cat(doff, np.uint64),
                       dropped=cat(drop, np.uint8), tok_start=cat(ts, np.uint64), tok_len=cat(tl, np.uint32),
                       tok_offsets=cat(toff, np.uint64), chunks=cat(ch, CHUNK_DTYPE),
                       chunk_offsets=cat(coff, np.uint64), n_records=base_r, n_denoised=base_d,
                       n_tokens=base_t, n_chunks=base_c)


def gather_shards(local: "BatchResult", group=None) -> Optional["BatchResult"]:
    """Collect the shard results of all ranks of a torch.distributed group on rank 0 (in rank order)
    and concatenate them.  Outside the timed path: this is result plumbing, not a data-path collective.
    Returns the whole-batch result on rank 0 and None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    box = [None] * world if rank == 0 else None
    dist.gather_object(local.__dict__, box, dst=0, group=group)
    if rank != 0:
        return None
    return concat_results([BatchResult(**d) for d in box])


class DeviceResult:
    """Device pointers (ints) + totals of the last device-resident batch of a Context."""

    def __init__(self, ctx, res):
        self.ctx = ctx
        self.denoised = res.denoised or 0
        self.denoised_offsets = res.denoised_offsets or 0
        self.dropped = res.dropped or 0
        self.tok_start = res.tok_start or 0
        self.tok_len = res.tok_len or 0
        self.tok_offsets = res.tok_offsets or 0
        self.chunks = res.chunks or 0
        self.chunk_offsets = res.chunk_offsets or 0
        self.n_records = res.n_records
        self.n_denoised = res.n_denoised
        self.n_tokens = res.n_tokens
        self.n_chunks = res.n_chunks

    def fetch(self, name, count, dtype, offset=0):
        """Copy `count` elements of the device array `name`, starting at element `offset`, to the host."""
        dtype = np.dtype(dtype)
        out = np.empty(int(count), dtype=dtype)
        if count:
            _check(load_library().trp_copy_to_host(self.ctx._h, out.ctypes.data,
                                                   getattr(self, name) + int(offset) * dtype.itemsize, out.nbytes))
        return out

    def to_host(self, denoised_from=None) -> BatchResult:
        R = int(self.n_records)
        return BatchResult(
            denoised=self.fetch("denoised", self.n_denoised, np.uint8),
            denoised_offsets=self.fetch("denoised_offsets", R + 1, np.uint64),
            dropped=self.fetch("dropped", R, np.uint8),
            tok_start=self.fetch("tok_start", self.n_tokens, np.uint32),
            tok_len=self.fetch("tok_len", self.n_tokens, np.uint32),
            tok_offsets=self.fetch("tok_offsets", R + 1, np.uint64),
            chunks=self.fetch("chunks", self.n_chunks, CHUNK_DTYPE),
            chunk_offsets=self.fetch("chunk_offsets", R + 1, np.uint64),
            n_records=R, n_denoised=int(self.n_denoised), n_tokens=int(self.n_tokens),
            n_chunks=int(self.n_chunks))


class Context:
    """One GPU context (one HIP stream).  Not thread-safe: use one per thread.

    stream=None: the context runs on a private non-blocking stream; device inputs produced on another
    stream must be complete (or ordered with wait_event) before a call.  stream=<hipStream_t as int>, e.g.
    torch.cuda.current_stream().cuda_stream (0 is the legacy default stream): the context enqueues on the
    caller's stream, so inputs produced there need no synchronisation."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        L = load_library()
        h = C.c_void_p()
        if stream is None:
            _check(L.trp_create(device, C.byref(h)))
        else:
            _check(L.trp_create_on_stream(device, C.c_void_p(int(stream)), C.byref(h)))
        self._h = h
        self._L = L
        self.device = int(device)
        self._helper: Optional["Context"] = None

    def helper(self) -> "Context":
        """A second context on the same GPU, created on first use and kept: for small side batches (prefixes,
        BM25 documents) that must not replace this context's last device batch."""
        if self._helper is None:
            self._helper = Context(self.device)
        return self._helper

    def wait_event(self, hip_event: int) -> None:
        """Work enqueued by later calls waits for the given hipEvent_t (e.g. a recorded
        torch.cuda.Event().cuda_event); the host does not block."""
        _check(self._L.trp_wait_event(self._h, C.c_void_p(int(hip_event))))

    def close(self):
        if getattr(self, "_helper", None) is not None:
            self._helper.close()
            self._helper = None
        if self._h:
            self._L.trp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return self._L.trp_stream(self._h) or 0

    def _host_result(self, res, R) -> "BatchResult":
        def arr(field, n, dt):
            dt = np.dtype(dt)
            ptr = getattr(res, field)
            if not n or not ptr:
                return np.zeros(0, dtype=dt)
            a, adopted = _take_out(self._L, ptr, int(n) * dt.itemsize, dt)
            if adopted:
                setattr(res, field, None)   # no longer freed with the rest below
            return a

        out = BatchResult(
            denoised=arr("denoised", res.n_denoised, np.uint8),
            denoised_offsets=arr("denoised_offsets", R + 1, np.uint64),
            dropped=arr("dropped", R, np.uint8),
            tok_start=arr("tok_start", res.n_tokens, np.uint32),
            tok_len=arr("tok_len", res.n_tokens, np.uint32),
            tok_offsets=arr("tok_offsets", R + 1, np.uint64),
            chunks=arr("chunks", res.n_chunks, CHUNK_DTYPE),
            chunk_offsets=arr("chunk_offsets", R + 1, np.uint64),
            n_records=R, n_denoised=int(res.n_denoised), n_tokens=int(res.n_tokens),
            n_chunks=int(res.n_chunks))
        self._L.trp_result_free(C.byref(res))
        return out

    def preprocess_batch(self, data: np.ndarray, offsets: np.ndarray, denoiser: DenoiserConfig,
                         strategy: ChunkingStrategy) -> BatchResult:
        """Packed batch in host memory -> host result (copies in and out)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        R = len(offsets) - 1
        cfg = _make_cfg(denoiser, strategy)
        res = _Result()
        _check(self._L.trp_preprocess_batch(self._h, data.ctypes.data if data.size else None,
                                            offsets.ctypes.data, R, C.byref(cfg), C.byref(res)))
        return self._host_result(res, R)

    @staticmethod
    def _fold_args(denoisers: Sequence[DenoiserConfig]):
        """(first denoiser, ctypes array of the others, their number) of a preprocessor fold."""
        denoisers = list(denoisers)
        first = denoisers[0] if denoisers else DenoiserConfig(False, 0.35, True)
        rest = denoisers[1:]
        arr = (_Denoiser * max(1, len(rest)))(*[_Denoiser(int(d.enabled), int(d.strip_markdown), float(d.max_digit_ratio))
                                                for d in rest])
        return first, arr, len(rest)

    def preprocess_batch_fold(self, data: np.ndarray, offsets: np.ndarray, denoisers: Sequence[DenoiserConfig],
                              strategy: ChunkingStrategy) -> BatchResult:
        """preprocess_batch for a strategy with several registered denoisers: the section texts are folded through
        them in order (src/chunking/sliding_window.rs:20-34), a section is dropped as soon as one returns None."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        R = len(offsets) - 1
        first, arr, n = self._fold_args(denoisers)
        cfg = _make_cfg(first, strategy)
        res = _Result()
        _check(self._L.trp_preprocess_batch_fold(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data, R,
                                                 C.byref(cfg), arr, n, OUT_ALL, C.byref(res), None))
        return self._host_result(res, R)

    def preprocess_batch_device_fold(self, d_bytes: int, d_offsets: int, n_records: int, n_bytes: int,
                                     denoisers: Sequence[DenoiserConfig], strategy: ChunkingStrategy) -> "DeviceResult":
        first, arr, n = self._fold_args(denoisers)
        cfg = _make_cfg(first, strategy)
        res = _Result()
        _check(self._L.trp_preprocess_batch_device_fold(self._h, d_bytes, d_offsets, n_records, n_bytes, C.byref(cfg),
                                                        arr, n, C.byref(res)))
        return DeviceResult(self, res)

    def preprocess_batch_select(self, data: np.ndarray, offsets: np.ndarray, denoiser: DenoiserConfig,
                                strategy: ChunkingStrategy, outputs: int) -> BatchResult:
        """preprocess_batch returning only the arrays named by `outputs` (OUT_* flags); arrays that were not
        selected are empty.  With OUT_CHUNK_TEXT the result has chunk_text (uint8) and chunk_text_offsets."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        R = len(offsets) - 1
        cfg = _make_cfg(denoiser, strategy)
        res, ct = _Result(), _ChunkText()
        _check(self._L.trp_preprocess_batch_select(self._h, data.ctypes.data if data.size else None,
                                                   offsets.ctypes.data, R, C.byref(cfg), int(outputs), C.byref(res),
                                                   C.byref(ct) if outputs & OUT_CHUNK_TEXT else None))
        n_chunks = int(res.n_chunks)
        text = offs = None
        if outputs & OUT_CHUNK_TEXT:
            offs, kept = _take_out(self._L, ct.offsets, (n_chunks + 1) * 8, np.uint64)
            if kept:
                ct.offsets = None
            text = np.zeros(0, np.uint8)
            if ct.n_bytes:
                text, kept = _take_out(self._L, ct.text, int(ct.n_bytes), np.uint8)
                if kept:
                    ct.text = None
            self._L.trp_chunk_text_free(C.byref(ct))
        out = self._host_result(res, R)
        out.chunk_text_bytes, out.chunk_text_offsets = text, offs
        return out

    def normalize_batch(self, data: np.ndarray, offsets: np.ndarray) -> BatchResult:
        """normalize_inline_whitespace (src/utils.rs:18-33) of every record of a packed host batch."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        R = len(offsets) - 1
        res = _Result()
        _check(self._L.trp_normalize_batch(self._h, data.ctypes.data if data.size else None,
                                           offsets.ctypes.data, R, C.byref(res)))
        return self._host_result(res, R)

    def normalize_batch_device(self, d_bytes: int, d_offsets: int, n_records: int, n_bytes: int) -> "DeviceResult":
        res = _Result()
        _check(self._L.trp_normalize_batch_device(self._h, d_bytes, d_offsets, n_records, n_bytes, C.byref(res)))
        return DeviceResult(self, res)

    def preprocess_batch_device(self, d_bytes: int, d_offsets: int, n_records: int, n_bytes: int,
                                denoiser: DenoiserConfig, strategy: ChunkingStrategy) -> DeviceResult:
        """Packed batch already in GPU memory (raw device pointers) -> device-resident result."""
        cfg = _make_cfg(denoiser, strategy)
        res = _Result()
        _check(self._L.trp_preprocess_batch_device(self._h, d_bytes, d_offsets, n_records, n_bytes,
                                                   C.byref(cfg), C.byref(res)))
        return DeviceResult(self, res)

    def preprocess_batch_device_noalloc(self, d_bytes: int, d_offsets: int, n_records: int, n_bytes: int,
                                        denoiser: DenoiserConfig, strategy: ChunkingStrategy,
                                        buffers: "_OutBuffers") -> DeviceResult:
        """preprocess_batch_device that allocates nothing: the working arrays are the library's fixed scratch
        array, the result arrays the caller's device buffers described by `buffers` (at most NOALLOC_MAX_BYTES
        bytes and NOALLOC_MAX_RECORDS records per call; calls that overlap on one GPU race)."""
        cfg = _make_cfg(denoiser, strategy)
        res = _Result()
        _check(self._L.trp_preprocess_batch_device_noalloc(self._h, d_bytes, d_offsets, n_records, n_bytes,
                                                           C.byref(cfg), C.byref(buffers), C.byref(res)))
        return DeviceResult(self, res)

    def submit_batch_device(self, d_bytes: int, d_offsets: int, n_records: int, n_bytes: int,
                            denoiser: DenoiserConfig, strategy: ChunkingStrategy) -> None:
        """Enqueue the path for a device-resident batch and return at once; wait_batch() gives the result.
        One batch per context may be in flight."""
        cfg = _make_cfg(denoiser, strategy)
        self._cfg_in_flight = cfg   # the overlaps array must outlive the call
        _check(self._L.trp_submit_batch_device(self._h, d_bytes, d_offsets, n_records, n_bytes, C.byref(cfg)))

    def wait_batch(self) -> DeviceResult:
        res = _Result()
        _check(self._L.trp_wait_batch(self._h, C.byref(res)))
        return DeviceResult(self, res)

    def materialize_chunk_text(self):
        """Chunk texts of the last device batch -> (bytes array, offsets[n_chunks+1]) on the host."""
        t, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.trp_materialize_chunk_text_device(self._h, C.byref(t), C.byref(o), C.byref(n)))
        return t.value or 0, o.value or 0, int(n.value)

    def head_tokens_text(self, max_tokens: int):
        """First-k-tokens text of every record of the last device batch (verbatim when the record has at
        most k tokens) -> (device text pointer, device offsets[R+1] pointer, byte count)."""
        t, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.trp_head_tokens_text_device(self._h, int(max_tokens), C.byref(t), C.byref(o), C.byref(n)))
        return t.value or 0, o.value or 0, int(n.value)

    def _texts_to_host(self, t, o, nbytes, n) -> List[bytes]:
        """Device text bytes + offsets[n+1] -> list of n byte strings."""
        offs = np.empty(n + 1, dtype=np.uint64)
        text = np.empty(nbytes, dtype=np.uint8)
        _check(self._L.trp_copy_to_host(self._h, offs.ctypes.data, o, offs.nbytes))
        if nbytes:
            _check(self._L.trp_copy_to_host(self._h, text.ctypes.data, t, text.nbytes))
        return [text[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(n)]

    def chunk_bodies(self, chunk_index: Sequence[int], budget: Sequence[int]) -> List[bytes]:
        """Text of the selected chunks of the last device batch cut to budget[i] tokens (verbatim when the
        budget covers the chunk): the body of a decorated chunk, src/sampler/mod.rs:1497-1504."""
        idx = np.ascontiguousarray(chunk_index, dtype=np.uint32)
        bud = np.ascontiguousarray(budget, dtype=np.uint32)
        t, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.trp_chunk_bodies_device(self._h, idx.ctypes.data, bud.ctypes.data, len(idx), C.byref(t),
                                               C.byref(o), C.byref(n)))
        return self._texts_to_host(t.value or 0, o.value or 0, int(n.value), len(idx))

    def _host_texts(self, ct, n) -> List[bytes]:
        """trp_chunk_text in host memory -> list of n byte strings (and release it)."""
        try:
            offs = np.frombuffer(C.string_at(ct.offsets, (n + 1) * 8), dtype=np.uint64)
            text = C.string_at(ct.text, int(ct.n_bytes)) if ct.n_bytes else b""
            return [text[int(offs[i]):int(offs[i + 1])] for i in range(n)]
        finally:
            self._L.trp_chunk_text_free(C.byref(ct))

    def bm25_texts_host(self, data: np.ndarray, offsets: np.ndarray, doc_offsets: Sequence[int], max_tokens: int,
                        newline: str = "\n") -> List[bytes]:
        """record_bm25_text (bm25_backend.rs:503-532) over a packed host batch whose records are the headings and
        section texts of the documents in order: document i is records [doc_offsets[i], doc_offsets[i+1])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        do = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        ct = _ChunkText()
        _check(self._L.trp_bm25_texts(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data,
                                      len(offsets) - 1, do.ctypes.data, len(do) - 1, int(max_tokens), newline.encode(),
                                      C.byref(ct)))
        return self._host_texts(ct, len(do) - 1)

    def head_tokens_text_host(self, data: np.ndarray, offsets: np.ndarray, max_tokens: int) -> List[bytes]:
        """Every record of a packed host batch cut to its first max_tokens tokens (verbatim when it has no more):
        the BM25 query cap, bm25_backend.rs:219-225, with max_tokens = 64."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ct = _ChunkText()
        _check(self._L.trp_head_tokens_text(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data,
                                            len(offsets) - 1, int(max_tokens), C.byref(ct)))
        return self._host_texts(ct, len(offsets) - 1)

    def chunk_bodies_host(self, data: np.ndarray, offsets: np.ndarray, denoiser: DenoiserConfig,
                          strategy: ChunkingStrategy, chunk_index: Sequence[int], budget: Sequence[int]) -> List[bytes]:
        """The packed host batch preprocessed and chunked as by preprocess_batch, then chunk chunk_index[i] cut to
        budget[i] tokens (the body of a decorated chunk, src/sampler/mod.rs:1497-1504), all in one call."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        idx = np.ascontiguousarray(chunk_index, dtype=np.uint32)
        bud = np.ascontiguousarray(budget, dtype=np.uint32)
        cfg = _make_cfg(denoiser, strategy)
        ct = _ChunkText()
        _check(self._L.trp_chunk_bodies(self._h, data.ctypes.data if data.size else None, offsets.ctypes.data,
                                        len(offsets) - 1, C.byref(cfg), idx.ctypes.data, bud.ctypes.data, len(idx),
                                        C.byref(ct)))
        return self._host_texts(ct, len(idx))

    def bm25_texts(self, doc_offsets: Sequence[int], max_tokens: int, newline: str = "\n") -> List[bytes]:
        """record_bm25_text (bm25_backend.rs:503-532) of the documents whose parts (headings and section texts,
        in order) are records [doc_offsets[i], doc_offsets[i+1]) of the last device batch.  A document
        without any token comes back empty: substitute the record id."""
        do = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        t, o, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._L.trp_bm25_texts_device(self._h, do.ctypes.data, len(do) - 1, int(max_tokens),
                                             newline.encode(), C.byref(t), C.byref(o), C.byref(n)))
        return self._texts_to_host(t.value or 0, o.value or 0, int(n.value), len(do) - 1)

    def set_profiling(self, on: bool):
        _check(self._L.trp_set_profiling(self._h, int(on)))

    def profile(self):
        arr = (_KernelTime * 64)()
        n = _check(self._L.trp_get_profile(self._h, arr, 64))
        return [(arr[i].name.decode(), float(arr[i].ms)) for i in range(min(n, 64))]


class ShardedPreprocessor:
    """One batch over several GPUs of this process: one context and one host thread per entry of `devices`
    (HIP device ordinals; an ordinal may be listed more than once, which gives that GPU several shards).  The
    batch is cut into byte-balanced record ranges (shard_ranges), every shard runs the whole path on its own
    context at the same time as the others (records are independent: nothing is exchanged between GPUs) and the
    results are concatenated in record order (concat_results: offset shifts only)."""

    def __init__(self, devices: Sequence[int]):
        if not len(devices):
            raise TripletsAmdError("ShardedPreprocessor needs at least one device")
        self.devices = [int(d) for d in devices]
        self._ctxs: List[Context] = []
        try:
            for d in self.devices:
                self._ctxs.append(Context(d))
        except Exception:
            self.close()
            raise

    def close(self):
        for c in self._ctxs:
            c.close()
        self._ctxs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _run(self, data, offsets, call) -> BatchResult:
        import threading
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        ranges = shard_ranges(offsets, len(self._ctxs))
        parts: List[Optional[BatchResult]] = [None] * len(ranges)
        errors: List[Optional[BaseException]] = [None] * len(ranges)

        def work(i):
            try:
                r0, r1 = ranges[i]
                sub, so = shard_slice(data, offsets, r0, r1)
                parts[i] = call(self._ctxs[i], sub, so)
            except BaseException as e:   # re-raised in the caller's thread
                errors[i] = e

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(ranges))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None:
                raise e
        return concat_results(parts)

    def preprocess_batch(self, data: np.ndarray, offsets: np.ndarray, denoiser: DenoiserConfig,
                         strategy: ChunkingStrategy) -> BatchResult:
        """Context.preprocess_batch over all devices.  tok_start of the result is u64 (the concatenated text of
        several shards may exceed the limit of one batch)."""
        return self._run(data, offsets, lambda c, d, o: c.preprocess_batch(d, o, denoiser, strategy))

    def normalize_batch(self, data: np.ndarray, offsets: np.ndarray) -> BatchResult:
        return self._run(data, offsets, lambda c, d, o: c.normalize_batch(d, o))


def preprocess_sharded(data: np.ndarray, offsets: np.ndarray, devices: Sequence[int],
                       denoiser: DenoiserConfig = None, strategy: ChunkingStrategy = None) -> BatchResult:
    """One packed host batch sharded by record range over `devices` (see ShardedPreprocessor, which keeps its
    contexts between batches; this convenience creates and destroys them per call)."""
    with ShardedPreprocessor(devices) as sp:
        return sp.preprocess_batch(data, offsets, denoiser or DenoiserConfig(), strategy or ChunkingStrategy())


def visible_devices() -> int:
    """Number of GPUs the HIP runtime shows to this process (0 without the runtime or a GPU)."""
    try:
        L = load_library()
    except (TripletsAmdError, OSError):
        return 0
    return max(0, int(L.trp_device_count()))


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def pack(records: Sequence[bytes]):
    """list of bytes -> (uint8 array, uint64 offsets[R+1])"""
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    if len(records):
        offs[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
    return np.frombuffer(b"".join(records), dtype=np.uint8).copy(), offs


def preprocess_batch(records: Sequence[bytes], denoiser: DenoiserConfig = None,
                     strategy: ChunkingStrategy = None, ctx: Context = None) -> BatchResult:
    data, offs = pack(records)
    return (ctx or default_context()).preprocess_batch(data, offs, denoiser or DenoiserConfig(),
                                                       strategy or ChunkingStrategy())


# ---------------------------------------------------------------------------------------------------
# single-record surface of the reference
# ---------------------------------------------------------------------------------------------------
def denoise_text(text: str, config: DenoiserConfig, ctx: Context = None) -> Optional[str]:
    """denoise_text(text, &config) -> Option<String>"""
    ctx = ctx or default_context()
    b = text.encode("utf-8")
    cfg = _make_cfg(config, ChunkingStrategy())
    out, n = C.c_void_p(), C.c_size_t()
    rc = _check(ctx._L.trp_denoise(ctx._h, b, len(b), C.byref(cfg), C.byref(out), C.byref(n)))
    if rc == 0:
        return None
    s = C.string_at(out, n.value) if n.value else b""
    ctx._L.trp_free(out)
    return s.decode("utf-8")


def normalize_inline_whitespace(text: str, ctx: Context = None) -> str:
    """src/utils.rs:18-33 for one text (a batch of one)."""
    ctx = ctx or default_context()
    raw = text.encode("utf-8")
    out = C.c_void_p()
    n = C.c_size_t()
    _check(ctx._L.trp_normalize_inline_whitespace(ctx._h, raw, len(raw), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value).decode("utf-8")
    finally:
        ctx._L.trp_free(out)


class DenoiserPreprocessor:
    """TextPreprocessor implementation (process(&self, &str) -> Option<String>)."""

    def __init__(self, config: DenoiserConfig):
        self.config = config

    def process(self, text: str, ctx: Context = None) -> Optional[str]:
        return denoise_text(text, self.config, ctx)


class WhitespaceTokenizer:
    def tokenize(self, text: str, ctx: Context = None) -> List[str]:
        ctx = ctx or default_context()
        b = text.encode("utf-8")
        s, ln, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _check(ctx._L.trp_tokenize(ctx._h, b, len(b), C.byref(s), C.byref(ln), C.byref(n)))
        starts = np.frombuffer(C.string_at(s, n.value * 4), dtype=np.uint32) if n.value else []
        lens = np.frombuffer(C.string_at(ln, n.value * 4), dtype=np.uint32) if n.value else []
        ctx._L.trp_free(s)
        ctx._L.trp_free(ln)
        return [b[int(a):int(a) + int(l)].decode("utf-8") for a, l in zip(starts, lens)]

    def token_count(self, text: str, ctx: Context = None) -> int:
        ctx = ctx or default_context()
        b = text.encode("utf-8")
        n = C.c_size_t()
        _check(ctx._L.trp_token_count(ctx._h, b, len(b), C.byref(n)))
        return int(n.value)


@dataclass
class ChunkView:
    kind: str            # "Window" | "SummaryFallback"
    index: int = 0
    overlap: int = 0
    span: int = 0
    strategy: str = ""
    weight: float = 0.0


@dataclass
class RecordChunk:
    record_id: str
    section_idx: int
    view: ChunkView
    text: str
    tokens_estimate: int


class SlidingWindowChunker:
    """ChunkingAlgorithm implementation: materialize(strategy, record, section_idx, section)."""

    def materialize(self, strategy: ChunkingStrategy, record_id: str, section_idx: int,
                    section_text: str, ctx: Context = None) -> List[RecordChunk]:
        den = DenoiserConfig()
        for p in strategy.preprocessors:
            if isinstance(p, DenoiserPreprocessor):
                den = p.config
            else:
                raise TripletsAmdError("only DenoiserPreprocessor is implemented on the GPU path")
        res = preprocess_batch([section_text.encode("utf-8")], den, strategy, ctx)
        out = []
        for i in range(res.n_chunks):
            c = res.chunks[i]
            if c["kind"] == 0:
                view = ChunkView("Window", int(c["index"]), int(c["overlap"]), int(c["span"]))
            else:
                view = ChunkView("SummaryFallback", strategy="head",
                                 weight=strategy.summary_fallback_weight)
            out.append(RecordChunk(record_id, section_idx, view, res.chunk_text(i).decode("utf-8"),
                                   int(c["tok_end"]) - int(c["tok_begin"])))
        return out


# ---------------------------------------------------------------------------------------------------
# the steps right after the path, on its token table
# ---------------------------------------------------------------------------------------------------
def decorate_chunks(ctx: Context, chunks: np.ndarray, chunk_index: Sequence[int], prefixes: Sequence[bytes],
                    max_window_tokens: int, newline: bytes = b"\n", prefix_ctx: Context = None):
    """decorate_chunk (src/sampler/mod.rs:1470-1508) for selected chunks of ctx's last device batch, each with
    the prefix the host's KvpPrefixSampler produced for it: returns [(text, tokens_estimate)].
    `chunks` is the batch's chunk descriptor array on the host.  The prefixes are tokenised on the GPU as a
    batch of their own (on prefix_ctx, so that ctx keeps its batch); the bodies are cut on the GPU."""
    pctx = prefix_ctx or ctx.helper()   # same GPU as ctx; kept with it, not created per call
    pdata, poffs = pack(list(prefixes))
    pres = pctx.preprocess_batch(pdata, poffs, DenoiserConfig(False), ChunkingStrategy())
    n = len(prefixes)
    np_tok = np.diff(pres.tok_offsets.astype(np.int64))
    sel = chunks[np.asarray(chunk_index, dtype=np.int64)]
    nb_tok = sel["tok_end"].astype(np.int64) - sel["tok_begin"].astype(np.int64)
    total = np_tok + nb_tok
    mw = int(max_window_tokens)
    cut = (total > mw) if mw > 0 else np.zeros(n, dtype=bool)          # mod.rs:1488
    prefix_only = cut & (np_tok >= mw)                                  # :1489
    budget = np.where(cut & ~prefix_only, mw - np_tok, 0xFFFFFFFF)      # :1497
    budget = np.where(prefix_only, 0, budget)
    bodies = ctx.chunk_bodies(chunk_index, budget.astype(np.uint32))
    out = []
    for i in range(n):
        if prefix_only[i]:
            out.append((b" ".join(pres.tokens(i)[:mw]), mw))            # :1490-1495
        elif cut[i]:
            out.append((bytes(prefixes[i]) + newline + bodies[i], mw))  # :1499-1501
        else:
            out.append((bytes(prefixes[i]) + newline + bodies[i], int(total[i])))   # :1504-1505
    return out


def record_bm25_texts(records, max_tokens: int, newline: str = "\n", ctx: Context = None) -> List[bytes]:
    """record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532) for many records at once, on ctx's GPU
    (the default context's when none is given).  records: [(record_id, [(heading or None, text), ...]), ...] with
    byte strings.  The batch runs on ctx's helper context, so ctx keeps its own last device batch."""
    ctx = ctx or default_context()
    parts, doc_offsets = [], [0]
    for _, sections in records:
        for heading, text in sections:
            if heading is not None:
                parts.append(heading)
            parts.append(text)
        doc_offsets.append(len(parts))
    data, offs = pack(parts)
    texts = ctx.helper().bm25_texts_host(data, offs, doc_offsets, max_tokens, newline)
    return [t if t else rid for t, (rid, _) in zip(texts, records)]       # :517-519
