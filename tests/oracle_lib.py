"""ctypes binding of oracle/liboracle.so - the CPU restatement of the reference path.
Test infrastructure only: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")


class OrcConfig(C.Structure):
    _fields_ = [("denoise_enabled", C.c_uint8), ("strip_markdown", C.c_uint8),
                ("max_digit_ratio", C.c_float), ("max_window_tokens", C.c_uint32),
                ("overlaps", C.POINTER(C.c_uint32)), ("n_overlaps", C.c_uint32),
                ("summary_fallback_tokens", C.c_uint32)]


class OrcChunk(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("record", "kind", "index", "overlap", "span", "tok_begin", "tok_end", "reserved")]


class OrcResult(C.Structure):
    _fields_ = [("denoised", C.POINTER(C.c_uint8)), ("denoised_offsets", C.POINTER(C.c_uint64)),
                ("dropped", C.POINTER(C.c_uint8)), ("tok_start", C.POINTER(C.c_uint32)),
                ("tok_len", C.POINTER(C.c_uint32)), ("tok_offsets", C.POINTER(C.c_uint64)),
                ("chunks", C.POINTER(OrcChunk)), ("chunk_offsets", C.POINTER(C.c_uint64)),
                ("n_records", C.c_uint64), ("n_denoised", C.c_uint64), ("n_tokens", C.c_uint64),
                ("n_chunks", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        L = C.CDLL(_SO)
        L.orc_denoise.restype = C.c_int
        L.orc_denoise.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_float, C.c_int,
                                  C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        L.orc_tokenize.restype = C.c_size_t
        L.orc_tokenize.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32), C.c_size_t]
        L.orc_chunk_views.restype = C.c_size_t
        L.orc_chunk_views.argtypes = [C.c_uint64, C.POINTER(OrcConfig), C.c_uint32,
                                      C.POINTER(OrcChunk), C.c_size_t]
        L.orc_batch_mt.restype = C.c_int
        L.orc_batch_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(OrcConfig),
                                   C.c_int, C.POINTER(OrcResult)]
        L.orc_result_free.argtypes = [C.POINTER(OrcResult)]
        L.orc_chunk_text.restype = C.c_size_t
        L.orc_chunk_text.argtypes = [C.POINTER(OrcResult), C.c_uint64, C.c_char_p, C.c_size_t]
        L.orc_is_alphabetic.argtypes = [C.c_uint32]
        L.orc_is_whitespace.argtypes = [C.c_uint32]
        for f in (L.orc_is_markdown_table_line, L.orc_is_markdown_table_separator):
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_strip_table_pipes.restype = C.c_size_t
        L.orc_strip_table_pipes.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_normalize_inline_whitespace.restype = C.c_size_t
        L.orc_normalize_inline_whitespace.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_decorate_chunk.restype = C.c_size_t
        L.orc_decorate_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p,
                                         C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_record_bm25_text.restype = C.c_size_t
        L.orc_record_bm25_text.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_char_p,
                                           C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_char_p,
                                           C.c_size_t, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.orc_cap_tokens.restype = C.c_size_t
        L.orc_cap_tokens.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def make_config(enabled=True, max_digit_ratio=0.35, strip_markdown=True, max_window_tokens=1024,
                overlaps=(64,), summary_fallback_tokens=512):
    ov = (C.c_uint32 * max(1, len(overlaps)))(*overlaps)
    cfg = OrcConfig(int(enabled), int(strip_markdown), max_digit_ratio, max_window_tokens,
                    C.cast(ov, C.POINTER(C.c_uint32)), len(overlaps), summary_fallback_tokens)
    cfg._keep = ov
    return cfg


def denoise(text: bytes, enabled=True, max_digit_ratio=0.35, strip_markdown=True):
    L = lib()
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t(0)
    some = L.orc_denoise(text, len(text), int(enabled), max_digit_ratio, int(strip_markdown),
                         C.byref(out), C.byref(n))
    if not some:
        return None
    res = C.string_at(out, n.value) if n.value else b""
    if out:
        L.free(out)
    return res


def tokenize(text: bytes):
    L = lib()
    cap = len(text) // 2 + 2
    s = (C.c_uint32 * cap)()
    ln = (C.c_uint32 * cap)()
    n = L.orc_tokenize(text, len(text), s, ln, cap)
    return [(s[i], ln[i]) for i in range(n)]


def chunk_views(n_tokens, cfg):
    L = lib()
    n = L.orc_chunk_views(n_tokens, C.byref(cfg), 0, None, 0)
    arr = (OrcChunk * max(1, n))()
    L.orc_chunk_views(n_tokens, C.byref(cfg), 0, arr, n)
    return [(c.kind, c.index, c.overlap, c.span, c.tok_begin, c.tok_end) for c in arr[:n]]


class BatchResult:
    """numpy views of one packed-batch result (arrays are copies; safe after free)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def record_text(self, r):
        return self.denoised[self.denoised_offsets[r]:self.denoised_offsets[r + 1]].tobytes()

    def chunk_text(self, c):
        rec = int(self.chunks[c]["record"])
        t0 = int(self.tok_offsets[rec])
        nt = int(self.tok_offsets[rec + 1]) - t0
        ch = self.chunks[c]
        if (self.chunk_offsets[rec + 1] - self.chunk_offsets[rec] == 1 and ch["kind"] == 0
                and ch["tok_end"] - ch["tok_begin"] == nt):
            return self.record_text(rec)
        parts = []
        for k in range(int(ch["tok_begin"]), int(ch["tok_end"])):
            s = int(self.tok_start[t0 + k])
            parts.append(self.denoised[s:s + int(self.tok_len[t0 + k])].tobytes())
        return b" ".join(parts)


CHUNK_DTYPE = np.dtype([(n, "<u4") for n in
                        ("record", "kind", "index", "overlap", "span", "tok_begin", "tok_end",
                         "reserved")])


def pack(records):
    """list of bytes -> (uint8 array, uint64 offsets[R+1])"""
    offs = np.zeros(len(records) + 1, dtype=np.uint64)
    if records:
        offs[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
    data = np.frombuffer(b"".join(records), dtype=np.uint8).copy()
    return data, offs


def batch(data: np.ndarray, offsets: np.ndarray, cfg, n_threads=1) -> BatchResult:
    L = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    R = len(offsets) - 1
    res = OrcResult()
    dptr = data.ctypes.data if data.size else None
    rc = L.orc_batch_mt(dptr, offsets.ctypes.data, R, C.byref(cfg), n_threads, C.byref(res))
    assert rc == 0

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dt).copy() if dt != CHUNK_DTYPE else \
            np.frombuffer(C.string_at(ptr, n * C.sizeof(OrcChunk)), dtype=CHUNK_DTYPE).copy()

    out = BatchResult(
        denoised=arr(res.denoised, res.n_denoised, np.uint8),
        denoised_offsets=arr(res.denoised_offsets, R + 1, np.uint64),
        dropped=arr(res.dropped, R, np.uint8),
        tok_start=arr(res.tok_start, res.n_tokens, np.uint32),
        tok_len=arr(res.tok_len, res.n_tokens, np.uint32),
        tok_offsets=arr(res.tok_offsets, R + 1, np.uint64),
        chunks=arr(res.chunks, res.n_chunks, CHUNK_DTYPE),
        chunk_offsets=arr(res.chunk_offsets, R + 1, np.uint64),
        n_records=R, n_denoised=int(res.n_denoised), n_tokens=int(res.n_tokens),
        n_chunks=int(res.n_chunks))
    L.orc_result_free(C.byref(res))
    return out


def _take(ptr, n):
    out = C.string_at(ptr, n) if n else b""
    if ptr:
        lib().free(ptr)
    return out


def decorate_chunk(text: bytes, prefix: bytes, max_window_tokens: int, newline: bytes = b"\n"):
    """decorate_chunk's text and tokens_estimate for a sampled prefix (src/sampler/mod.rs:1484-1506)."""
    out, est = C.c_void_p(), C.c_uint64()
    n = lib().orc_decorate_chunk(text, len(text), prefix, len(prefix), max_window_tokens, newline, len(newline),
                                 C.byref(out), C.byref(est))
    return _take(out, n), int(est.value)


def record_bm25_text(record_id: bytes, sections, max_tokens: int, newline: bytes = b"\n") -> bytes:
    """record_bm25_text (bm25_backend.rs:503-532); sections = [(heading or None, text), ...]."""
    n = len(sections)
    hs = (C.c_char_p * max(1, n))(*[h if h is not None else b"" for h, _ in sections])
    hl = (C.c_size_t * max(1, n))(*[len(h) if h is not None else 0 for h, _ in sections])
    has = bytes(1 if h is not None else 0 for h, _ in sections) + b"\0"
    ts = (C.c_char_p * max(1, n))(*[t for _, t in sections])
    tl = (C.c_size_t * max(1, n))(*[len(t) for _, t in sections])
    out = C.c_void_p()
    ln = lib().orc_record_bm25_text(hs, hl, has, ts, tl, n, record_id, len(record_id), max_tokens, newline,
                                    len(newline), C.byref(out))
    return _take(out, ln)


def cap_tokens(text: bytes, limit: int) -> bytes:
    out = C.c_void_p()
    n = lib().orc_cap_tokens(text, len(text), limit, C.byref(out))
    return _take(out, n)
