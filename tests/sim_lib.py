"""ctypes binding of the host simulator of the GPU algorithm (rust-triplets_amd/csrc/libtrp_sim.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(ROOT, "rust-triplets_amd", "csrc")
_SO = os.path.join(_DIR, "libtrp_sim.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_DIR, f) for f in ("trp_sim.cpp", "trp_lane.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["make", "-C", _DIR, "libtrp_sim.so"])
        L = C.CDLL(_SO)
        L.trp_sim_batch.restype = C.c_int
        L.trp_sim_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(O.OrcConfig),
                                    C.POINTER(O.OrcResult), C.c_int]
        L.trp_sim_free.argtypes = [C.POINTER(O.OrcResult)]
        L.trp_sim_check_monoids.restype = C.c_int
        L.trp_sim_check_monoids.argtypes = [C.c_uint, C.c_int]
        _lib = L
    return _lib


def batch(data, offsets, cfg, tile_blocks=256):
    L = lib()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    R = len(offsets) - 1
    res = O.OrcResult()
    rc = L.trp_sim_batch(data.ctypes.data if data.size else None, offsets.ctypes.data, R,
                         C.byref(cfg), C.byref(res), tile_blocks)
    assert rc == 0

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        if dt == O.CHUNK_DTYPE:
            return np.frombuffer(C.string_at(ptr, n * C.sizeof(O.OrcChunk)), dtype=dt).copy()
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    out = O.BatchResult(
        denoised=arr(res.denoised, res.n_denoised, np.uint8),
        denoised_offsets=arr(res.denoised_offsets, R + 1, np.uint64),
        dropped=arr(res.dropped, R, np.uint8),
        tok_start=arr(res.tok_start, res.n_tokens, np.uint32),
        tok_len=arr(res.tok_len, res.n_tokens, np.uint32),
        tok_offsets=arr(res.tok_offsets, R + 1, np.uint64),
        chunks=arr(res.chunks, res.n_chunks, O.CHUNK_DTYPE),
        chunk_offsets=arr(res.chunk_offsets, R + 1, np.uint64),
        n_records=R, n_denoised=int(res.n_denoised), n_tokens=int(res.n_tokens),
        n_chunks=int(res.n_chunks))
    L.trp_sim_free(C.byref(res))
    return out
