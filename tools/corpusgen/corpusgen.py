"""ctypes wrapper of libcorpusgen.so: the synthetic corpora of the benchmark configurations."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libcorpusgen.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _DIR])
        L = C.CDLL(_SO)
        L.cg_plan.restype = C.c_uint64
        L.cg_plan.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        L.cg_generate.restype = C.c_uint64
        L.cg_generate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


KINDS = {"B": 0, "C": 1, "D": 2, "W": 3}


def generate(kind, seed, total_bytes, lo, hi, mode=0, threads=8, out=None):
    """-> (uint8 array, uint64 offsets[R+1]).  kind 'B' | 'C' | 'D'; mode 0 uniform lengths in [lo, hi],
    mode 1 skewed (lo or hi, half of the bytes each)."""
    L = lib()
    max_records = total_bytes // max(1, min(lo, hi)) + 16
    targets = np.zeros(max_records, dtype=np.uint64)
    R = L.cg_plan(seed, total_bytes, lo, hi, mode, targets.ctypes.data, max_records)
    offsets = np.zeros(R + 1, dtype=np.uint64)
    total = L.cg_generate(KINDS[kind], seed, R, targets.ctypes.data, offsets.ctypes.data, None, threads)
    data = out if out is not None else np.empty(total, dtype=np.uint8)
    assert data.size >= total
    L.cg_generate(KINDS[kind], seed, R, targets.ctypes.data, offsets.ctypes.data, data.ctypes.data, threads)
    return data[:total], offsets


def config_b(total_bytes=1_000_000_000, seed=42, threads=8):
    return generate("B", seed, total_bytes, 3072, 5120, 0, threads)


def config_c(total_bytes=4_000_000_000, seed=43, threads=8):
    return generate("C", seed, total_bytes, 2048, 16384, 0, threads)


def config_d(total_bytes=1_000_000_000, seed=44, threads=8):
    return generate("D", seed, total_bytes, 200, 1 << 20, 1, threads)


def config_w(total_bytes=1_000_000_000, seed=46, threads=8):
    """Config C with half of its multi-byte words in scripts outside the classifier's bit-plane rules."""
    return generate("W", seed, total_bytes, 2048, 16384, 0, threads)
