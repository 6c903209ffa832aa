"""Host-side helpers of the Python binding that need no GPU: result arrays handed over without a copy (and given
back exactly once), copies of short arrays, the per-setting configuration cache."""
import ctypes as C
import gc
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-triplets_amd"))
import triplets_amd as T   # noqa: E402

libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]


class _FakeLib:
    def __init__(self):
        self.freed = []

    def trp_free(self, p):
        self.freed.append(p)
        libc.free(p)


def test_long_arrays_are_adopted_and_freed_once_with_the_last_view():
    lib = _FakeLib()
    n = 1 << 18
    p = libc.malloc(n * 4)
    src = np.arange(n, dtype=np.uint32)
    C.memmove(p, src.ctypes.data, n * 4)
    a, adopted = T._take_out(lib, p, n * 4, np.uint32)
    assert adopted and a.ctypes.data == p and np.array_equal(a, src)
    view = a[1000:2000]
    del a
    gc.collect()
    assert lib.freed == []          # a view still needs the memory
    assert int(view[0]) == 1000
    del view
    gc.collect()
    assert lib.freed == [p]


def test_short_arrays_are_copied_and_left_to_the_caller():
    lib = _FakeLib()
    p = libc.malloc(64)
    C.memset(p, 7, 64)
    a, adopted = T._take_out(lib, p, 64, np.uint8)
    assert not adopted and a.ctypes.data != p and a.tolist() == [7] * 64
    libc.free(p)
    assert lib.freed == []


def test_configurations_are_kept_per_setting():
    st = T.ChunkingStrategy(max_window_tokens=7, overlap_tokens=[3, 1], summary_fallback_tokens=2)
    a = T._make_cfg(T.DenoiserConfig(True, 0.35, True), st)
    b = T._make_cfg(T.DenoiserConfig(True, 0.35, True),
                    T.ChunkingStrategy(max_window_tokens=7, overlap_tokens=[3, 1], summary_fallback_tokens=2))
    assert a is b
    c = T._make_cfg(T.DenoiserConfig(True, 0.36, True), st)
    assert c is not a and abs(c.max_digit_ratio - 0.36) < 1e-6 and c.n_overlaps == 2
    assert [c.overlaps[i] for i in range(2)] == [3, 1] and c.max_window_tokens == 7
    n1 = T._make_cfg(T.DenoiserConfig(True, float("nan"), True), st)
    n2 = T._make_cfg(T.DenoiserConfig(True, float("nan"), True), st)
    assert n1 is not n2 and n1.max_digit_ratio != n1.max_digit_ratio
