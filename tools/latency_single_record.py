#!/usr/bin/env python3
"""Latency of the single-record conveniences (trp_denoise, trp_token_count: a batch of one in host memory) and of
small device-resident batches.  python tools/latency_single_record.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402
import textgen       # noqa: E402
import triplets_amd as T   # noqa: E402

ctx = T.Context(0)
L = T.load_library()
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
cfg = T._make_cfg(den, st)
for n in (100, 1000, 4000, 16000, 64000):
    text = b"\n".join(textgen.corpus(5, 400))[:n]
    out, out_len = C.c_void_p(), C.c_size_t()
    for _ in range(20):
        L.trp_denoise(ctx._h, text, len(text), C.byref(cfg), C.byref(out), C.byref(out_len))
        L.trp_free(out)
    t0 = time.perf_counter()
    for _ in range(200):
        L.trp_denoise(ctx._h, text, len(text), C.byref(cfg), C.byref(out), C.byref(out_len))
        L.trp_free(out)
    d1 = (time.perf_counter() - t0) / 200
    ntok = C.c_size_t()
    t0 = time.perf_counter()
    for _ in range(200):
        L.trp_token_count(ctx._h, text, len(text), C.byref(ntok))
    d2 = (time.perf_counter() - t0) / 200
    print("text of %6d bytes: trp_denoise %.1f us, trp_token_count %.1f us per call" % (len(text), d1 * 1e6, d2 * 1e6), flush=True)
