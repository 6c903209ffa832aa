import sys, time
import numpy as np
sys.path.insert(0, "rust-triplets_amd"); sys.path.insert(0, "tools/corpusgen")
import corpusgen, triplets_amd as T
data, offs = corpusgen.config_b(2_000_000, 42, 4)
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
ctx = T.Context(0)
for r in (1, 4):
    n = int(offs[r]); d = data[:n].copy(); o = offs[:r+1].copy()
    for mask, name in ((T.OUT_CHUNKS, "descriptors"), (T.OUT_CHUNKS | T.OUT_CHUNK_TEXT, "descriptors + chunk text"), (T.OUT_ALL, "all arrays")):
        for _ in range(50): ctx.preprocess_batch_select(d, o, den, st, mask)
        runs = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(100): ctx.preprocess_batch_select(d, o, den, st, mask)
            runs.append((time.perf_counter() - t0) / 100)
        print("records %d bytes %6d %-26s %6.1f us" % (r, n, name, sorted(runs)[3] * 1e6), flush=True)
