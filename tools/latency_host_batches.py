import os, sys, time
import numpy as np
sys.path.insert(0, "rust-triplets_amd"); sys.path.insert(0, "tools/corpusgen")
import corpusgen, triplets_amd as T
data, offs = corpusgen.config_b(20_000_000, 42, 4)
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
ctx = T.Context(0)
for want in (4000, 30000, 64000, 256000, 1_000_000, 4_000_000, 16_000_000):
    r = max(1, int(np.searchsorted(offs, want, side="right")) - 1); n = int(offs[r])
    d = data[:n].copy(); o = offs[:r+1].copy()
    for mask, name in ((T.OUT_ALL, "all"), (T.OUT_CHUNKS, "chunks")):
        for _ in range(10): ctx.preprocess_batch_select(d, o, den, st, mask)
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): ctx.preprocess_batch_select(d, o, den, st, mask)
            runs.append((time.perf_counter() - t0) / 20)
        dt = sorted(runs)[2]
        print("host batch %9d bytes %5d records, outputs %-6s: %8.1f us, %6.2f GB/s" % (n, r, name, dt * 1e6, n / dt / 1e9), flush=True)
