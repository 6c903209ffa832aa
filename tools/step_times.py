#!/usr/bin/env python3
"""Duration of every single blocking call from the first one on (config B, 1e9 bytes): how long a freshly
idle GPU takes to reach its steady rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd")); sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
import numpy as np, torch, corpusgen, triplets_amd as T
data, offs = corpusgen.config_b(1_000_000_000, 42, 16)
db, do = torch.from_numpy(data).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda()
torch.cuda.synchronize()
ctx = T.Context(0)
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
R, N = len(offs) - 1, len(data)
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for rnd in range(2):
    ts = []
    for i in range(40):
        t0 = time.perf_counter()
        ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), R, N, den, st)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("round %d, ms per call:" % rnd, " ".join("%.3f" % t for t in ts), flush=True)
    time.sleep(idle)
