#!/usr/bin/env python3
"""How much do wide numeric table rows cost?  Config B's text with one digit-heavy row of 300-600 bytes appended
to every record (one per 4 KB): such rows are longer than the 256 bytes one lane solves and go to the
solvers of k_gate3_solve in which lanes work together.  Other lengths: tools/wide_rows_bench.py LO HI.
Prints GB/s and the per-kernel times."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
import numpy as np
import torch
import corpusgen
import triplets_amd as T

rng = random.Random(3)
LO, HI = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 600)   # row length in bytes
data, offs = corpusgen.config_b(200_000_000, 42, 16)[:2]
raw = bytes(data)
parts = []
noffs = [0]
size = 0
for r in range(len(offs) - 1):
    rec = raw[offs[r]:offs[r + 1]]
    n = rng.randint(LO, HI)
    row = "| Item %d | " % r + " | ".join("%d,%03d.%02d" % (rng.randint(1, 999), rng.randint(0, 999), rng.randint(0, 99))
                                          for _ in range(n // 12)) + " | Total |"
    rec = rec + b"\n" + row.encode()
    parts.append(rec)
    size += len(rec)
    noffs.append(size)
blob = np.frombuffer(b"".join(parts), dtype=np.uint8)
db = torch.from_numpy(np.concatenate([blob, np.zeros(64, np.uint8)])).cuda()
do = torch.from_numpy(np.asarray(noffs, dtype=np.int64)).cuda()
torch.cuda.synchronize()
ctx = T.Context(0)
den = T.DenoiserConfig(True, 0.35, True)
st = T.ChunkingStrategy()
import time
for _ in range(3):
    res = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(blob), den, st)
t0 = time.perf_counter()
for _ in range(10):
    res = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(blob), den, st)
dt = (time.perf_counter() - t0) / 10
print("%d bytes, %d records with one wide row each: %.1f GB/s, %.3f ms" % (len(blob), len(noffs) - 1, len(blob) / dt / 1e9, dt * 1e3))
ctx.set_profiling(True)
ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(blob), den, st)
print([(k, round(v, 4)) for k, v in ctx.profile()])
