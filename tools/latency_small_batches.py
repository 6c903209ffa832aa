#!/usr/bin/env python3
"""Blocking device calls on small batches of config-B text (ASCII English, records of 3-5 KB): microseconds per
call and GB/s by batch size, with the single-launch path (k_small_batch, up to 32 KiB) and with the kernel sequence
forced (TRIPLETS_AMD_NO_SMALL=1).  python3 tools/latency_small_batches.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
import corpusgen          # noqa: E402
import triplets_amd as T  # noqa: E402

data, offs = corpusgen.config_b(8_000_000, 42, 4)
SIZES = [int(x) for x in os.environ.get("SIZES", "4000,16000,60000,256000,1000000,4000000").split(",")]
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()


def run(ctx, label):
    for want in SIZES:
        r = max(1, int(np.searchsorted(offs, want, side="right")) - 1)
        n = int(offs[r])
        db = torch.from_numpy(np.concatenate([data[:n], np.zeros(64, np.uint8)])).cuda()
        do = torch.from_numpy(offs[:r + 1].astype(np.int64)).cuda()
        torch.cuda.synchronize()
        for _ in range(30):
            ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), r, n, den, st)
        reps, runs = 100, []
        for _ in range(7):          # median of 7 runs of 100 calls: a stall of the host lands in one run only
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), r, n, den, st)
            runs.append((time.perf_counter() - t0) / reps)
        dt = sorted(runs)[len(runs) // 2]
        print("%-16s records %5d bytes %8d: %7.1f us per call, %7.2f GB/s" % (label, r, n, dt * 1e6, n / dt / 1e9), flush=True)


run(T.Context(0), "single launch")
os.environ["TRIPLETS_AMD_NO_SMALL"] = "1"
run(T.Context(0), "kernel sequence")
