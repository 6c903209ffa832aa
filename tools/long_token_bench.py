#!/usr/bin/env python3
"""What does one very long token cost (a base64 blob, a data URI: no whitespace for megabytes)?  Config B's text
with one token of the given length inside one record.  Prints the per-kernel times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
import numpy as np
import torch
import corpusgen
import triplets_amd as T

data, offs = corpusgen.config_b(100_000_000, 42, 16)[:2]
raw = bytes(data)
for blob_len in (0, 1 << 20, 16 << 20):
    cut = int(offs[len(offs) // 2])
    blob = (b"QWxhZGRpbjpvcGVuIHNlc2FtZQ" * (blob_len // 26 + 1))[:blob_len]
    body = raw[:cut] + b"image " + blob + b" end\n" + raw[cut:]
    shift = len(body) - len(raw)
    noffs = np.asarray([o if o <= cut and i <= len(offs) // 2 else o + shift for i, o in enumerate(offs)], dtype=np.int64)
    noffs[len(offs) // 2] = cut   # the blob belongs to the record that starts at `cut`
    arr = np.frombuffer(body, dtype=np.uint8)
    db = torch.from_numpy(np.concatenate([arr, np.zeros(64, np.uint8)])).cuda()
    do = torch.from_numpy(noffs).cuda()
    torch.cuda.synchronize()
    for den in (T.DenoiserConfig(True, 0.35, True), T.DenoiserConfig(False, 0.35, True)):
        ctx = T.Context(0)
        st = T.ChunkingStrategy()
        for _ in range(2):
            ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(arr), den, st)
        ctx.set_profiling(True)
        r = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(arr), den, st)
        prof = ctx.profile()
        print("token of %8d bytes, denoiser %-5s: total %.3f ms  %s" % (
            blob_len, den.enabled, sum(v for _, v in prof), [(k, round(v, 3)) for k, v in prof if v > 0.05]), flush=True)
        ctx.close()
