#!/usr/bin/env python3
"""What does one very long digit-heavy line cost (minified numeric data with a few words in it)?  Config B's
text with one such line of the given length as a record of its own.  Checks the result against the CPU
restatement for the shorter lengths."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import corpusgen
import triplets_amd as T
import oracle_lib as O

rng = random.Random(9)
data, offs = corpusgen.config_b(20_000_000, 42, 16)[:2]
raw = bytes(data)
for n in (100_000, 1_000_000, 16_000_000):
    words = []
    size = 0
    while size < n:
        w = "item" if rng.random() < 0.01 else str(rng.randint(0, 10 ** rng.randint(1, 7)))
        words.append(w)
        size += len(w) + 1
    line = " ".join(words).encode()
    body = raw + line
    noffs = np.concatenate([np.asarray(offs, dtype=np.int64), np.asarray([len(body)], dtype=np.int64)])
    arr = np.frombuffer(body, dtype=np.uint8)
    db = torch.from_numpy(np.concatenate([arr, np.zeros(64, np.uint8)])).cuda()
    do = torch.from_numpy(noffs).cuda()
    torch.cuda.synchronize()
    ctx = T.Context(0)
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
    for _ in range(2):
        res = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(arr), den, st)
    ctx.set_profiling(True)
    res = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(noffs) - 1, len(arr), den, st)
    prof = dict(ctx.profile())
    msg = ""
    if n <= 1_000_000:
        exp = O.batch(arr, noffs.astype(np.uint64), O.make_config(True, 0.35, True, 1024, (64,), 512))
        got = res.to_host()
        same = exp.n_denoised == got.n_denoised and exp.n_tokens == got.n_tokens and bytes(exp.denoised) == bytes(got.denoised[:exp.n_denoised])
        msg = "matches the CPU restatement" if same else "DIFFERS from the CPU restatement"
    print("digit-heavy line of %9d bytes: k_gate3 %.3f ms, whole batch %.3f ms  %s" % (len(line), prof["k_gate3"], sum(prof.values()), msg), flush=True)
    ctx.close()
