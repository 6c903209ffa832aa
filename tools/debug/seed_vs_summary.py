#!/usr/bin/env python3
"""Debug aid: the same batch through the k_seed path and through the forced per-tile-summary path;
prints the first record whose denoised text differs, with the position of the difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
import numpy as np
import corpusgen
import triplets_amd as T

cfgname = sys.argv[1] if len(sys.argv) > 1 else "B"
size = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
gen = {"B": corpusgen.config_b, "C": corpusgen.config_c, "D": corpusgen.config_d}[cfgname]
data, offs = gen(size, {"B": 42, "C": 43, "D": 44}[cfgname], 16)
den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
os.environ.pop("TRIPLETS_AMD_FORCE_SUMMARY", None)
a = T.Context(0)
os.environ["TRIPLETS_AMD_FORCE_SUMMARY"] = "1"
b = T.Context(0)
import torch
d_bytes = torch.from_numpy(data).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
ra = a.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), len(offs) - 1, len(data), den, st).to_host()
rb = b.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), len(offs) - 1, len(data), den, st).to_host()
print("totals", ra.n_denoised, rb.n_denoised, ra.n_tokens, rb.n_tokens)
da, db = ra.denoised_offsets, rb.denoised_offsets
bad = np.nonzero(da != db)[0]
print("offset mismatches:", len(bad))
shown = 0
R = len(offs) - 1
la, lb = np.diff(da.astype(np.int64)), np.diff(db.astype(np.int64))
for r in np.nonzero(la != lb)[0][:5]:
    ta = ra.denoised[int(da[r]):int(da[r + 1])].tobytes()
    tb = rb.denoised[int(db[r]):int(db[r + 1])].tobytes()
    i = next((k for k in range(min(len(ta), len(tb))) if ta[k] != tb[k]), min(len(ta), len(tb)))
    src = data[int(offs[r]):int(offs[r + 1])].tobytes()
    print("record", r, "input range", int(offs[r]), int(offs[r + 1]), "tile of start", int(offs[r]) // 16384,
          "len seed/summary", len(ta), len(tb), "first diff at", i)
    print("  seed   :", ta[max(0, i - 120):i + 120])
    print("  summary:", tb[max(0, i - 120):i + 120])
    # where in the input is that text?
    key = tb[max(0, i - 40):i]
    p = src.find(key)
    if p >= 0:
        pos = int(offs[r]) + p + len(key)
        print("  input position ~", pos, "= tile", pos // 16384, "+", pos % 16384, "block in tile", (pos % 16384) // 64)
        print("  input  :", src[max(0, p - 80):p + 200])

# first half of the records as its own batch, through both contexts
r1 = T.shard_ranges(offs, 2)[0][1]
b1 = int(offs[r1])
sub = d_bytes[:b1].clone()
sub_offs = torch.from_numpy(offs[:r1 + 1].astype(np.int64)).cuda()
for name, ctx in (("seed", a), ("summary", b)):
    h = ctx.preprocess_batch_device(sub.data_ptr(), sub_offs.data_ptr(), r1, b1, den, st).to_host()
    lh = np.diff(h.denoised_offsets.astype(np.int64))
    diff = np.nonzero(lh != la[:r1])[0]
    print("half via", name, ": records whose length differs from the whole run:", len(diff), diff[:5])
    for r in diff[:3]:
        ta = h.denoised[int(h.denoised_offsets[r]):int(h.denoised_offsets[r + 1])].tobytes()
        tb = ra.denoised[int(da[r]):int(da[r + 1])].tobytes()
        i = next((k for k in range(min(len(ta), len(tb))) if ta[k] != tb[k]), min(len(ta), len(tb)))
        src = data[int(offs[r]):int(offs[r + 1])].tobytes()
        print("record", r, "input range", int(offs[r]), int(offs[r + 1]), "len half/whole", len(ta), len(tb), "first diff at", i)
        print("  half :", ta[max(0, i - 100):i + 100])
        print("  whole:", tb[max(0, i - 100):i + 100])
        key = tb[max(0, i - 40):i]
        p = src.find(key)
        if p >= 0:
            pos = int(offs[r]) + p + len(key)
            print("  input position ~", pos, "= tile", pos // 16384, "+", pos % 16384, "block in tile", (pos % 16384) // 64)
            print("  input  :", src[max(0, p - 80):p + 200])
