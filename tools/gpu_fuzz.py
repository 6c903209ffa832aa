#!/usr/bin/env python3
"""Soak test: random corpora and random configurations through the GPU library against the oracle.

  python tools/gpu_fuzz.py [--seconds 120] [--seed0 1000] [--small]

Prints one line per ~20 batches to keep a remote run visibly alive; exits non-zero at the first mismatch
(the failing seed and configuration are printed)."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O   # noqa: E402
import textgen           # noqa: E402
import triplets_amd as T  # noqa: E402
from parity import assert_same  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--small", action="store_true",
                    help="cut every batch to at most 64 KiB: the single-launch path and its gathered results")
    args = ap.parse_args()
    ctx = T.Context(0)
    t_end = time.time() + args.seconds
    seed = args.seed0
    n = 0
    while time.time() < t_end:
        rng = random.Random(seed)
        nrec = rng.choice([1, 3, 40, 300, 1200])
        recs = textgen.corpus(seed, nrec)
        if rng.random() < 0.3:   # many scripts, edge characters of the classifier's range rules
            recs += textgen.script_corpus(seed, max(1, nrec // 4))
            rng.shuffle(recs)
        if rng.random() < 0.05:  # a long letterless stretch and a long line: tile carries from the summaries
            recs.append(("12 34\n" * rng.randint(2000, 9000) + "word\n").encode())
            recs.append((" ".join("w%d" % i for i in range(rng.randint(3000, 9000)))).encode())
        if rng.random() < 0.3:   # glue records together: long records, many lines
            k = rng.randint(2, 50)
            recs = [b"\n".join(recs[i:i + k]) for i in range(0, len(recs), k)]
        if rng.random() < 0.2:
            recs.insert(rng.randrange(len(recs) + 1), b"")
        enabled = rng.random() < 0.8
        ratio = rng.choice([0.0, 0.1, 0.35, 0.35, 0.5, 0.9, 1.0, 1.0 / 3.0, 0.34999999, 1e-9, -0.5, 1.5, float("nan"),
                            float("inf"), 0.25, 0.75])
        md = rng.random() < 0.8
        mwt = rng.choice([1, 2, 7, 64, 1024])
        ov = tuple(rng.choice([0, 1, 3, 64, 2000]) for _ in range(rng.choice([0, 1, 1, 2, 3])))
        sft = rng.choice([0, 1, 16, 512])
        if args.small:
            limit = rng.choice([2000, 16000, 32768, 65536])
            while len(recs) > 1 and sum(len(r) for r in recs) > limit:
                recs.pop()
            if sum(len(r) for r in recs) > limit:
                recs = [recs[0][:limit].decode("utf-8", "ignore").encode()]
        data, offs = O.pack(recs)
        exp = O.batch(data, offs, O.make_config(enabled, ratio, md, mwt, ov, sft), n_threads=8)
        got = ctx.preprocess_batch(data, offs, T.DenoiserConfig(enabled, ratio, md),
                                   T.ChunkingStrategy(max_window_tokens=mwt, overlap_tokens=list(ov),
                                                      summary_fallback_tokens=sft))
        try:
            assert_same(exp, got, recs, "seed %d" % seed)
        except AssertionError as e:
            print("MISMATCH seed=%d nrec=%d enabled=%s ratio=%s md=%s mwt=%d ov=%s sft=%d\n%s"
                  % (seed, nrec, enabled, ratio, md, mwt, ov, sft, e), flush=True)
            sys.exit(1)
        if args.small:
            # the texts of the chunks as well (descriptors + chunk text rendered on the GPU, what a table of chunks asks for)
            sel = ctx.preprocess_batch_select(data, offs, T.DenoiserConfig(enabled, ratio, md),
                                              T.ChunkingStrategy(max_window_tokens=mwt, overlap_tokens=list(ov),
                                                                 summary_fallback_tokens=sft),
                                              T.OUT_CHUNKS | T.OUT_CHUNK_TEXT)
            o = sel.chunk_text_offsets
            ok = len(o) == exp.n_chunks + 1
            for c in range(exp.n_chunks if ok else 0):
                if sel.chunk_text_bytes[int(o[c]):int(o[c + 1])].tobytes() != exp.chunk_text(c):
                    ok = False
                    break
            if not ok:
                print("CHUNK TEXT MISMATCH seed=%d nrec=%d enabled=%s ratio=%s md=%s mwt=%d ov=%s sft=%d"
                      % (seed, nrec, enabled, ratio, md, mwt, ov, sft), flush=True)
                sys.exit(1)
        n += 1
        seed += 1
        if n % 20 == 0:
            print("%d batches ok (seed %d, last batch %d bytes)" % (n, seed, len(data)), flush=True)
    print("fuzz ok: %d batches" % n, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
