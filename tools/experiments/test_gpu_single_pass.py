"""The single-pass token table of the pass-through mode (k_pass_single, switched on with TRIPLETS_AMD_SINGLE_PASS=4
or 8 when the context is created): the same arrays as the kernel sequence and the oracle, on batches whose tokens,
records and lengths meet the piece edges (64 KiB / 128 KiB) in every way."""
import os

import numpy as np
import pytest

import oracle_lib as O
import textgen
from parity import assert_same

pytestmark = pytest.mark.gpu

PIECES = (4, 8)


@pytest.fixture(scope="module")
def T():
    import triplets_amd
    return triplets_amd


@pytest.fixture(scope="module")
def contexts(T):
    out = {0: T.Context(0)}
    for k in PIECES:
        os.environ["TRIPLETS_AMD_SINGLE_PASS"] = str(k)
        try:
            out[k] = T.Context(0)
        finally:
            del os.environ["TRIPLETS_AMD_SINGLE_PASS"]
    yield out
    for c in out.values():
        c.close()


def _batches():
    out = []
    for seed in (1, 2, 3):
        recs = textgen.corpus(4000 + seed, 120) + textgen.script_corpus(seed, 20)
        out.append(recs)
    word = b"alpha beta\tgamma\xe3\x80\x80delta\n"
    for edge in (65536, 131072):
        # a token that ends exactly at / just before / just after a piece edge; a record start on the edge
        for d in (-2, -1, 0, 1, 2):
            filler = word * ((edge + d) // len(word))
            pad = b"x" * (edge + d - len(filler))
            out.append([filler + pad + b" tail words follow here", b"next record " * 3000])
            out.append([filler + pad, b"second starts on the edge " * 3000])
    out.append([b"y" * 300000 + b" after the long token", b"z" * 70000])            # tokens longer than a piece
    out.append([b" " * 200000 + b"w"])                                               # pieces without a token
    out.append([b"a"] * 150000)                                                      # one-byte records
    out.append([b"a b " * 50000])                                                    # densest token table
    out.append([(" ".join("w%d" % i for i in range(60000))).encode()])
    out.append([b"q" * (16384 * 4 * 3)])                                             # exactly three pieces of four tiles
    out.append([b"one two three"] * 3 + [b""] * 5 + [b"four " * 40000])
    return out


CONFIGS = [(False, 0.35, True, 1024, (64,), 512), (False, 0.35, True, 7, (3, 1), 2)]


def test_single_pass_equals_the_kernel_sequence_and_the_oracle(T, contexts):
    for recs in _batches():
        data, offs = O.pack(recs)
        for cfg in CONFIGS:
            den = T.DenoiserConfig(cfg[0], cfg[1], cfg[2])
            st = T.ChunkingStrategy(max_window_tokens=cfg[3], overlap_tokens=list(cfg[4]), summary_fallback_tokens=cfg[5])
            exp = O.batch(data, offs, O.make_config(*cfg), n_threads=8)
            for k, ctx in contexts.items():
                assert_same(exp, ctx.preprocess_batch(data, offs, den, st), recs, "pieces of %d tiles, %s" % (k, cfg))


def test_single_pass_kernel_is_the_one_that_runs(T, contexts):
    import torch
    recs = textgen.corpus(9, 400)
    data, offs = O.pack(recs)
    db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
    do = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    den, st = T.DenoiserConfig(False, 0.35, True), T.ChunkingStrategy()
    for k in PIECES:
        ctx = contexts[k]
        ctx.set_profiling(True)
        ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st)
        names = [n for n, _ in ctx.profile()]
        ctx.set_profiling(False)
        assert "k_pass_single" in names and "k_count_tokens" not in names, names
