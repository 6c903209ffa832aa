"""Batches of a few tiles take one launch (k_small_batch: the bodies of the path's kernels run one after the other
by one workgroup) instead of the kernel sequence.  Both ways must give the oracle's results: the same small batches
go through a context that takes the single launch and through one that was created with TRIPLETS_AMD_NO_SMALL=1
(the full sequence), with the denoiser on and off, in normalize mode, and through the entry point that allocates
nothing."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import textgen
from parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import triplets_amd
    return triplets_amd


@pytest.fixture(scope="module")
def both(T):
    small = T.Context(0)
    os.environ["TRIPLETS_AMD_NO_SMALL"] = "1"
    try:
        full = T.Context(0)
    finally:
        del os.environ["TRIPLETS_AMD_NO_SMALL"]
    yield small, full
    small.close()
    full.close()


def _batches():
    rng = random.Random(12)
    out = [[b""], [b"", b"", b""], [b"x"], [b"one two three"], [b" \n\n "], [b"12 34 56\n7 8 9"]]
    for seed in range(40):
        n = rng.choice([1, 2, 5, 17, 60])
        recs = textgen.corpus(1000 + seed, n)
        if seed % 3 == 0:
            recs += textgen.script_corpus(seed, max(1, n // 3))
        while sum(len(r) for r in recs) > 32500 and len(recs) > 1:   # at most two tiles: the single-launch range
            recs.pop()
        out.append(recs)
    # exactly at and just beyond the limits of the single launch
    word = b"alpha 12 beta 345 gamma\n"
    out.append([word * (32768 // len(word))])
    out.append([word * (32768 // len(word) + 3)])
    out.append([b"w"] * 4096)
    out.append([b"w"] * 4097)
    # long lines and a long letterless stretch inside 32 KiB: the tile seeds need the summaries (the batch is run again)
    out.append([b"12 34\n" * 5000 + b"word\n", (" ".join("w%d" % i for i in range(4000))).encode()])
    return out


CONFIGS = [(True, 0.35, True, 1024, (64,), 512), (True, 0.6, False, 7, (3, 1), 2), (False, 0.35, True, 16, (4,), 8),
           (True, 0.0, True, 1, (), 0)]


def test_single_launch_equals_the_kernel_sequence_and_the_oracle(T, both):
    small, full = both
    for recs in _batches():
        if sum(len(r) for r in recs) > 200000:
            continue
        data, offs = O.pack(recs)
        for cfg in CONFIGS:
            den = T.DenoiserConfig(cfg[0], cfg[1], cfg[2])
            st = T.ChunkingStrategy(max_window_tokens=cfg[3], overlap_tokens=list(cfg[4]), summary_fallback_tokens=cfg[5])
            exp = O.batch(data, offs, O.make_config(*cfg))
            assert_same(exp, small.preprocess_batch(data, offs, den, st), recs, "single launch %s" % (cfg,))
            assert_same(exp, full.preprocess_batch(data, offs, den, st), recs, "kernel sequence %s" % (cfg,))


def test_single_launch_normalize_mode(T, both):
    from test_normalize import oracle_normalize
    small, full = both
    for recs in _batches()[:30]:
        data, offs = O.pack(recs)
        a, b = small.normalize_batch(data, offs), full.normalize_batch(data, offs)
        for r in range(len(recs)):
            want = oracle_normalize(recs[r])
            assert a.record_text(r) == want and b.record_text(r) == want, r
        assert np.array_equal(a.tok_offsets, b.tok_offsets) and np.array_equal(a.tok_start, b.tok_start)
        assert np.array_equal(a.chunk_offsets, b.chunk_offsets)


def test_single_launch_is_used_by_the_device_entry_points(T, both):
    """The profile of a small device batch names one kernel; a large one names the sequence."""
    import torch
    small, _ = both
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
    for nrec, want_small in ((3, True), (3000, False)):
        recs = textgen.corpus(77, nrec)
        data, offs = O.pack(recs)
        db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
        do = torch.from_numpy(offs.astype(np.int64)).cuda()
        torch.cuda.synchronize()
        small.set_profiling(True)
        got = small.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st).to_host()
        names = [n for n, _ in small.profile()]
        small.set_profiling(False)
        assert ("k_small_batch" in names) == want_small, names
        assert_same(O.batch(data, offs, O.make_config(True, 0.35, True, 1024, (64,), 512)), got, recs, "device %d" % nrec)


def test_small_host_batches_come_back_gathered_or_by_copies(T, both):
    """A small batch in host memory is fetched by the single launch itself and its result arrays are gathered into
    one page-locked block; results too large for the block (many one-token windows) come back by copy commands.
    Either way: the oracle's arrays, for every output selection."""
    small, full = both
    many = [b"a " * 16000]                                   # 16000 tokens in 32000 bytes
    prose = textgen.corpus(5, 6)
    while sum(len(r) for r in prose) > 32000:
        prose.pop()
    longer = textgen.corpus(9, 40)                           # beyond the single launch: the kernel sequence, short result arrays packed by one launch
    while sum(len(r) for r in longer) > 65000:
        longer.pop()
    word = b"alpha 12 beta 345 gamma\n"
    cases = [(longer, (True, 0.35, True, 64, (8,), 16)),
             ([word * (65536 // len(word))], (True, 0.35, True, 64, (8,), 16)),
             ([word * (65536 // len(word) + 2)], (True, 0.35, True, 64, (8,), 16)),   # just beyond: the kernel sequence
             (prose, (True, 0.35, True, 16, (4,), 8)),          # gathered
             (prose, (False, 0.35, True, 16, (4,), 8)),         # gathered, the text is the input itself
             (many, (True, 0.35, True, 2, (1,), 0)),            # ~16000 windows: 640 KiB, still gathered
             (many, (True, 0.35, True, 2, (1, 0, 1), 0)),       # ~40000 windows: beyond the block
             (many, (False, 0.35, True, 2, (1, 0, 1), 0))]
    for recs, cfg in cases:
        data, offs = O.pack(recs)
        den = T.DenoiserConfig(cfg[0], cfg[1], cfg[2])
        st = T.ChunkingStrategy(max_window_tokens=cfg[3], overlap_tokens=list(cfg[4]), summary_fallback_tokens=cfg[5])
        exp = O.batch(data, offs, O.make_config(*cfg))
        ref = full.preprocess_batch(data, offs, den, st)
        assert_same(exp, ref, recs, "kernel sequence %s" % (cfg,))
        assert_same(exp, small.preprocess_batch(data, offs, den, st), recs, "single launch %s" % (cfg,))
        for mask in (0, T.OUT_TEXT, T.OUT_TOKENS, T.OUT_CHUNKS, T.OUT_TOKENS | T.OUT_CHUNKS, T.OUT_ALL | T.OUT_CHUNK_TEXT):
            got = small.preprocess_batch_select(data, offs, den, st, mask)
            for k in ("denoised_offsets", "tok_offsets", "chunk_offsets", "dropped"):
                assert np.array_equal(getattr(got, k), getattr(ref, k)), (cfg, mask, k)
            for flag, names in ((T.OUT_TEXT, ("denoised",)), (T.OUT_TOKENS, ("tok_start", "tok_len")), (T.OUT_CHUNKS, ("chunks",))):
                for k in names:
                    if mask & flag:
                        assert np.array_equal(getattr(got, k), getattr(ref, k)), (cfg, mask, k)
                    else:
                        assert len(getattr(got, k)) == 0, (cfg, mask, k)
            if mask & T.OUT_CHUNK_TEXT:
                o = got.chunk_text_offsets
                for c in (0, ref.n_chunks // 2, ref.n_chunks - 1):
                    assert got.chunk_text_bytes[int(o[c]):int(o[c + 1])].tobytes() == ref.chunk_text(c), (cfg, c)
