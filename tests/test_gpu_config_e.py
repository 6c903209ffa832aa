"""BASELINE config E on one GPU: the 64e9-byte corpus cut into 8 record-range shards (the bench's own
recipe: OCR-noisy markdown generator, seed 45 + 1000 * rank + slice), every shard's first slice of 4e9
bytes run one after the other on the single GPU.

Per shard:
  * structural invariants over the WHOLE slice (offsets monotone and consistent with the totals, dropped <=>
    empty text, token table ordered / non-overlapping / inside the text, no whitespace byte at token edges,
    chunk counts and descriptors follow the closed-form window arithmetic),
  * exact comparison with the oracle on a record range of 1e9 bytes (start of the range rotates with the
    shard so that the eight ranges cover the start, the middle and the end of a slice),
and over all shards: concat_results of the eight results is continuous (global offsets monotone, totals
add up, every shard's arrays are found again at its base).

TRP_E_SLICE_BYTES / TRP_E_SAMPLE_BYTES shrink the run for quick checks."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
import triplets_amd as T
from parity import assert_same
from test_gpu_large import expected_chunk_counts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))

pytestmark = pytest.mark.gpu

TOTAL = 64_000_000_000
WORLD = 8
SLICE = int(os.environ.get("TRP_E_SLICE_BYTES", "4000000000"))
SAMPLE = int(os.environ.get("TRP_E_SAMPLE_BYTES", "1000000000"))
STEP = 1 << 26   # token-table checks run over this many tokens at a time (bounded temporaries)

WS_ASCII = np.zeros(256, dtype=bool)
WS_ASCII[[9, 10, 11, 12, 13, 32]] = True


def check_token_table(name, den, tok_start, tok_len, tok_off, n_denoised):
    n = len(tok_start)
    assert n == int(tok_off[-1])
    first = np.zeros(n + 1, dtype=bool)            # first token of a record
    first[tok_off[:-1][np.diff(tok_off.astype(np.int64)) > 0].astype(np.int64)] = True
    prev_end = 0
    for a in range(0, n, STEP):
        b = min(n, a + STEP)
        s = tok_start[a:b].astype(np.int64)
        ln = tok_len[a:b].astype(np.int64)
        assert ln.min() > 0, name
        e = s + ln
        assert s[0] >= prev_end, name
        gaps = s[1:] - e[:-1]
        # tokens of one record are separated by at least one byte; records may touch
        assert np.all(gaps >= np.where(first[a + 1:b], 0, 1)), name + ": tokens overlap or are unordered"
        prev_end = int(e[-1])
        # no ASCII whitespace at either end of a token (multi-byte whitespace is covered by the oracle range)
        assert not WS_ASCII[den[s]].any() and not WS_ASCII[den[e - 1]].any(), name
    assert prev_end <= n_denoised, name


def fetch_range(dev, doff, toff, coff, a, b):
    """Records [a, b) of a device result as a stand-alone host result."""
    d0, t0, c0 = int(doff[a]), int(toff[a]), int(coff[a])
    ch = dev.fetch("chunks", int(coff[b]) - c0, T.CHUNK_DTYPE, c0)
    ch["record"] -= np.uint32(a)
    return T.BatchResult(
        denoised=dev.fetch("denoised", int(doff[b]) - d0, np.uint8, d0), denoised_offsets=doff[a:b + 1] - np.uint64(d0),
        dropped=dev.fetch("dropped", b - a, np.uint8, a),
        tok_start=(dev.fetch("tok_start", int(toff[b]) - t0, np.uint32, t0).astype(np.int64) - d0).astype(np.uint32),
        tok_len=dev.fetch("tok_len", int(toff[b]) - t0, np.uint32, t0), tok_offsets=toff[a:b + 1] - np.uint64(t0),
        chunks=ch, chunk_offsets=coff[a:b + 1] - np.uint64(c0), n_records=b - a, n_denoised=int(doff[b]) - d0,
        n_tokens=int(toff[b]) - t0, n_chunks=int(coff[b]) - c0)


def test_config_e_eight_shards_on_one_gpu():
    import corpusgen
    import torch
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()   # reference defaults 1024 / [64] / 512
    ocfg = O.make_config(True, 0.35, True, 1024, (64,), 512)
    ctx = T.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    light = []      # per shard: offsets, dropped and chunks (the arrays concat_results shifts) + totals
    for rank in range(WORLD):
        lo, hi = TOTAL * rank // WORLD, TOTAL * (rank + 1) // WORLD
        n_slices = max(1, -(-(hi - lo) // 4_000_000_000))
        size = min(SLICE, (hi - lo) // n_slices)                       # slice 0 of the shard
        data, offs = corpusgen.config_c(size, 45 + 1000 * rank + 0, 16)
        R, N = len(offs) - 1, int(offs[-1])
        name = "config E shard %d" % rank
        d_bytes = torch.from_numpy(data).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        dev = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), R, N, den, st)
        doff = dev.fetch("denoised_offsets", R + 1, np.uint64)
        toff = dev.fetch("tok_offsets", R + 1, np.uint64)
        coff = dev.fetch("chunk_offsets", R + 1, np.uint64)
        dropped = dev.fetch("dropped", R, np.uint8)
        chunks = dev.fetch("chunks", dev.n_chunks, T.CHUNK_DTYPE)
        # ---- structure of the whole slice
        for arr, total in ((doff, dev.n_denoised), (toff, dev.n_tokens), (coff, dev.n_chunks)):
            assert arr[0] == 0 and int(arr[-1]) == int(total) and np.all(arr[1:] >= arr[:-1]), name
        assert 0 < int(dev.n_denoised) <= N
        sizes = np.diff(doff.astype(np.int64))
        assert np.array_equal(dropped != 0, sizes == 0), name
        T_r = np.diff(toff.astype(np.int64))
        assert np.all(T_r[dropped != 0] == 0) and np.all(T_r[dropped == 0] > 0), name
        assert np.array_equal(np.diff(coff.astype(np.int64)), expected_chunk_counts(T_r, 1024, (64,), 512)), name
        assert np.all(chunks["tok_end"] > chunks["tok_begin"])
        assert np.all(chunks["tok_end"].astype(np.int64) <= T_r[chunks["record"]])
        assert np.all(np.diff(chunks["record"].astype(np.int64)) >= 0)
        den_h = dev.fetch("denoised", dev.n_denoised, np.uint8)
        check_token_table(name, den_h, dev.fetch("tok_start", dev.n_tokens, np.uint32),
                          dev.fetch("tok_len", dev.n_tokens, np.uint32), toff, int(dev.n_denoised))
        del den_h
        # ---- exact comparison with the oracle on a 1e9-byte record range
        sample = min(SAMPLE, N)
        start = (N - sample) * rank // (WORLD - 1)
        a = int(np.searchsorted(offs, np.uint64(start), side="left"))
        b = int(np.searchsorted(offs, np.uint64(min(N, start + sample)), side="right")) - 1
        a = min(a, R - 1)
        b = max(min(b, R), a + 1)
        d, o = T.shard_slice(data, offs, a, b)
        exp = O.batch(d, o, ocfg, n_threads=16)
        assert_same(exp, fetch_range(dev, doff, toff, coff, a, b), None, "%s records [%d,%d)" % (name, a, b))
        del exp
        z8, z32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        light.append(T.BatchResult(denoised=z8, denoised_offsets=doff, dropped=dropped, tok_start=z32, tok_len=z32,
                                   tok_offsets=toff, chunks=chunks, chunk_offsets=coff, n_records=R,
                                   n_denoised=int(dev.n_denoised), n_tokens=int(dev.n_tokens),
                                   n_chunks=int(dev.n_chunks)))
        del d_bytes, d_offs, data, offs
    ctx.close()
    # ---- the eight shard results as one: continuity of the concatenation
    whole = T.concat_results(light)
    assert whole.n_records == sum(p.n_records for p in light)
    for arr, total in ((whole.denoised_offsets, whole.n_denoised), (whole.tok_offsets, whole.n_tokens),
                       (whole.chunk_offsets, whole.n_chunks)):
        assert len(arr) == whole.n_records + 1
        assert arr[0] == 0 and int(arr[-1]) == total and np.all(arr[1:] >= arr[:-1])
    assert np.all(np.diff(whole.chunks["record"].astype(np.int64)) >= 0)
    r0 = c0 = 0
    bd = bt = 0
    for p in light:
        r1, c1 = r0 + p.n_records, c0 + p.n_chunks
        assert np.array_equal(whole.denoised_offsets[r0:r1 + 1] - np.uint64(bd), p.denoised_offsets)
        assert np.array_equal(whole.tok_offsets[r0:r1 + 1] - np.uint64(bt), p.tok_offsets)
        assert np.array_equal(whole.chunk_offsets[r0:r1 + 1] - np.uint64(c0), p.chunk_offsets)
        assert np.array_equal(whole.dropped[r0:r1], p.dropped)
        assert np.array_equal(whole.chunks["record"][c0:c1] - np.uint32(r0), p.chunks["record"])
        r0, c0, bd, bt = r1, c1, bd + p.n_denoised, bt + p.n_tokens
