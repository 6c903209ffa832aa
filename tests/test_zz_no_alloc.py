"""GPU test of the entry point that allocates nothing (trp_preprocess_batch_device_noalloc): its results, in the
caller's buffers, equal those of trp_preprocess_batch_device on the same batch, bit for bit, on two calls in a
row - the second one smaller, so that whatever the first call left in the shared scratch array is still there."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import textgen
from parity import assert_same

pytestmark = pytest.mark.gpu


def _corpus(seed, n_records):
    recs = textgen.corpus(seed, n_records)
    # digit-heavy lines that cross tile edges and long lines, so that the work list and the tile carries are used
    recs.append(("x1 " * 9000 + "\n" + "12 34 ab 56 " * 3000 + "\n| a | 1 |\n|---|---|\n").encode())
    return recs


def _buffers(T, torch, n_bytes, n_records, chunk_cap):
    tok_cap = int(T.load_library().trp_token_bound(n_bytes, n_records))
    t = {
        "denoised": torch.full((n_bytes + 64,), 0xEE, dtype=torch.uint8, device="cuda"),
        "denoised_offsets": torch.full((n_records + 1,), -1, dtype=torch.int64, device="cuda"),
        "dropped": torch.full((n_records + 1,), 0xEE, dtype=torch.uint8, device="cuda"),
        "tok_start": torch.full((tok_cap,), -1, dtype=torch.int32, device="cuda"),
        "tok_len": torch.full((tok_cap,), -1, dtype=torch.int32, device="cuda"),
        "tok_offsets": torch.full((n_records + 1,), -1, dtype=torch.int64, device="cuda"),
        "chunks": torch.full((chunk_cap * 8,), -1, dtype=torch.int32, device="cuda"),
        "chunk_offsets": torch.full((n_records + 1,), -1, dtype=torch.int64, device="cuda"),
    }
    b = T._OutBuffers()
    for k, v in t.items():
        setattr(b, k, v.data_ptr())
    b.denoised_cap, b.tok_cap, b.chunk_cap = n_bytes + 64, tok_cap, chunk_cap
    return t, b


def _compare(T, torch, ctx, recs, den, st):
    data, offs = O.pack(recs)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    R, N = len(recs), int(len(data))
    ref = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), R, N, den, st).to_host()
    keep, bufs = _buffers(T, torch, N, R, int(ref.n_chunks) + 3)
    torch.cuda.synchronize()
    got = ctx.preprocess_batch_device_noalloc(d_bytes.data_ptr(), d_offs.data_ptr(), R, N, den, st, bufs)
    if den.enabled:
        assert got.denoised == keep["denoised"].data_ptr()
    assert got.tok_start == keep["tok_start"].data_ptr() and got.chunks == keep["chunks"].data_ptr()
    got = got.to_host()
    assert (got.n_denoised, got.n_tokens, got.n_chunks) == (ref.n_denoised, ref.n_tokens, ref.n_chunks)
    # the batch is now the context's "last device batch": the chunk text renderer reads it through the caller's buffers
    tptr, optr, nbytes = ctx.materialize_chunk_text()
    toff = np.empty(got.n_chunks + 1, dtype=np.uint64)
    text = np.empty(nbytes, dtype=np.uint8)
    L = T.load_library()
    assert L.trp_copy_to_host(ctx._h, toff.ctypes.data, optr, toff.nbytes) == 0
    if nbytes:
        assert L.trp_copy_to_host(ctx._h, text.ctypes.data, tptr, text.nbytes) == 0
    for c in range(0, got.n_chunks, max(1, got.n_chunks // 200)):
        assert text[int(toff[c]):int(toff[c + 1])].tobytes() == ref.chunk_text(c), c
    assert_same(ref, got, recs, "no-alloc entry point vs trp_preprocess_batch_device")
    # and both against the oracle
    exp = O.batch(data, offs, O.make_config(den.enabled, den.max_digit_ratio, den.strip_markdown,
                                            st.max_window_tokens, tuple(st.overlap_tokens), st.summary_fallback_tokens))
    assert_same(exp, got, recs, "no-alloc entry point vs oracle")
    # nothing was written past the totals (the text buffer is working space up to n_bytes, untouched beyond)
    if den.enabled:
        assert int(keep["denoised"][N:].min()) == 0xEE
    assert int(keep["tok_start"][got.n_tokens + 3:].max()) == -1 if got.n_tokens + 3 < len(keep["tok_start"]) else True
    return keep


def test_no_alloc_entry_point_matches_the_allocating_one():
    import torch
    import triplets_amd as T
    ctx = T.Context(0)
    try:
        st = T.ChunkingStrategy(max_window_tokens=16, overlap_tokens=[4, 2], summary_fallback_tokens=8)
        den = T.DenoiserConfig(True, 0.35, True)
        _compare(T, torch, ctx, _corpus(7, 1500), den, st)            # first call: the larger batch
        _compare(T, torch, ctx, _corpus(8, 90), den, st)              # second call: a smaller one
        _compare(T, torch, ctx, _corpus(9, 40), T.DenoiserConfig(False, 0.35, True), T.ChunkingStrategy())   # pass-through
        _compare(T, torch, ctx, [b""], den, st)                       # empty batch text
    finally:
        ctx.close()


def test_no_alloc_entry_point_rejects_what_it_cannot_hold():
    import torch
    import triplets_amd as T
    ctx = T.Context(0)
    try:
        den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy()
        n = T.NOALLOC_MAX_BYTES + 64
        big = torch.zeros(n, dtype=torch.uint8, device="cuda")
        offs = torch.tensor([0, n], dtype=torch.int64, device="cuda")
        keep, bufs = _buffers(T, torch, 1024, 1, 4)
        torch.cuda.synchronize()
        with pytest.raises(T.TripletsAmdError, match="TRP_NOALLOC_MAX_BYTES"):
            ctx.preprocess_batch_device_noalloc(big.data_ptr(), offs.data_ptr(), 1, n, den, st, bufs)
        # too few chunk descriptors: the error names the number needed
        recs = [b"word " * 5000]
        data, o = O.pack(recs)
        d_bytes, d_offs = torch.from_numpy(data).cuda(), torch.from_numpy(o.astype(np.int64)).cuda()
        keep, bufs = _buffers(T, torch, len(data), 1, 2)
        torch.cuda.synchronize()
        with pytest.raises(T.TripletsAmdError, match="chunk descriptor buffer too small"):
            ctx.preprocess_batch_device_noalloc(d_bytes.data_ptr(), d_offs.data_ptr(), 1, len(data), den, st, bufs)
    finally:
        ctx.close()
