"""Input that is not valid UTF-8 (the header leaves its classification unspecified, the reference cannot
receive it at all: Rust `&str`): the call must succeed, run twice to the same result, and return arrays
that are consistent with each other.  There is no oracle for the values, so none is consulted."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALPHABET = bytes([0x0A, 0x0D, 0x20, 0x7C, 0x31, 0x39, 0x61, 0x5A, 0xC2, 0xA0, 0xE3, 0x80, 0xF0, 0x9F, 0xE0,
                  0xA4, 0xFF, 0xC0, 0x85, 0x09, 0xE2, 0xED, 0xF4, 0xBF])


def _records(seed, n_records, max_len):
    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(ALPHABET, dtype=np.uint8)
    text = ("wörd 123 | α β 東京 🚀 45 67 89 x\n| a | 1 |\n" * (max_len // 40 + 2)).encode()
    recs = []
    for _ in range(n_records):
        kind, n = int(rng.integers(0, 4)), int(rng.integers(0, max_len))
        if kind == 0:        # every byte value
            recs.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif kind == 1:      # separators, digits, letters and loose pieces of multi-byte sequences
            recs.append(alphabet[rng.integers(0, len(alphabet), n)].tobytes())
        elif kind == 2:      # valid text with bytes overwritten and the record cut inside a character
            b = bytearray(text[:n])
            for _ in range(n // 16):
                b[int(rng.integers(0, n))] = int(rng.integers(0x80, 0x100))
            recs.append(bytes(b))
        else:                # long runs of continuation or lead bytes
            recs.append(bytes([int(rng.integers(0x80, 0x100))]) * n)
    return recs


def _check_consistent(res, data, offs, enabled):
    R, N = len(offs) - 1, len(data)
    assert res.n_records == R
    do, to, co = (np.asarray(res.denoised_offsets, dtype=np.int64), np.asarray(res.tok_offsets, dtype=np.int64),
                  np.asarray(res.chunk_offsets, dtype=np.int64))
    for o, total in ((do, res.n_denoised), (to, res.n_tokens), (co, res.n_chunks)):
        assert len(o) == R + 1 and o[0] == 0 and o[-1] == total and np.all(np.diff(o) >= 0)
    assert res.n_denoised <= N
    if not enabled:
        assert res.n_denoised == N and np.array_equal(do, np.asarray(offs, dtype=np.int64))
        assert np.array_equal(np.asarray(res.denoised), data)
    ts, tl = np.asarray(res.tok_start, dtype=np.int64), np.asarray(res.tok_len, dtype=np.int64)
    assert len(ts) == res.n_tokens and len(tl) == res.n_tokens
    if res.n_tokens:
        rec_of_tok = np.repeat(np.arange(R), np.diff(to))
        assert np.all(tl > 0)
        assert np.all(ts >= do[rec_of_tok]) and np.all(ts + tl <= do[rec_of_tok + 1])
        same = rec_of_tok[1:] == rec_of_tok[:-1]
        assert np.all((ts[1:] > ts[:-1] + tl[:-1])[same])          # at least one separator byte between tokens
    # a record that was dropped has no text, no tokens and no chunks
    dropped = np.asarray(res.dropped).astype(bool)
    assert not np.any(np.diff(do)[dropped]) and not np.any(np.diff(to)[dropped]) and not np.any(np.diff(co)[dropped])
    ch = res.chunks
    if res.n_chunks:
        rec_of_chunk = np.repeat(np.arange(R), np.diff(co))
        assert np.array_equal(ch["record"].astype(np.int64), rec_of_chunk)
        T_r = np.diff(to)[rec_of_chunk]
        assert np.all(ch["tok_begin"] <= ch["tok_end"]) and np.all(ch["tok_end"].astype(np.int64) <= T_r)
    assert np.array_equal(np.diff(co) > 0, np.diff(to) > 0)       # chunks exactly where there are tokens


@pytest.mark.parametrize("seed,n_records,max_len", [(1, 600, 400), (2, 60, 40000), (3, 3000, 70), (4, 8, 600000)])
def test_malformed_input_is_handled(seed, n_records, max_len):
    import oracle_lib as O   # only for its packing helper
    import triplets_amd as T
    recs = _records(seed, n_records, max_len)
    data, offs = O.pack(recs)
    ctx = T.Context(0)
    try:
        for enabled, md in ((True, True), (True, False), (False, True)):
            den = T.DenoiserConfig(enabled, 0.35, md)
            st = T.ChunkingStrategy(max_window_tokens=16, overlap_tokens=[4], summary_fallback_tokens=8)
            a = ctx.preprocess_batch(data, offs, den, st)
            _check_consistent(a, data, offs, enabled)
            b = ctx.preprocess_batch(data, offs, den, st)
            for f in ("denoised", "denoised_offsets", "dropped", "tok_start", "tok_len", "tok_offsets", "chunk_offsets"):
                assert np.array_equal(np.asarray(getattr(a, f)), np.asarray(getattr(b, f))), f
            assert a.chunks.tobytes() == b.chunks.tobytes()
        n = ctx.normalize_batch(data, offs)
        _check_consistent(n, data, offs, True)
        assert not np.any(np.asarray(n.dropped))
    finally:
        ctx.close()
