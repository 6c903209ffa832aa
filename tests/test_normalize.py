"""normalize_inline_whitespace (reference src/utils.rs:18-33) as a batched GPU mode.

Oracle: orc_normalize_inline_whitespace per record, pinned by the reference's own test vector
(src/utils.rs:166-170, both platform newlines) and by the definition "tokens joined by one space".
CPU: the host simulator of the GPU algorithm against the oracle.  GPU: the library against the oracle.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import sim_lib as S
import textgen
import triplets_amd as T
from parity import assert_same

EDGE = [
    b"", b" ", b"\n\n", b"a", b" a", b"a ", b"  a  b  ", b"Alpha\n\n  Beta\tGamma", b"Alpha\r\n\r\n  Beta\tGamma",
    "x　y z w\u0085v".encode(), "　　lead".encode(), "trail ".encode(),
    b"a" * 63 + b" " + b"b" * 64, b"a" * 64 + b" " * 64 + b"b", b" " * 200 + b"x", b"x" + b" " * 200,
    b"tok " * 5000, (b"y" * 70 + b"\t\t") * 300, b"\t".join([b"w%d" % i for i in range(3000)]),
]


def oracle_normalize(text: bytes) -> bytes:
    L = O.lib()
    buf = C.create_string_buffer(len(text) + 1)
    n = L.orc_normalize_inline_whitespace(text, len(text), buf)
    return buf.raw[:n]


def expected(records):
    norm = [oracle_normalize(r) for r in records]
    data, offs = O.pack(norm)
    exp = O.batch(data, offs, O.make_config(False, 0.35, True, 1024, (64,), 512))
    return norm, exp


def test_reference_vector_and_definition():
    for nl in (b"\n", b"\r\n"):
        assert oracle_normalize(b"Alpha" + nl + nl + b"  Beta\tGamma") == b"Alpha Beta Gamma"
    for r in EDGE + textgen.corpus(3, 200):
        toks = [r[a:a + n] for a, n in O.tokenize(r)]
        assert oracle_normalize(r) == b" ".join(toks)


def sim_normalize(records, tile_blocks):
    L = S.lib()
    L.trp_sim_normalize.restype = C.c_int
    L.trp_sim_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(O.OrcResult), C.c_int]
    data, offs = O.pack(records)
    res = O.OrcResult()
    rc = L.trp_sim_normalize(data.ctypes.data if data.size else None, offs.ctypes.data, len(records), C.byref(res),
                             tile_blocks)
    assert rc == 0
    R = len(records)

    def arr(ptr, n, dt):
        if n == 0:
            return np.zeros(0, dtype=dt)
        if dt == O.CHUNK_DTYPE:
            return np.frombuffer(C.string_at(ptr, n * C.sizeof(O.OrcChunk)), dtype=dt).copy()
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    out = O.BatchResult(
        denoised=arr(res.denoised, res.n_denoised, np.uint8), denoised_offsets=arr(res.denoised_offsets, R + 1, np.uint64),
        dropped=arr(res.dropped, R, np.uint8), tok_start=arr(res.tok_start, res.n_tokens, np.uint32),
        tok_len=arr(res.tok_len, res.n_tokens, np.uint32), tok_offsets=arr(res.tok_offsets, R + 1, np.uint64),
        chunks=arr(res.chunks, res.n_chunks, O.CHUNK_DTYPE), chunk_offsets=arr(res.chunk_offsets, R + 1, np.uint64),
        n_records=R, n_denoised=int(res.n_denoised), n_tokens=int(res.n_tokens), n_chunks=int(res.n_chunks))
    L.trp_sim_free(C.byref(res))
    return out


@pytest.mark.parametrize("tile_blocks", [1, 2, 3, 256])
def test_simulator_matches_oracle(tile_blocks):
    for recs in (EDGE, textgen.corpus(17, 300), [b"".join(EDGE)], textgen.script_corpus(18, 150)):
        norm, exp = expected(recs)
        got = sim_normalize(recs, tile_blocks)
        assert_same(exp, got, recs, "normalize sim tile_blocks=%d" % tile_blocks)


@pytest.mark.gpu
def test_gpu_matches_oracle():
    ctx = T.Context(0)
    for recs in (EDGE, textgen.corpus(17, 2000), [b"".join(EDGE)], [], [b""], textgen.script_corpus(18, 600)):
        norm, exp = expected(recs)
        data, offs = O.pack(recs)
        got = ctx.normalize_batch(data, offs)
        assert_same(exp, got, recs, "normalize gpu")
    assert T.normalize_inline_whitespace("Alpha\r\n\r\n  Beta\tGamma", ctx) == "Alpha Beta Gamma"
    assert T.normalize_inline_whitespace("   ", ctx) == ""
    ctx.close()


@pytest.mark.gpu
def test_gpu_large_batch_properties():
    """256 MB of config-C text: the output is the pass-through tokens joined by single spaces."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "corpusgen"))
    import corpusgen
    data, offs = corpusgen.config_c(256_000_000, 43, 16)
    R, N = len(offs) - 1, int(offs[-1])
    ctx = T.Context(0)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()   # the library runs on its own stream: inputs must be complete
    st = T.ChunkingStrategy()
    thru = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), R, N, T.DenoiserConfig(False, 0.35, True), st)
    t_len = thru.fetch("tok_len", thru.n_tokens, np.uint32)
    t_off = thru.fetch("tok_offsets", R + 1, np.uint64)
    norm = ctx.normalize_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), R, N).to_host()
    assert norm.n_tokens == int(thru.n_tokens)
    assert np.array_equal(norm.tok_len, t_len) and np.array_equal(norm.tok_offsets, t_off)
    assert not norm.dropped.any()
    # sizes: token bytes + one space between consecutive tokens of a record
    T_r = np.diff(t_off.astype(np.int64))
    csum = np.concatenate([[0], np.cumsum(t_len.astype(np.int64))])
    sizes = csum[t_off[1:].astype(np.int64)] - csum[t_off[:-1].astype(np.int64)] + np.maximum(T_r - 1, 0)
    assert np.array_equal(np.diff(norm.denoised_offsets.astype(np.int64)), sizes)
    # every byte between two tokens of a record is exactly one space, and tokens hold no whitespace byte
    starts = norm.tok_start.astype(np.int64)
    ends = starts + norm.tok_len
    gap = starts[1:] - ends[:-1]
    first_of_record = np.zeros(norm.n_tokens, dtype=bool)
    first_of_record[t_off[:-1][T_r > 0].astype(np.int64)] = True
    assert np.all(gap[~first_of_record[1:]] == 1)
    assert np.all(gap[first_of_record[1:]] == 0)
    assert np.all(norm.denoised[ends[:-1][~first_of_record[1:]]] == 0x20)
    # exact comparison with the oracle on the first records
    k = 2000
    recs = [data[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(k)]
    exp_text = b"".join(oracle_normalize(r) for r in recs)
    assert norm.denoised[:int(norm.denoised_offsets[k])].tobytes() == exp_text
    ctx.close()
