"""trp_head_tokens_text_device: the "first k tokens joined by a space, or the text verbatim when it has no
more than k tokens" rule of the reference (bm25_backend.rs:212-229 and :503-532, sampler/mod.rs:1484-1497),
checked against the same rule applied to the oracle's tokens."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import textgen
import triplets_amd as T

pytestmark = pytest.mark.gpu


def expected_heads(exp, k):
    out = []
    for r in range(exp.n_records):
        t0, t1 = int(exp.tok_offsets[r]), int(exp.tok_offsets[r + 1])
        if t1 - t0 <= k:
            out.append(exp.record_text(r))
        else:
            out.append(b" ".join(exp.denoised[int(s):int(s) + int(n)].tobytes()
                                 for s, n in zip(exp.tok_start[t0:t0 + k], exp.tok_len[t0:t0 + k])))
    return out


@pytest.mark.parametrize("enabled", [True, False])
def test_head_tokens_text(enabled):
    import torch
    recs = textgen.corpus(23, 600) + [b"", b"   ", b"one", b"a b c d e f g h i j k l m n o p q r s t u v w x y z " * 40]
    data, offs = O.pack(recs)
    exp = O.batch(data, offs, O.make_config(enabled, 0.35, True, 1024, (64,), 512))
    ctx = T.Context(0)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()   # the library runs on its own stream: inputs must be complete
    dev = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), len(recs), len(data),
                                      T.DenoiserConfig(enabled, 0.35, True), T.ChunkingStrategy())
    R = len(recs)
    for k in (0, 1, 5, 64, 100000):
        t_ptr, o_ptr, n = ctx.head_tokens_text(k)
        o = np.empty(R + 1, dtype=np.uint64)
        T._check(ctx._L.trp_copy_to_host(ctx._h, o.ctypes.data, o_ptr, o.nbytes))
        txt = np.empty(n, dtype=np.uint8)
        if n:
            T._check(ctx._L.trp_copy_to_host(ctx._h, txt.ctypes.data, t_ptr, n))
        want = expected_heads(exp, k)
        assert int(o[-1]) == n == sum(len(w) for w in want)
        for r in range(R):
            assert txt[int(o[r]):int(o[r + 1])].tobytes() == want[r], (k, r)
    assert dev.n_records == R
    ctx.close()
