"""The two steps right after the path that re-use its tokenizer: decorate_chunk's token-budget truncation
(src/sampler/mod.rs:1470-1508) and the BM25 text helpers (src/sampler/backends/bm25_backend.rs:219-225,
:503-532).  CPU part: the oracle's restatements against what the reference's own tests assert
(tests/golden/after_path.json).  GPU part: the device entry points against the oracle."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
import textgen

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    with open(os.path.join(G, "after_path.json"), encoding="utf-8") as f:
        return json.load(f)


def _ntok(b: bytes) -> int:
    return len(O.tokenize(b))


def test_oracle_decorate_chunk_matches_the_reference_tests():
    for case in _golden()["decorate"]:
        a = case["asserted"]
        text, prefix = case["text"].encode(), case["prefix"].encode()
        if "tokens_before" in a:
            assert _ntok(text) == a["tokens_before"], case["ref"]
        out, est = O.decorate_chunk(text, prefix, case["max_window_tokens"], case["newline"].encode())
        assert est == a["tokens_estimate"], case["ref"]
        if "tokens_after" in a:
            assert _ntok(out) == a["tokens_after"], case["ref"]
        if "starts_with" in a:
            assert out.startswith(a["starts_with"].encode()), case["ref"]


def test_oracle_record_bm25_text_matches_the_reference_tests():
    g = _golden()
    for case in g["record_bm25_text"]:
        secs = [(h.encode() if h is not None else None, t.encode()) for h, t in case["sections"]]
        for a in case["asserted"]:
            out = O.record_bm25_text(case["id"].encode(), secs, a["max_tokens"], case["newline"].encode())
            if "tokens" in a:
                assert _ntok(out) == a["tokens"], case["ref"]
            if "tokens_more_than" in a:
                assert _ntok(out) > a["tokens_more_than"], case["ref"]
            if "contains" in a:
                assert a["contains"].encode() in out, case["ref"]
    # the paths the reference's tests do not reach (restated from the code): id fallback, blank parts
    assert O.record_bm25_text(b"rid", [(b"  ", b"\t\n")], 7) == b"rid"
    assert O.record_bm25_text(b"rid", [], 0) == b"rid"
    assert O.record_bm25_text(b"rid", [(None, b"a  b"), (b" ", b"c")], 0) == b"a  b\nc\n"
    assert O.record_bm25_text(b"rid", [(b"H 1", b"a  b"), (None, b"c d")], 3, b"\r\n") == b"H 1 a"
    for case in g["query_cap"]:
        text = case["text"].encode()
        assert _ntok(text) == case["asserted"]["tokens_before"]
        capped = O.cap_tokens(text, case["limit"])
        assert _ntok(capped) == case["limit"] and capped == b" ".join(text.split()[:case["limit"]])
        assert O.cap_tokens(b" a  b ", 2) == b" a  b "           # at most `limit` tokens: the text itself


# ---- GPU ----------------------------------------------------------------------------------------------
def _prefix(rng):
    kind = rng.randrange(6)
    if kind == 0:
        return b"meta: source=unit"
    if kind == 1:
        return b"meta: " + b" | ".join(b"k%d=v %d" % (i, rng.randrange(100)) for i in range(rng.randrange(1, 30)))
    if kind == 2:
        return b""
    if kind == 3:
        return "meta:　date=2025-01-01 |  a=b ".encode()
    if kind == 4:
        return b"meta: long=" + b" ".join(b"w%d" % i for i in range(rng.randrange(1, 80)))
    return b"   "


@pytest.mark.gpu
def test_gpu_decorate_chunks_match_the_oracle():
    import torch
    import triplets_amd as T
    rng = random.Random(5)
    recs = textgen.corpus(41, 300) + [b"one two three", b"", b"solo", b"x " * 500]
    data, offs = O.pack(recs)
    for enabled, mwt, ov, sft in ((True, 12, (3, 5), 4), (False, 40, (8,), 16), (True, 1024, (64,), 512)):
        exp = O.batch(data, offs, O.make_config(enabled, 0.35, True, mwt, ov, sft))
        ctx = T.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
        do = torch.from_numpy(offs.astype(np.int64)).cuda()
        dev = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data),
                                          T.DenoiserConfig(enabled, 0.35, True), T.ChunkingStrategy(mwt, list(ov), 0.35, sft))
        chunks = dev.fetch("chunks", dev.n_chunks, T.CHUNK_DTYPE)
        assert dev.n_chunks == exp.n_chunks
        sel = [rng.randrange(exp.n_chunks) for _ in range(400)] + [0, exp.n_chunks - 1]
        prefixes = [_prefix(rng) for _ in sel]
        for window in (mwt, 5, 0, 1):
            for nl in (b"\n", b"\r\n"):
                got = T.decorate_chunks(ctx, chunks, sel, prefixes, window, nl)
                for i, c in enumerate(sel):
                    want = O.decorate_chunk(exp.chunk_text(c), prefixes[i], window, nl)
                    assert got[i] == want, (enabled, window, c, prefixes[i])
        # the host-memory entry point (trp_chunk_bodies: batch + bodies in one call) gives the same bodies as the
        # device one on the resident batch
        budgets = [rng.choice([0, 1, 3, 7, 0xFFFFFFFF]) for _ in sel]
        den_h, st_h = T.DenoiserConfig(enabled, 0.35, True), T.ChunkingStrategy(mwt, list(ov), 0.35, sft)
        assert ctx.helper().chunk_bodies_host(data, offs, den_h, st_h, sel, budgets) == ctx.chunk_bodies(sel, budgets)
        for i in range(0, len(sel), 37):
            if 0 < budgets[i] < 0xFFFFFFFF:
                assert ctx.chunk_bodies([sel[i]], [budgets[i]])[0] == O.cap_tokens(exp.chunk_text(sel[i]), budgets[i])
        # the reference's own cases through the device path
        one = recs.index(b"one two three")
        c_one = int(exp.chunk_offsets[one])
        for case in _golden()["decorate"]:
            if not enabled or mwt < 3:
                continue
            (text, est), = T.decorate_chunks(ctx, chunks, [c_one], [case["prefix"].encode()],
                                             case["max_window_tokens"], case["newline"].encode())
            a = case["asserted"]
            assert est == a["tokens_estimate"]
            if "tokens_after" in a:
                assert _ntok(text) == a["tokens_after"]
            if "starts_with" in a:
                assert text.startswith(a["starts_with"].encode())
        ctx.close()


@pytest.mark.gpu
def test_gpu_record_bm25_texts_match_the_oracle():
    import triplets_amd as T
    rng = random.Random(9)
    pool = textgen.corpus(43, 200) + [b"", b"  \t", "　".encode(), b"Heading", b"a  b", b"x" * 300]
    records = []
    for i in range(300):
        secs = []
        for _ in range(rng.randrange(0, 5)):
            h = rng.choice([None, b"", b" ", b"Heading %d" % i, rng.choice(pool)])
            secs.append((h, rng.choice(pool)))
        records.append((b"record-%d" % i, secs))
    g = _golden()["record_bm25_text"][0]
    records.append((g["id"].encode(), [(h.encode() if h is not None else None, t.encode()) for h, t in g["sections"]]))
    for max_tokens in (0, 5, 64, 1024):
        for nl in ("\n", "\r\n"):
            got = T.record_bm25_texts(records, max_tokens, nl)
            for (rid, secs), text in zip(records, got):
                assert text == O.record_bm25_text(rid, secs, max_tokens, nl.encode()), (rid, max_tokens)
    # the BM25 query cap is the head-tokens rule with the reference's limit of 64
    q = _golden()["query_cap"][0]
    ctx = T.Context(0)
    texts = [q["text"].encode(), b" a  b ", b"", b" ".join(b"t%d" % i for i in range(64))]
    data, offs = O.pack(texts)
    import torch
    db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
    do = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(texts), len(data), T.DenoiserConfig(False),
                                T.ChunkingStrategy())
    t, o, n = ctx.head_tokens_text(q["limit"])
    assert ctx._texts_to_host(t, o, n, len(texts)) == [O.cap_tokens(x, q["limit"]) for x in texts]
    assert ctx.head_tokens_text_host(data, offs, q["limit"]) == [O.cap_tokens(x, q["limit"]) for x in texts]
    ctx.close()
