"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle (bit-exact) on
the reference's golden vectors, adversarial fuzz corpora and edge cases."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import textgen
from parity import assert_same

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def T():
    import triplets_amd
    return triplets_amd


@pytest.fixture(scope="module")
def ctx(T):
    c = T.Context(0)
    yield c
    c.close()


def _load(name):
    with open(os.path.join(G, name), encoding="utf-8") as f:
        return json.load(f)


def _run(T, ctx, recs, cfg_tuple):
    enabled, ratio, md, mwt, ov, sft = cfg_tuple
    data, offs = O.pack(recs)
    exp = O.batch(data, offs, O.make_config(enabled, ratio, md, mwt, tuple(ov), sft))
    got = ctx.preprocess_batch(data, offs, T.DenoiserConfig(enabled, ratio, md),
                               T.ChunkingStrategy(max_window_tokens=mwt, overlap_tokens=list(ov),
                                                  summary_fallback_tokens=sft))
    return exp, got


def test_golden_denoiser_vectors(T, ctx):
    den = _load("denoiser.json")
    for case in den["exact"]:
        cfg = case["config"]
        out = T.denoise_text(case["input"], T.DenoiserConfig(cfg["enabled"], cfg["max_digit_ratio"],
                                                             cfg["strip_markdown"]), ctx)
        assert out == case["expected"], case["ref"]
    for case in den["property"]:
        res = T.DenoiserPreprocessor(T.DenoiserConfig(True, 0.35, True)).process(case["input"], ctx)
        assert res is not None
        for s in case["contains"]:
            assert s in res, case["ref"]
        for s in case["not_contains"]:
            assert s not in res, case["ref"]
        if case["line_count"] is not None:
            assert len(res.split("\n")) == case["line_count"], case["ref"]


def test_golden_tokenizer_vectors(T, ctx):
    tok = _load("tokenizer.json")
    wt = T.WhitespaceTokenizer()
    for case in tok["tokenize"]:
        assert wt.tokenize(case["input"], ctx) == case["tokens"], case["ref"]
        assert wt.token_count(case["input"], ctx) == len(case["tokens"]), case["ref"]


def test_golden_chunker_vectors(T, ctx):
    for case in _load("chunker.json")["cases"]:
        st = T.ChunkingStrategy(max_window_tokens=case["max_window_tokens"],
                                overlap_tokens=list(case["overlaps"]),
                                summary_fallback_tokens=case["summary_fallback_tokens"])
        if case["denoise"]:
            d = case["denoise"]
            st.register_preprocessor(T.DenoiserPreprocessor(
                T.DenoiserConfig(d["enabled"], d["max_digit_ratio"], d["strip_markdown"])))
        chunks = T.SlidingWindowChunker().materialize(st, "r1", 0, case["text"], ctx)
        views = [[0 if c.view.kind == "Window" else 1, c.view.index, c.view.overlap, c.view.span]
                 for c in chunks]
        assert views == [v[:4] for v in case["views"]], case["ref"]
        assert [c.tokens_estimate for c in chunks] == [v[5] - v[4] for v in case["views"]], case["ref"]
        if "texts" in case:
            assert [c.text for c in chunks] == case["texts"], case["ref"]
        a = case["asserted"]
        if "window_texts" in a:
            assert [c.text for c in chunks if c.view.kind == "Window"] == a["window_texts"]


def test_example_dataset_config_a(T, ctx):
    """multi_source_demo configuration: default chunking, no denoiser, the twelve example files."""
    import ctypes as C
    L = O.lib()
    sections = []
    for name, raw in sorted(_load("example_dataset.json")["files"].items()):
        b = raw.encode()
        buf = C.create_string_buffer(len(b) + 1)
        n = L.orc_normalize_inline_whitespace(b, len(b), buf)
        norm = buf.raw[:n]
        sections += [norm.split(b".")[0].strip(), norm]
    exp, got = _run(T, ctx, sections, (False, 0.35, True, 1024, (64,), 512))
    assert_same(exp, got, sections, "config A")
    assert got.n_chunks == len(sections)
    for r, text in enumerate(sections):
        assert got.chunk_text(r) == text


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_enabled(T, ctx, seed):
    recs = textgen.corpus(seed, 400)
    cfg = (True, [0.35, 0.2, 0.5, 0.0, 1.0][seed % 5], seed % 3 != 0, [1024, 16, 5, 0, 1][seed % 5],
           [(64,), (2, 3), (0,), (), (7, 1, 100)][seed % 5], [512, 4, 0, 3, 1][(seed // 2) % 5])
    exp, got = _run(T, ctx, recs, cfg)
    assert_same(exp, got, recs, "fuzz%d" % seed)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_disabled(T, ctx, seed):
    recs = textgen.corpus(100 + seed, 400)
    exp, got = _run(T, ctx, recs, (False, 0.35, True, [1024, 7][seed % 2], (64, 3), 5))
    assert_same(exp, got, recs, "dis%d" % seed)


def test_edge_batches(T, ctx):
    cfgs = [(True, 0.35, True, 4, (1,), 2), (False, 0.35, True, 4, (1,), 2)]
    cases = ([], [b""], [b"", b"", b""], [b"\n"], [b"a"], [b"a", b"", b"b\r", b"\nc"],
             [b"x" * 64, b"y" * 64], [b"word " * 100 + b"\r", b"\n42 42 42 a"], [b"|" * 200],
             [b" " * 130 + b"a" + b" " * 130], [b"| a |" + b" " * 70 + b"| b |"],
             ["a　".encode() * 40], [("1 " * 40 + "w ").encode() * 5],
             [b"tok" * 30000],                                   # one token across several tiles
             [b"a " * 9000, b"b" * 16384, b"", b"c d"],          # records ending on tile edges
             [b"x" * 16383 + b"\r", b"\ny"], [b"x" * 16383 + b"\r\ny"],
             [("| " + "cell " * 4000 + "|").encode()],           # table row across tiles
             [("9 " * 9000 + "word " + "7 " * 9000).encode()])   # digit-heavy line across tiles
    for cfg in cfgs:
        for recs in cases:
            exp, got = _run(T, ctx, recs, cfg)
            assert_same(exp, got, recs, repr(recs)[:60])


def test_many_digit_heavy_lines_grow_the_work_list(T, ctx):
    """more gate-3 lines than the optimistic work list holds: the batch is re-run with the exact size"""
    # lines of 71 tokens do not fit the in-tile solver (64 tokens) and are short enough to outnumber the
    # list's first capacity of one entry per 256 bytes
    recs = [b"12 ab 34\n" * 20000, (b"1 " * 70 + b"a\n") * 20000, b"7 x 9 9\n| 5 | y | 77 |\n" * 15000,
            b"plain words only\n" * 10]
    exp, got = _run(T, ctx, recs, (True, 0.35, True, 64, (8,), 16))
    assert_same(exp, got, recs, "g3list")
    exp, got = _run(T, ctx, recs, (True, 0.35, True, 64, (8,), 16))   # second call: buffer already large
    assert_same(exp, got, recs, "g3list2")


def test_extreme_record_counts_and_line_lengths(T, ctx):
    # two million empty records around a few real ones
    recs = [b""] * 1_000_000 + [b"alpha 1 2 3\nbeta"] + [b""] * 1_000_000 + [b"| x | 1 |", b""]
    exp, got = _run(T, ctx, recs, (True, 0.35, True, 4, (1,), 2))
    assert_same(exp, got, None, "empty records")
    exp, got = _run(T, ctx, recs, (False, 0.35, True, 4, (1,), 2))
    assert_same(exp, got, None, "empty records, pass-through")
    # one digit-heavy line of 300 KB (serial solver, 19 tiles), a kept and a dropped line of 3 MB, a long
    # table row; the higher ratios make the expansion run for tens of waves (more than one window of the
    # serial solver's distance histogram)
    recs = [b"word " + b"12345 " * 50_000 + b"tail", b"letters " * 400_000, b"0123456789 " * 300_000,
            (b"| " + b"9 | " * 20_000 + b"z |"), b"ab " * 10 + b"1 " * 3000 + b"cd " * 10]
    for ratio in (0.35, 0.9, 0.97):
        exp, got = _run(T, ctx, recs, (True, ratio, True, 1024, (64,), 512))
        assert_same(exp, got, None, "long lines ratio %s" % ratio)


def test_long_records_and_multiple_overlaps(T, ctx):
    import random
    rng = random.Random(7)
    recs = [textgen.long_line_record(rng, 40000), b"short one", textgen.long_line_record(rng, 5000),
            b"\n".join(textgen.record(rng) for _ in range(800))]
    for cfg in [(True, 0.35, True, 1024, (64, 256), 512), (False, 0.35, True, 100, (99, 0), 7)]:
        exp, got = _run(T, ctx, recs, cfg)
        assert_same(exp, got, recs, "long")


def test_chunk_text_materialisation(T, ctx):
    recs = textgen.corpus(5, 200)
    data, offs = O.pack(recs)
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy(12, [3, 5], 0.35, 4)
    import torch
    db = torch.from_numpy(data).cuda()
    do = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    dres = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st)
    host = dres.to_host()
    exp = O.batch(data, offs, O.make_config(True, 0.35, True, 12, (3, 5), 4))
    assert_same(exp, host, recs, "device")
    tptr, optr, nbytes = ctx.materialize_chunk_text()
    toff = np.empty(host.n_chunks + 1, dtype=np.uint64)
    text = np.empty(nbytes, dtype=np.uint8)
    L = T.load_library()
    assert L.trp_copy_to_host(ctx._h, toff.ctypes.data, optr, toff.nbytes) == 0
    if nbytes:
        assert L.trp_copy_to_host(ctx._h, text.ctypes.data, tptr, text.nbytes) == 0
    for c in range(host.n_chunks):
        assert text[int(toff[c]):int(toff[c + 1])].tobytes() == exp.chunk_text(c), c


def test_submit_and_wait_on_two_contexts(T, ctx):
    """Two batches in flight on two contexts (the halves of the blocking call) give the blocking results."""
    import torch
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy(16, [3], 0.35, 4)
    other = T.Context(0)
    try:
        with pytest.raises(T.TripletsAmdError):
            ctx.wait_batch()                      # nothing in flight
        batches = []
        for seed in (21, 22, 23, 24):
            recs = textgen.corpus(seed, 400) + [textgen.digit_heavy_long_line(__import__("random").Random(seed), 90)]
            data, offs = O.pack(recs)
            batches.append((recs, data, offs, torch.from_numpy(data).cuda(),
                            torch.from_numpy(offs.astype(np.int64)).cuda()))
        torch.cuda.synchronize()
        cs = [ctx, other]
        inflight = [None, None]
        got = {}
        for i, (recs, data, offs, db, do) in enumerate(batches):
            k = i % 2
            if inflight[k] is not None:
                got[inflight[k]] = cs[k].wait_batch().to_host()
            cs[k].submit_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st)
            inflight[k] = i
            if i == 0:
                with pytest.raises(T.TripletsAmdError):   # one batch per context
                    cs[k].submit_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st)
        for k in (0, 1):
            got[inflight[k]] = cs[k].wait_batch().to_host()
        for i, (recs, data, offs, db, do) in enumerate(batches):
            exp = O.batch(data, offs, O.make_config(True, 0.35, True, 16, (3,), 4))
            assert_same(exp, got[i], recs, "in flight %d" % i)
    finally:
        other.close()


@pytest.mark.parametrize("seed", range(4))
def test_scripts_and_rule_boundaries(T, ctx, seed):
    """Many scripts and the characters at the edges of the classifier's range rules, all modes."""
    recs = textgen.script_corpus(800 + seed, 400)
    for cfg in ((True, 0.35, True, 12, (3,), 4), (True, 0.8, False, 1024, (64,), 512), (False, 0.35, True, 12, (3,), 4)):
        exp, got = _run(T, ctx, recs, cfg)
        assert_same(exp, got, recs, "scripts%d" % seed)


def test_tile_carries_from_windows_and_from_summaries(T, ctx):
    """Batches whose tile carries are settled by the windows next to each tile, batches that need the
    per-tile summaries (long letterless stretches, lines longer than the windows reach), and the same
    batches with the summary path forced: all equal to the oracle."""
    plain = textgen.corpus(31, 600)
    letterless = ("12 34 56 78\n" * 20000 + "word at last\n" + "9 9 9\n" * 3000).encode()       # ~300 KB without a letter
    long_line = (" ".join("w%d" % i for i in range(12000)) + "\nnext line\n").encode()          # one 70 KB line
    long_digits = ("7 " * 40000 + "x\n").encode()                                               # digit-heavy 80 KB line
    spaces = b" " * 150000 + b"x" + b" " * 150000 + b"y"
    batches = [plain, plain[:50] + [letterless] + plain[50:100], [long_line] + plain[:30] + [long_digits],
               [spaces, b"", spaces], plain[:10] + [letterless, long_line, spaces] + plain[10:20]]
    os.environ["TRIPLETS_AMD_FORCE_SUMMARY"] = "1"
    try:
        forced = T.Context(0)
    finally:
        del os.environ["TRIPLETS_AMD_FORCE_SUMMARY"]
    try:
        for i, recs in enumerate(batches):
            for cfg in ((True, 0.35, True, 1024, (64,), 512), (True, 0.9, False, 16, (3,), 4)):
                exp, got = _run(T, ctx, recs, cfg)
                assert_same(exp, got, recs, "windows %d" % i)
                exp2, got2 = _run(T, forced, recs, cfg)
                assert_same(exp2, got2, recs, "summaries %d" % i)
            data, offs = O.pack(recs)
            a, b = ctx.normalize_batch(data, offs), forced.normalize_batch(data, offs)
            assert a.denoised.tobytes() == b.denoised.tobytes() and np.array_equal(a.tok_offsets, b.tok_offsets)
    finally:
        forced.close()


def test_one_byte_records_fill_the_token_table(T, ctx):
    """A batch of one-byte records holds as many tokens as bytes (a record start begins a token): the
    token arrays must be sized for that, in the denoising, pass-through and normalize modes; the mixed
    batch also takes more tokens per 64-byte block than the staged token table of a tile holds."""
    import test_normalize as TN
    for recs in ([b"a"] * 200_000, [b"a", b"b c", b"d", b"1", b"e f", b"", b"gh"] * 40_000,
                 [b"x" if i % 3 else b"y z" for i in range(150_001)]):
        for cfg in ((True, 0.35, True, 4, (1,), 2), (False, 0.35, True, 1024, (64,), 512)):
            exp, got = _run(T, ctx, recs, cfg)
            assert_same(exp, got, None, "one-byte records, denoiser %s" % cfg[0])
        norm, exp = TN.expected(recs)
        data, offs = O.pack(recs)
        assert_same(exp, ctx.normalize_batch(data, offs), None, "one-byte records, normalize")


def test_token_density_around_the_staged_table_of_a_sub_tile(T, ctx):
    """k_emit stages the tokens of a 2 KiB sub-tile in LDS up to a fixed number and stores the rest directly:
    token densities below, at and above that number, tokens that straddle every sub-tile edge (word lengths
    that do not divide 2048), a token longer than several sub-tiles between dense stretches, many records per
    sub-tile and records that end exactly on sub-tile edges; denoiser on and off, and normalize mode."""
    import test_normalize as TN
    dense = [b"a b " * 5000, b"ab cd " * 4000, b"abc def " * 3000, b"abcd efg " * 2500, b"abcdefg hi " * 2000]
    mixed = [(b"w " * 1500 + b"longtoken" * 1500 + b" x y z " + b"q" * 4096 + b" " + b"r s " * 900)]
    edges = [b"ab " * 1365 + b"c", b"d" * 4096, b"e f " * 1024, b"", b"g" * 4095 + b" ", b"h i"] * 3
    crowded = [b"k l"] * 5000 + [b"m" * 70] * 300 + [b"n o p q\n12 34 56 ab\n"] * 2000
    # tokens of hundreds of KB (their start is looked up from the sub-tile where they end, 64 blocks at a time):
    # starting at a record start, after a short word, ending at a record end, followed by more text
    blobs = [b"x" * 300000 + b" y", b"a " + b"z" * 70001, b"w" * 5000, b"pre\n" + b"Q" * 200003 + b"\nafter it", b"v" * 4097]
    for recs in (dense, mixed, edges, crowded, blobs, dense + mixed + blobs + edges + crowded):
        for cfg in ((True, 0.35, True, 16, (4, 2), 8), (False, 0.35, True, 1024, (64,), 512)):
            exp, got = _run(T, ctx, recs, cfg)
            assert_same(exp, got, recs, "token density, denoiser %s" % cfg[0])
        norm, exp = TN.expected(recs)
        data, offs = O.pack(recs)
        assert_same(exp, ctx.normalize_batch(data, offs), recs, "token density, normalize")


def test_line_lengths_around_the_short_line_carries(T, ctx):
    """k_plan takes the line carries of a wavefront from its neighbouring lanes when every block has a line start
    within three blocks on either side, and from prefix sums otherwise: line lengths below, at and above that
    reach (in blocks of 64 bytes), wavefronts (4 KiB) that mix both, table rows and digit-heavy lines among them,
    lines without a letter, records that start in the middle of such stretches."""
    import random
    rng = random.Random(5)

    def line(n, kind):
        if kind == 0:
            words = "lorem ipsum dolor sit amet consectetur adipiscing elit sed do".split()
        elif kind == 1:
            words = "| cell | 12 | other cell | 3.5 |".split(" ")
        elif kind == 2:
            words = "2019 1,234 56.7 total 8 9 10".split()
        else:
            words = "12 34 -- 56 ++ 78".split()
        out = []
        size = 0
        while size < n:
            w = rng.choice(words)
            out.append(w)
            size += len(w) + 1
        return " ".join(out)[:max(n, 1)]

    recs = []
    for lengths in ((60, 100), (180, 200), (190, 260), (250, 330), (60, 700), (1, 64), (120, 130)):
        body = []
        for i in range(400):
            body.append(line(rng.randint(*lengths), rng.choice((0, 0, 0, 1, 2, 3))))
            if rng.random() < 0.02:
                recs.append("\n".join(body).encode())
                body = []
        recs.append("\n".join(body).encode())
    # every block exactly one line of 63 bytes + terminator, then lines of exactly 192 and 256 bytes
    recs.append(("\n".join("x" * 63 for _ in range(300))).encode())
    recs.append(("\n".join(("y z " * 48)[:191] for _ in range(100))).encode())
    recs.append(("\n".join(("y z " * 64)[:255] for _ in range(100))).encode())
    for cfg in ((True, 0.35, True, 16, (4, 2), 8), (True, 0.9, False, 1024, (64,), 512)):
        exp, got = _run(T, ctx, recs, cfg)
        assert_same(exp, got, recs, "line lengths, ratio %s" % cfg[1])


def test_characters_across_block_edges(T, ctx):
    """Multi-byte characters placed so that they straddle the 64-byte blocks and the 4 KiB wavefront edges of the
    classifier (which settles them with the first bytes of the next block): letters and whitespace characters of
    two, three and four bytes at every offset against the block grid, in the scripts the range rules cover and in
    scripts that take the per-character path; denoiser on and off."""
    chars = ["\u00e9", "\u00a0", "\u0085", "\u03a9", "\u0416", "\u4e2d", "\u3000", "\u2028", "\u2003", "\u3042",
             "\U0001f600", "\U00020000", "\u10d0", "\u1680", "\u0e01", "\u0915", "\uac00", "\u205f", "\ufeff"]
    recs = []
    for shift in range(0, 9):
        parts = []
        for i, ch in enumerate(chars * 12):
            # the character lands near the end of a block: filler brings the running length to 64 k - (0..3)
            cur = sum(len(x.encode()) for x in parts) + shift
            pad = (-cur - (i % 4)) % 64
            parts.append("a" * max(pad - 1, 0) + (" " if pad else "") + ch + ch + ("\n" if i % 7 == 0 else " "))
        recs.append(("b" * shift + "".join(parts)).encode())
    recs.append(("\u4e2d" * 3000).encode())
    recs.append(("\u3000" * 700 + "x" + "\u00a0" * 500).encode())
    for cfg in ((True, 0.35, True, 16, (4,), 8), (False, 0.35, True, 1024, (64,), 512)):
        exp, got = _run(T, ctx, recs, cfg)
        assert_same(exp, got, recs, "block edges, denoiser %s" % cfg[0])


def test_inputs_produced_on_the_callers_stream_need_no_sync(T):
    """A context created on the caller's stream (here torch's current stream) is ordered after the
    asynchronous copies that produce its inputs; a context with a private stream gets the same ordering
    from an event.  No device-wide synchronisation anywhere before the calls."""
    import torch
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy(64, [8], 0.35, 16)
    ocfg = O.make_config(True, 0.35, True, 64, (8,), 16)
    recs = textgen.corpus(77, 3000) * 6
    data, offs = O.pack(recs)
    exp = O.batch(data, offs, ocfg, n_threads=8)
    pin_d = torch.from_numpy(data).pin_memory()
    pin_o = torch.from_numpy(offs.astype(np.int64)).pin_memory()
    side = torch.cuda.Stream()
    on_torch = T.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    on_side = T.Context(0, stream=side.cuda_stream)
    private = T.Context(0)
    try:
        for rep in range(3):
            # (1) producer = torch's current stream, context on that stream
            filler = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
            filler.fill_(rep)                                   # keeps the stream busy ahead of the copies
            db = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
            db[:len(data)].copy_(pin_d, non_blocking=True)
            do = pin_o.to("cuda", non_blocking=True)
            got = on_torch.preprocess_batch_device(db.data_ptr(), do.data_ptr(), len(recs), len(data), den, st).to_host()
            assert_same(exp, got, None, "context on torch's stream, rep %d" % rep)
            # (2) producer = a side stream, context on the side stream
            with torch.cuda.stream(side):
                filler2 = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
                filler2.fill_(rep)
                db2 = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
                db2[:len(data)].copy_(pin_d, non_blocking=True)
                do2 = pin_o.to("cuda", non_blocking=True)
            got = on_side.preprocess_batch_device(db2.data_ptr(), do2.data_ptr(), len(recs), len(data), den, st).to_host()
            assert_same(exp, got, None, "context on a side stream, rep %d" % rep)
            # (3) producer = the side stream, private-stream context ordered by an event
            with torch.cuda.stream(side):
                filler2.fill_(rep + 1)
                db3 = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
                db3[:len(data)].copy_(pin_d, non_blocking=True)
                do3 = pin_o.to("cuda", non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            private.wait_event(ev.cuda_event)
            got = private.preprocess_batch_device(db3.data_ptr(), do3.data_ptr(), len(recs), len(data), den, st).to_host()
            assert_same(exp, got, None, "private stream after an event, rep %d" % rep)
            side.synchronize()
            torch.cuda.current_stream().synchronize()
    finally:
        on_torch.close()
        on_side.close()
        private.close()


def test_long_digit_heavy_lines_by_length(T, ctx):
    """Digit-heavy lines are solved by one lane up to 256 bytes, by a wavefront up to 4 KiB and up to 16 KiB, and
    walked by one lane beyond: lengths on both sides of every limit, at several offsets against the 64-byte
    blocks and across tile edges, plain and table rows, first line of a record and later lines, few and many
    expansion rounds (letters at the start, in the middle, spread out), CRLF and end-of-record terminators."""
    import random
    rng = random.Random(11)

    def digits_line(n, kind):
        parts = []
        size = 0
        i = 0
        while size < n:
            if kind == 0:
                w = "total" if i == 0 else str(rng.randint(0, 99999))
            elif kind == 1:
                w = "mid" if abs(size - n // 2) < 6 else str(rng.randint(0, 9999))
            elif kind == 2:
                w = "ab" if i % 9 == 0 else str(rng.randint(0, 999999))
            else:
                w = "| x%d |" % i if i % 11 == 0 else "| %d" % rng.randint(0, 99999)
            parts.append(w)
            size += len(w) + 1
            i += 1
        return " ".join(parts)[:n].rstrip() or "7 a"

    recs = []
    for n in (250, 256, 257, 300, 1000, 4000, 4090, 4096, 4100, 4200, 8000, 16000, 16380, 16390, 17000, 40000):
        for kind in range(4):
            pad = "x" * rng.randint(0, 70)
            body = ["plain first line " + pad, digits_line(n, kind), "after the line", digits_line(n // 2 + 130, (kind + 1) % 4)]
            recs.append("\n".join(body).encode())
            recs.append((digits_line(n, kind) + "\r\nnext one").encode())          # first line of its record, CRLF
            recs.append(("intro\n" + digits_line(n, kind)).encode())               # ends with its record
    rng.shuffle(recs)
    for cfg in ((True, 0.35, True, 32, (4,), 8), (True, 0.6, False, 1024, (64,), 512)):
        exp, got = _run(T, ctx, recs, cfg)
        assert_same(exp, got, recs, "long digit-heavy lines, ratio %s" % cfg[1])


def test_device_entry_point_rejects_offsets_that_break_the_contract(T, ctx):
    """The device entry point cannot look at the offsets on the host: the first kernel checks them (start at 0,
    non-decreasing, end at n_bytes) and the call fails with an argument error instead of running away; the context
    is usable afterwards, with and without the denoiser."""
    import torch
    recs = textgen.corpus(77, 400)
    data, offs = O.pack(recs)
    R, N = len(recs), len(data)
    db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
    good = offs.astype(np.int64)
    bad_sets = []
    a = good.copy(); a[R // 2], a[R // 2 + 1] = good[R // 2 + 1], good[R // 2]; bad_sets.append(a)     # one pair swapped
    a = good.copy(); a[0] = 1; bad_sets.append(a)                                                        # does not start at 0
    a = good.copy(); a[R] = N - 1; bad_sets.append(a)                                                    # does not end at N
    a = good[::-1].copy(); bad_sets.append(a)                                                            # decreasing throughout
    den_on, den_off = T.DenoiserConfig(True, 0.35, True), T.DenoiserConfig(False, 0.35, True)
    st = T.ChunkingStrategy(max_window_tokens=16, overlap_tokens=[3], summary_fallback_tokens=4)
    torch.cuda.synchronize()
    for den in (den_on, den_off):
        for bad in bad_sets:
            do = torch.from_numpy(bad).cuda()
            torch.cuda.synchronize()
            with pytest.raises(T.TripletsAmdError):
                ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), R, N, den, st)
        do = torch.from_numpy(good).cuda()
        torch.cuda.synchronize()
        got = ctx.preprocess_batch_device(db.data_ptr(), do.data_ptr(), R, N, den, st).to_host()
        exp = O.batch(data, offs, O.make_config(den.enabled, 0.35, True, 16, (3,), 4))
        assert_same(exp, got, recs, "after rejected batches, denoiser %s" % den.enabled)
    # a digit-heavy line of more than 16 KiB (the streamed gate-3 solver, which looks for the end of its record in
    # the offsets) in a batch whose last offset lies before the line: rejected, nothing read beyond offsets[R]
    long_line = ("12 34 ab 56 " * 3400).encode()
    dl = torch.from_numpy(np.concatenate([np.frombuffer(long_line, np.uint8), np.zeros(64, np.uint8)])).cuda()
    for bad in ([0, 8], [0, 0], [0, 3, 8]):
        do = torch.tensor(bad, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(T.TripletsAmdError):
            ctx.preprocess_batch_device(dl.data_ptr(), do.data_ptr(), len(bad) - 1, len(long_line), den_on, st)
    do = torch.tensor([0, len(long_line)], dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    got = ctx.preprocess_batch_device(dl.data_ptr(), do.data_ptr(), 1, len(long_line), den_on, st).to_host()
    d1, o1 = O.pack([long_line])
    assert_same(O.batch(d1, o1, O.make_config(True, 0.35, True, 16, (3,), 4)), got, [long_line], "long line after rejects")
