"""A strategy with several registered denoisers: the texts are folded through them in registration order and a
section is dropped as soon as one returns None (src/chunking/sliding_window.rs:20-34, src/config.rs:105-113).
Expected values: the oracle applied stage by stage on the host, each stage on the text the one before returned."""
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


def oracle_fold(recs, denoisers, chunking):
    """denoisers: (enabled, ratio, strip_markdown) tuples; chunking: (window, overlaps, fallback)."""
    texts = list(recs)
    dead = [False] * len(recs)
    for en, ratio, md in denoisers:
        if not en:
            continue                                     # returns its text unchanged, even an empty one
        data, offs = O.pack(texts)
        r = O.batch(data, offs, O.make_config(True, ratio, md, 1024, (), 0))
        for i in range(len(texts)):
            if dead[i] or r.dropped[i]:
                dead[i] = True
                texts[i] = b""
            else:
                texts[i] = r.record_text(i)
    # the last step: tokens and chunks of the folded texts; a dropped section has no text, no tokens, no chunks
    data, offs = O.pack(texts)
    final = O.batch(data, offs, O.make_config(False, 0.35, True, *chunking))
    final.dropped = np.array([1 if d else 0 for d in dead], dtype=np.uint8)
    return final


FOLDS = [
    [(True, 0.35, True), (True, 0.1, True)],
    [(True, 0.6, False), (True, 0.35, True), (True, 0.2, True)],
    [(False, 0.35, True), (True, 0.35, True)],
    [(True, 0.35, True), (False, 0.35, True)],
    [(False, 0.35, True), (False, 0.1, False)],
    [(True, 0.35, True)],
    [],
]


def test_fold_of_denoisers_matches_the_oracle_stage_by_stage(T):
    import torch
    ctx = T.Context(0)
    chunking = (16, (4,), 8)
    st = T.ChunkingStrategy(max_window_tokens=16, overlap_tokens=[4], summary_fallback_tokens=8)
    batches = [textgen.corpus(31, 3), textgen.corpus(32, 200) + textgen.script_corpus(5, 30),
               [b"", b"12 34 56\n78 90", b"| a | 1 |\n|---|---|\n| 22 | b |", b"plain words only", b"7 8 9 x\n1 2 3 4 5 6"]]
    for recs in batches:
        data, offs = O.pack(recs)
        for fold in FOLDS:
            exp = oracle_fold(recs, fold, chunking)
            dens = [T.DenoiserConfig(*f) for f in fold]
            got = ctx.preprocess_batch_fold(data, offs, dens, st)
            assert_same(exp, got, recs, "host fold %s" % (fold,))
            db = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).cuda()
            do = torch.from_numpy(offs.astype(np.int64)).cuda()
            torch.cuda.synchronize()
            dev = ctx.preprocess_batch_device_fold(db.data_ptr(), do.data_ptr(), len(recs), len(data), dens, st)
            assert_same(exp, dev.to_host(), recs, "device fold %s" % (fold,))
    ctx.close()
