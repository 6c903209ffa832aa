"""Shared comparison of two packed-batch results (oracle vs simulator / oracle vs GPU)."""
import numpy as np

FIELDS = ("denoised_offsets", "dropped", "tok_offsets", "tok_start", "tok_len", "chunk_offsets",
          "chunks", "denoised")


def first_diff(a, b):
    n = min(len(a), len(b))
    if n:
        d = np.nonzero(a[:n] != b[:n])[0]
        if len(d):
            return int(d[0])
    return n if len(a) != len(b) else -1


def assert_same(exp, got, records=None, what=""):
    """bit-exact equality of every output array; on failure names the first differing record."""
    for k in FIELDS:
        a, b = getattr(exp, k), getattr(got, k)
        i = first_diff(a, b)
        if i < 0:
            continue
        msg = "%s: field %s differs at index %d (len exp %d, got %d)" % (what, k, i, len(a), len(b))
        if k in ("denoised_offsets", "dropped", "tok_offsets", "chunk_offsets"):
            r = max(0, i - 1) if k != "dropped" else i
            msg += "\n record %d" % r
            if records is not None and r < len(records):
                msg += " input=%r" % records[r][:400]
            if r < exp.n_records:
                msg += "\n expected text=%r" % exp.record_text(r)[:400]
                msg += "\n got      text=%r" % got.denoised[int(got.denoised_offsets[r]):int(
                    got.denoised_offsets[min(r + 1, got.n_records)])].tobytes()[:400]
        raise AssertionError(msg)
