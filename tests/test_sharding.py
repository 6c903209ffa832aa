"""Multi-GPU path: byte-balanced record-range shards, one rank per shard, no data-path collective.

The CPU tests run the shard / gather / concatenate plumbing under torch.distributed (gloo, world size 2)
with the oracle standing in for the per-shard GPU call; the GPU test runs real shards through the
library one after the other and compares with the unsharded call and the oracle.
"""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as O
import textgen
import triplets_amd as T
from parity import assert_same

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_ranges_cover_and_balance():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 5000, size=1000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for g in (1, 2, 3, 4, 8, 16):
        rr = T.shard_ranges(offs, g)
        assert len(rr) == g and rr[0][0] == 0 and rr[-1][1] == 1000
        assert all(rr[i][1] == rr[i + 1][0] for i in range(g - 1))
        sizes = [int(offs[b]) - int(offs[a]) for a, b in rr]
        assert max(sizes) - min(sizes) <= 2 * 5000
    # degenerate inputs
    assert T.shard_ranges(np.zeros(1, np.uint64), 4) == [(0, 0)] * 4
    assert T.shard_ranges(np.array([0, 10], np.uint64), 2) in ([(0, 0), (0, 1)], [(0, 1), (1, 1)])


def test_concat_of_shard_results_equals_whole_batch():
    recs = textgen.corpus(11, 400)
    data, offs = O.pack(recs)
    cfg = O.make_config(True, 0.35, True, 48, (8, 16), 12)
    whole = O.batch(data, offs, cfg)
    for g in (2, 3, 7):
        parts = []
        for r0, r1 in T.shard_ranges(offs, g):
            d, o = T.shard_slice(data, offs, r0, r1)
            parts.append(O.batch(d, o, cfg))
        got = T.concat_results([T.BatchResult(**p.__dict__) for p in parts])
        assert_same(whole, got, recs, "concat g=%d" % g)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-triplets_amd"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs = textgen.corpus(5, 300)           # every rank derives the same corpus and its own range
        data, offs = O.pack(recs)
        cfg = O.make_config(True, 0.35, True, 32, (4,), 8)
        r0, r1 = T.shard_ranges(offs, world)[rank]
        d, o = T.shard_slice(data, offs, r0, r1)
        local = O.batch(d, o, cfg)
        whole = T.gather_shards(T.BatchResult(**local.__dict__))
        if rank == 0:
            exp = O.batch(data, offs, cfg)
            assert_same(exp, whole, recs, "gloo world=%d" % world)
            q.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_run_reassembles_the_batch():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


@pytest.mark.gpu
def test_gpu_shards_equal_unsharded_call():
    recs = textgen.corpus(21, 1500)
    data, offs = O.pack(recs)
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy(40, [8], 0.35, 10)
    ctx = T.Context(0)
    whole = ctx.preprocess_batch(data, offs, den, st)
    exp = O.batch(data, offs, O.make_config(True, 0.35, True, 40, (8,), 10))
    assert_same(exp, whole, recs, "unsharded")
    for g in (2, 8):
        parts = []
        for r0, r1 in T.shard_ranges(offs, g):
            d, o = T.shard_slice(data, offs, r0, r1)
            parts.append(ctx.preprocess_batch(d, o, den, st))
        assert_same(exp, T.concat_results(parts), recs, "gpu shards g=%d" % g)
    ctx.close()
