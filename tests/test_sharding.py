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


@pytest.mark.gpu
def test_in_process_sharded_host_api():
    """preprocess_sharded / ShardedPreprocessor: one thread and one context per listed device, all at once.  A box
    with one GPU lists it four times: four contexts and four host threads run their shards concurrently."""
    recs = textgen.corpus(23, 2500)
    recs.append(("x1 " * 9000 + "\n" + "12 34 ab 56 " * 3000).encode())   # one record far longer than a shard
    data, offs = O.pack(recs)
    from test_normalize import oracle_normalize
    den, st = T.DenoiserConfig(True, 0.35, True), T.ChunkingStrategy(40, [8, 3], 0.35, 10)
    exp = O.batch(data, offs, O.make_config(True, 0.35, True, 40, (8, 3), 10))
    assert T.visible_devices() >= 1
    got = T.preprocess_sharded(data, offs, [0] * 4, den, st)
    assert_same(exp, got, recs, "preprocess_sharded over [0]*4")
    with T.ShardedPreprocessor([0, 0, 0]) as sp:
        for seed in (24, 25):   # the contexts are reused from batch to batch
            r2 = textgen.corpus(seed, 700)
            d2, o2 = O.pack(r2)
            assert_same(O.batch(d2, o2, O.make_config(True, 0.35, True, 40, (8, 3), 10)),
                        sp.preprocess_batch(d2, o2, den, st), r2, "ShardedPreprocessor seed %d" % seed)
        nrm = sp.normalize_batch(data, offs)
        assert nrm.n_records == len(recs)
        for r in (0, 17, len(recs) - 1):
            assert nrm.record_text(r) == oracle_normalize(recs[r])
    # more shards than records, and an empty batch
    few = recs[:2]
    d3, o3 = O.pack(few)
    assert_same(O.batch(d3, o3, O.make_config(True, 0.35, True, 40, (8, 3), 10)),
                T.preprocess_sharded(d3, o3, [0] * 5, den, st), few, "five shards, two records")
    with pytest.raises(T.TripletsAmdError):
        T.preprocess_sharded(data, offs, [T.visible_devices() + 7], den, st)   # no such GPU: no silent fallback


# ---- bench.py's own launcher (python bench.py --gpus N without torch.distributed.run) ---------------------
def test_bench_launcher_starts_one_rank_per_gpu(tmp_path):
    """bench.launch_ranks starts N fresh processes with the environment torch.distributed.run would give them:
    here two CPU ranks (gloo) that each run the oracle over their own record range and reassemble the batch."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    out = tmp_path / "rank0.txt"
    rc = bench.launch_ranks(2, [sys.executable, os.path.join(HERE, "launch_child.py"), str(out)], timeout=180)
    assert rc == 0
    assert out.read_text().strip() == "ok world=2"
    # a failing rank fails the launch (and the other rank is stopped, not left waiting)
    rc = bench.launch_ranks(2, [sys.executable, os.path.join(HERE, "launch_child.py"), str(out), "fail1"], timeout=180)
    assert rc != 0


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` where fewer than two GPUs are visible must fail with a clear message instead of
    reporting a smaller run."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two or more GPUs visible")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRP_BENCH_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "--gpus 2" in p.stderr and "visible" in p.stderr
    assert "n_gpus" not in p.stdout
