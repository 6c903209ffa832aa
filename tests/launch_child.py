"""Child process of test_bench_launcher_starts_one_rank_per_gpu: one rank of a CPU (gloo) rehearsal of the
multi-rank bench.  Reads RANK / WORLD_SIZE / MASTER_* from the environment the launcher made, runs the oracle over
its own record range and lets rank 0 check the reassembled batch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-triplets_amd"))


def main():
    out_path = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if mode == "fail1" and rank == 1:
        sys.exit(3)
    import torch.distributed as dist
    import oracle_lib as O
    import textgen
    import triplets_amd as T
    from parity import assert_same
    dist.init_process_group("gloo")
    try:
        recs = textgen.corpus(5, 300)
        data, offs = O.pack(recs)
        cfg = O.make_config(True, 0.35, True, 32, (4,), 8)
        r0, r1 = T.shard_ranges(offs, world)[rank]
        d, o = T.shard_slice(data, offs, r0, r1)
        whole = T.gather_shards(T.BatchResult(**O.batch(d, o, cfg).__dict__))
        if rank == 0:
            assert_same(O.batch(data, offs, cfg), whole, recs, "launcher world=%d" % world)
            with open(out_path, "w") as f:
                f.write("ok world=%d\n" % world)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
