#!/usr/bin/env python3
"""Benchmark of the denoise -> tokenise -> chunk path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N=1 by default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (no WORLD_SIZE in the environment) with --gpus N > 1, the script starts the N ranks
itself, one fresh process per GPU, before anything touches a GPU, and fails loudly when fewer than N GPUs are
visible: it never reports a one-GPU number under a multi-GPU request.

A step is one pass of the whole path (all kernels of trp_preprocess_batch_device) over one packed
batch that is already resident in HBM.  Workload at every N: BASELINE.json configs[1] - 1.0e9 bytes of
synthetic ASCII English, records of 3072-5120 bytes, denoiser enabled, default chunking - per GPU
(weak scaling: records shard by range, no inter-GPU traffic in the timed region).  Rank 0 prints one
JSON line.  Before the W warm-up steps every workload runs untimed passes for 50 ms ("settle_ms"): a GPU that
sat idle while the host generated the corpus takes about 25 ms of load to reach its steady clocks.

--config E is BASELINE.json configs[4]: a 64e9-byte corpus (OCR-noisy markdown generator) cut by record
range into one shard per GPU (strong scaling); a shard is held in HBM as slices of at most 4e9 bytes,
each one batch, and a step is one pass over all slices of the shard.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

# A GPU that has been idle (here: while the host generates the corpus) needs about 25 ms of load before its
# clocks reach their steady state: the first twenty blocking calls on 1 GB run 1.44 ... 1.20 ms
# (tools/step_times.py).  Every workload therefore runs untimed passes for SETTLE_S seconds before its W warm-up
# steps, so that the W / K the caller chose measure the steady rate; the JSON line says so ("settle_ms").
SETTLE_S = 0.05


def settle(step):
    """Untimed passes until SETTLE_S seconds of GPU work have gone by (at least two)."""
    t0, n = time.perf_counter(), 0
    while n < 2 or time.perf_counter() - t0 < SETTLE_S:
        step()
        n += 1
    return n

# Algorithmic HBM bytes of each kernel per launch: what the path has to move through that kernel once, from the
# batch figures (N input bytes, D denoised bytes, T tokens, R records).  Passes the implementation chose to make
# (the N/2 bytes of plan bit planes k_plan writes and k_emit reads back) do not count; "with_plan" is the same
# with those bytes included, i.e. what the kernel really has to move as it is built.
KERNEL_BYTES = {
    "k_summary": lambda s: s["N"],
    "k_plan": lambda s: s["N"],
    "k_gate3": lambda s: s["N"] // 8,
    "k_count_tokens": lambda s: s["N"],
    "k_emit": lambda s: s["N"] + (s["D"] if s["enabled"] else 0) + 8 * s["T"] + 16 * (s["R"] + 1),
}
PLAN_BYTES = {"k_plan": lambda s: s["N"] // 2, "k_emit": lambda s: s["N"] // 2 if s["enabled"] else 0}


def path_algorithmic_bytes(s):
    """B_alg of the whole path: input read once, denoised bytes, token table, chunk descriptors, the
    four u64 offset arrays and the dropped flags."""
    return s["N"] + (s["D"] if s["enabled"] else 0) + 8 * s["T"] + 32 * s["C"] + 32 * (s["R"] + 1) + s["R"]


def workload_name(config, nbytes):
    if config == "B":
        return "config B: %.1e B synthetic ASCII English, records 3072-5120 B" % nbytes
    if config == "C":
        return "config C: %.1e B synthetic OCR-noisy markdown, multibyte UTF-8, records 2-16 KB" % nbytes
    if config == "W":
        return ("config C with wide scripts: %.1e B, half of the multibyte words Georgian / Ethiopic / Bengali / "
                "Tamil / Armenian / Khmer / Tibetan / Cherokee (per-character classifier path)" % nbytes)
    return "config D: %.1e B skewed lengths, half the bytes in 200 B and half in 1 MiB records" % nbytes


def committed_traffic(workload, n_bytes, denoiser_enabled, kernel):
    """HBM bytes per launch of `kernel` from the newest profiles/r*_traffic.json that was collected on exactly
    this workload (same generator line, byte count and denoiser setting), or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            with open(path) as f:
                tj = json.load(f)
        except Exception:
            continue
        meta = tj.get("_meta", {})
        if (meta.get("bytes_per_gpu") == n_bytes and meta.get("workload") == workload
                and meta.get("denoiser") == ("enabled" if denoiser_enabled else "disabled")):
            v = (tj.get(kernel) or {}).get("bytes")
            if v is not None:
                best = v   # file names sort by round
    return best


def kernel_profile(ctx, step, nb, reps):
    """Average duration of every kernel per launch (HIP events on the context's stream) over `reps` passes."""
    ctx.set_profiling(True)
    acc = {}

    def collect():
        for name, ms in ctx.profile():
            acc[name] = acc.get(name, 0.0) + ms

    for _ in range(reps):
        step(collect)
    ctx.set_profiling(False)
    return {k: v / (reps * nb) for k, v in acc.items()}


def roofline_of(kern, stats, nb, ms_per_step):
    """Roofline object of the dominant kernel; stats are the totals of one pass over nb batches."""
    per_launch = {k: (v // nb if k != "enabled" else v) for k, v in stats.items()}   # average batch
    dominant = max(kern, key=kern.get)
    dom_bytes = KERNEL_BYTES.get(dominant, lambda s: s["N"])(per_launch)
    with_plan = dom_bytes + PLAN_BYTES.get(dominant, lambda s: 0)(per_launch)
    achieved = dom_bytes / (kern[dominant] * 1e-3) / 1e9
    b_alg = path_algorithmic_bytes(stats)
    return {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "kernel_ms": round(kern[dominant], 4), "algorithmic_bytes": dom_bytes,
            "frac_with_plan": round(with_plan / (kern[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "path_algorithmic_bytes": b_alg,
            "path_frac": round(b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}}


def run_in_flight(T, device, batches, den, st, steps, warmup, stagger_s=0.0):
    """The same passes with one batch in flight on each of two contexts (trp_submit_batch_device /
    trp_wait_batch): the kernels of consecutive batches overlap, the way a consumer of one batch overlaps with
    the production of the next.  Every batch of every pass is waited for inside the timed region.
    stagger_s: the second context's first batch is submitted that much later than the first context's, so that
    the two run out of phase (one batch's issue-bound planning beside the other's bandwidth-bound emit); the
    delay is inside the timed region."""
    import torch
    cs = (T.Context(device), T.Context(device))

    def passes(k):
        busy, i = [False, False], 0
        for _ in range(k):
            for d_bytes, d_offs, r, n in batches:
                j = i % 2
                if busy[j]:
                    cs[j].wait_batch()
                elif i == 1 and stagger_s > 0:
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < stagger_s:
                        pass
                cs[j].submit_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n, den, st)
                busy[j] = True
                i += 1
        for j in (0, 1):
            if busy[j]:
                cs[j].wait_batch()

    settle(lambda: passes(1))
    passes(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for c in cs:
        c.close()
    N = sum(b[3] for b in batches)
    return {"value": round(N * steps / elapsed / 1e9, 2), "unit": "GB/s", "ms_per_step": round(elapsed / steps * 1e3, 4),
            "steps": steps, "calls": "submit/wait, one batch in flight on each of two contexts"
                                     + (", the second %.2f ms out of phase" % (stagger_s * 1e3) if stagger_s else "")}


def run_extra(T, ctx, name, workload, batches, den, st, steps, warmup, normalize=False, traffic_workload=None):
    """One more single-GPU workload on the same context: timed like the headline one (blocking calls,
    synchronize on both sides), then a profiled pass for the per-kernel durations.  normalize: time
    normalize_inline_whitespace (trp_normalize_batch_device) instead of the path."""
    import torch

    def step(after_batch=None):
        d = t = c = 0
        for d_bytes, d_offs, r, n in batches:
            if normalize:
                res = ctx.normalize_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n)
            else:
                res = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n, den, st)
            d, t, c = d + int(res.n_denoised), t + int(res.n_tokens), c + int(res.n_chunks)
            if after_batch is not None:
                after_batch()
        return d, t, c

    settle(step)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    N, R, nb = sum(b[3] for b in batches), sum(b[2] for b in batches), len(batches)
    ms = elapsed / steps * 1e3
    stats = {"N": N, "R": R, "D": res[0], "T": res[1], "C": res[2], "enabled": den.enabled or normalize}
    kern = kernel_profile(ctx, step, nb, 3)
    roof = roofline_of(kern, stats, nb, ms)
    # HBM bytes per launch from the committed counter runs of this very workload (traffic_workload: the name the
    # workload has when it is the main one of a run, i.e. in the profile's own bench line)
    if (nb == 1 or traffic_workload) and not normalize:
        roof["traffic"] = committed_traffic(traffic_workload or workload, N, den.enabled, roof["kernel"])
    return {"workload": workload, "value": round(N * steps / elapsed / 1e9, 2), "unit": "GB/s",
            "denoiser": "normalize_inline_whitespace" if normalize else ("enabled" if den.enabled else "disabled"),
            "ms_per_step": round(ms, 4), "steps": steps, "bytes": N, "records": R, "batches_per_step": nb,
            "overlap_tokens": list(st.overlap_tokens),
            "outputs": {"denoised_bytes": res[0], "tokens": res[1], "chunks": res[2]},
            "roofline": roof}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, env=None, timeout=None):
    """Start `argv` (a command line) as n fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set the way torch.distributed.run sets them, wait for all of them and return the largest exit
    status.  The caller must not have initialised a GPU: the children are new processes (no fork of GPU state, no
    exec from a process that holds one).  When a rank fails the others are stopped (by their own PIDs)."""
    import subprocess
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(free_port())})
    procs = []
    for r in range(n):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen(argv, env=e))
    t_end = None if timeout is None else time.time() + timeout
    worst, live = 0, list(procs)
    while live:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            worst = max(worst, abs(rc))
            if rc != 0:
                for q in live:
                    q.terminate()
        if t_end is not None and time.time() > t_end:
            for q in live:
                q.kill()
            worst = max(worst, 124)
            t_end = None
        time.sleep(0.05)
    return worst


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: check that N GPUs are there, then run the N ranks."""
    import torch
    have = torch.cuda.device_count()   # counting devices does not initialise one
    if "TRP_BENCH_DEVICE" not in os.environ and have < n:
        raise SystemExit("bench.py: --gpus %d asked for but %d GPU(s) visible; refusing to report a smaller run "
                         "as a %d-GPU one" % (n, have, n))
    rc = launch_ranks(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=None,
                    help="bytes per GPU (default 1e9); with --config E the bytes of the whole corpus (default 64e9)")
    ap.add_argument("--config", default="B", choices=["B", "C", "D", "E", "W"])
    ap.add_argument("--no-denoise", action="store_true")
    ap.add_argument("--normalize", action="store_true",
                    help="time normalize_inline_whitespace (the step before the path) instead of the path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", action="store_true",
                    help="keep one batch in flight on each of two contexts (trp_submit_batch_device / "
                         "trp_wait_batch) instead of timing blocking calls on one context")
    ap.add_argument("--host-path", action="store_true",
                    help="also time trp_preprocess_batch on host buffers (PCIe both ways included)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other single-GPU workloads (configs C, D, E on one GPU, host-memory path) that "
                         "the default one-GPU run reports next to the headline value")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)   # does not return

    import numpy as np
    import torch
    import corpusgen
    import triplets_amd as T

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    # Rehearsal aids (never set by the driver): TRP_BENCH_DEVICE pins every rank to one GPU and
    # TRP_BENCH_BACKEND=gloo replaces RCCL, so that the multi-rank code path can run on a one-GPU box.
    device = int(os.environ.get("TRP_BENCH_DEVICE", local_rank))
    backend = os.environ.get("TRP_BENCH_BACKEND", "nccl")
    if device >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but %d GPU(s) are visible" % (rank, device, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    # ---- synthetic batches for this rank (shard = its own record range of the corpus) ---------------
    threads = max(1, 16 // max(1, world))
    t0 = time.time()
    batches = []      # (device bytes, device offsets, R, N), all resident in HBM before the timed region
    host0 = None      # host copy of the first batch (CPU baseline, host path)
    if args.config == "E":
        total = args.bytes or 64_000_000_000
        lo, hi = total * rank // world, total * (rank + 1) // world     # this rank's byte range of the corpus
        n_slices = max(1, -(-(hi - lo) // 4_000_000_000))
        for i in range(n_slices):
            a, b = lo + (hi - lo) * i // n_slices, lo + (hi - lo) * (i + 1) // n_slices
            data, offs = corpusgen.config_c(b - a, 45 + 1000 * rank + i, threads)
            batches.append((torch.from_numpy(data).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda(),
                            int(len(offs) - 1), int(len(data))))
            if host0 is None:
                host0 = (data, offs)
            del data, offs
        workload = ("config E: %.1e B synthetic OCR-noisy markdown corpus, one record-range shard per GPU, "
                    "%d slice(s) of <= 4e9 B per shard" % (total, n_slices))
    else:
        nbytes = args.bytes or 1_000_000_000
        seed = {"B": 42, "C": 43, "D": 44, "W": 46}[args.config] + 1000 * rank
        gen = {"B": corpusgen.config_b, "C": corpusgen.config_c, "D": corpusgen.config_d, "W": corpusgen.config_w}[args.config]
        data, offs = gen(nbytes, seed, threads)
        workload = workload_name(args.config, nbytes)
        batches.append((torch.from_numpy(data).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda(),
                        int(len(offs) - 1), int(len(data))))
        host0 = (data, offs)
    gen_s = time.time() - t0
    data, offs = host0
    N = sum(b[3] for b in batches)
    R = sum(b[2] for b in batches)
    nb = len(batches)
    torch.cuda.synchronize()

    den = T.DenoiserConfig(enabled=not args.no_denoise, max_digit_ratio=0.35, strip_markdown=True)
    st = T.ChunkingStrategy()   # reference defaults: 1024 / [64] / 512
    ctx = T.Context(device)

    def step(after_batch=None):
        """One pass over the rank's shard; returns (denoised bytes, tokens, chunks)."""
        d = t = c = 0
        for d_bytes, d_offs, r, n in batches:
            if args.normalize:
                res = ctx.normalize_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n)
            else:
                res = ctx.preprocess_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n, den, st)
            d, t, c = d + int(res.n_denoised), t + int(res.n_tokens), c + int(res.n_chunks)
            if after_batch is not None:
                after_batch()
        return d, t, c

    # Timed region: back-to-back blocking calls on one context (each returns after its kernels have
    # completed).  With --in-flight, batches are kept in flight on two contexts (two sets of result buffers,
    # one stream each) the way a consumer of one batch overlaps with the production of the next; every
    # batch of every step is still waited for inside the region.
    ctx2 = T.Context(device) if (args.in_flight and not args.normalize) else None

    def run_steps(k):
        """k passes over the shard; returns the (denoised bytes, tokens, chunks) of the last pass."""
        if ctx2 is None:
            for _ in range(k):
                r = step()
            return r
        cs, busy, tot = (ctx, ctx2), [False, False], [0, 0, 0]
        i = 0

        def reap(j, last):
            r = cs[j].wait_batch()
            busy[j] = False
            if last:
                tot[0] += int(r.n_denoised); tot[1] += int(r.n_tokens); tot[2] += int(r.n_chunks)

        pend = [False, False]   # whether the batch in flight on context j belongs to the last pass
        for s in range(k):
            for d_bytes, d_offs, r, n in batches:
                j = i % 2
                if busy[j]:
                    reap(j, pend[j])
                cs[j].submit_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), r, n, den, st)
                busy[j], pend[j] = True, s == k - 1
                i += 1
        for j in (i % 2, (i + 1) % 2):
            if busy[j]:
                reap(j, pend[j])
        return tuple(tot)

    def timed(k):
        """K steps bracketed by barrier + synchronize on both sides; the time is the maximum over the ranks."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run_steps(k)                 # returns after every batch of every step has completed
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, r

    if dist is not None:
        nn = torch.tensor([N], dtype=torch.float64, device="cuda")
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_bytes = float(nn.item())
    else:
        total_bytes = float(N)
    # first the plain reading: W warm-up steps and K timed steps straight after the corpus was generated (the GPU
    # has been idle and is still clocking up), reported as "value_no_settle" ...
    if args.warmup:
        run_steps(args.warmup)
    elapsed_cold, _ = timed(args.steps)
    # ... then the steady rate: untimed passes for SETTLE_S, the caller's W, the caller's K
    settle_steps = settle(lambda: run_steps(1))
    if args.warmup:
        run_steps(args.warmup)
    elapsed, res = timed(args.steps)

    stats = {"N": N, "R": R, "D": res[0], "T": res[1], "C": res[2], "enabled": den.enabled}
    per_launch = {k: (v // nb if k != "enabled" else v) for k, v in stats.items()}   # average batch of the shard
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e9

    # ---- per-kernel durations (HIP events on the launch stream), dominant kernel roofline ----------
    kern = kernel_profile(ctx, step, nb, max(3, min(args.steps, 10)))
    roofline = roofline_of(kern, stats, nb, ms_per_step)
    dominant = roofline["kernel"]
    # HBM bytes per launch of the dominant kernel from the committed counter runs (tools/collect_profiles.sh),
    # used only when they were taken on this very workload
    if nb == 1 and not args.normalize:
        roofline["traffic"] = committed_traffic(workload, N, den.enabled, dominant)

    out = {
        "metric": "device-resident GB/s of text denoised+tokenised+chunked",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_ms": SETTLE_S * 1e3, "settle_steps": settle_steps,
        "value_no_settle": round(total_bytes * args.steps / elapsed_cold / 1e9, 2),
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if args.config == "E" else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload, "bytes_per_gpu": N, "records_per_gpu": R, "batches_per_step": nb,
                   "denoiser": "normalize_inline_whitespace" if args.normalize else ("enabled" if den.enabled else "disabled"),
                   "chunking": "max_window_tokens=1024 overlap=[64] summary_fallback_tokens=512",
                   "sharding": "record ranges, one shard per GPU, no collective",
                   "calls": "blocking, one context" if ctx2 is None else
                            "submit/wait, one batch in flight on each of two contexts"},
        "outputs": {"denoised_bytes": stats["D"], "tokens": stats["T"], "chunks": stats["C"]},
        "roofline": roofline,
        "corpus_generation_s": round(gen_s, 2),
    }

    # ---- the same batch handed over in host memory: H2D of the input, D2H of the selected result arrays ----
    extras = (world == 1 and rank == 0 and args.config == "B" and not args.no_extras and not args.normalize
              and not args.no_denoise and ctx2 is None and args.bytes is None)
    if (args.host_path or extras) and rank == 0:
        N0, R0 = batches[0][3], batches[0][2]     # the first batch of the shard
        pinned = torch.from_numpy(data).pin_memory()
        offs64 = np.ascontiguousarray(offs, dtype=np.uint64)
        L = T.load_library()
        cfg = T._make_cfg(den, st)
        masks = [("all arrays", T.OUT_ALL), ("chunk descriptors + chunk text", T.OUT_CHUNKS | T.OUT_CHUNK_TEXT)]
        hp = {}
        for label, mask in masks:
            best = None
            for _ in range(3):
                r, ct = T._Result(), T._ChunkText()
                t0 = time.perf_counter()
                rc = L.trp_preprocess_batch_select(ctx._h, pinned.data_ptr(), offs64.ctypes.data, R0, C.byref(cfg), mask,
                                                   C.byref(r), C.byref(ct) if mask & T.OUT_CHUNK_TEXT else None)
                dt = time.perf_counter() - t0
                if rc != 0:
                    raise SystemExit("trp_preprocess_batch_select failed: %d" % rc)
                L.trp_result_free(C.byref(r))
                if mask & T.OUT_CHUNK_TEXT:
                    L.trp_chunk_text_free(C.byref(ct))
                best = dt if best is None else min(best, dt)
            hp[label] = {"value": round(N0 / best / 1e9, 3), "unit": "GB/s", "ms": round(best * 1e3, 2)}
        out["host_path"] = {"value": hp["all arrays"]["value"], "unit": "GB/s", "ms": hp["all arrays"]["ms"],
                            "by_outputs": hp,
                            "note": "pinned input in host memory, H2D + kernels + D2H of the selected result arrays "
                                    "(the per-record arrays always come back), best of 3"}
        del pinned
        L.trp_trim_host_pool()

    # ---- small batches: microseconds per blocking call (a sampler's ingest refresh is kilobytes to megabytes) ----
    if extras:
        sb = {}
        d_all, o_all, _, _ = batches[0][0], batches[0][1], batches[0][2], batches[0][3]
        for label, want in (("one_record_4KB", 1), ("64KB", 16), ("1MB", 256), ("4MB", 1024)):
            r = min(want, int(batches[0][2]))
            n = int(offs[r])
            for _ in range(30):
                ctx.preprocess_batch_device(d_all.data_ptr(), o_all.data_ptr(), r, n, den, st)
            runs = []
            for _ in range(7):
                t0 = time.perf_counter()
                for _ in range(100):
                    ctx.preprocess_batch_device(d_all.data_ptr(), o_all.data_ptr(), r, n, den, st)
                runs.append((time.perf_counter() - t0) / 100)
            dt = sorted(runs)[3]
            sb[label] = {"records": r, "bytes": n, "us_per_call": round(dt * 1e6, 1), "GBps": round(n / dt / 1e9, 2)}
        out["small_batches"] = {"calls": "blocking trp_preprocess_batch_device from Python, device-resident input, "
                                         "median of 7 runs of 100 calls", **sb}

    # ---- the other single-GPU workloads of BASELINE.json, same context, same timing rules ---------------
    if extras:
        ex = {}
        keep_host = (data, offs)
        out["in_flight"] = run_in_flight(T, device, batches, den, st, max(args.steps, 10), 3)
        out["in_flight_out_of_phase"] = run_in_flight(T, device, batches, den, st, max(args.steps, 10), 3,
                                                      stagger_s=0.45 * out["ms_per_step"] / max(1, nb) * 1e-3)

        def to_dev(dd, oo):
            return (torch.from_numpy(dd).cuda(), torch.from_numpy(oo.astype(np.int64)).cuda(), int(len(oo) - 1), int(len(dd)))

        # config B a second time with the denoiser disabled (tokenise + chunk only), and the step before the
        # path (normalize_inline_whitespace) over the same batch
        ex["B_denoiser_disabled"] = run_extra(T, ctx, "B", workload, batches,
                                              T.DenoiserConfig(enabled=False, max_digit_ratio=0.35, strip_markdown=True),
                                              st, args.steps, args.warmup)
        ex["B_normalize_inline_whitespace"] = run_extra(T, ctx, "B", workload, batches, den, st, args.steps, args.warmup,
                                                        normalize=True)

        dd, oo = corpusgen.config_c(4_000_000_000, 43, threads)
        bt = [to_dev(dd, oo)]
        del dd, oo
        ex["C"] = run_extra(T, ctx, "C", workload_name("C", 4_000_000_000), bt, den, st, 5, 2)
        del bt
        torch.cuda.empty_cache()
        dd, oo = corpusgen.config_d(1_000_000_000, 44, threads)
        bt = [to_dev(dd, oo)]
        del dd, oo
        wl = workload_name("D", 1_000_000_000)
        ex["D"] = run_extra(T, ctx, "D", wl, bt, den, st, 5, 2)
        ex["D_overlaps_64_256"] = run_extra(T, ctx, "D", wl, bt, den, T.ChunkingStrategy(overlap_tokens=[64, 256]), 5, 2)
        del bt
        torch.cuda.empty_cache()
        dd, oo = corpusgen.config_w(1_000_000_000, 46, threads)
        bt = [to_dev(dd, oo)]
        del dd, oo
        ex["C_wide_scripts"] = run_extra(T, ctx, "W", workload_name("W", 1_000_000_000), bt, den, st, 5, 2)
        del bt
        torch.cuda.empty_cache()
        bt = []
        for i in range(2):
            dd, oo = corpusgen.config_c(4_000_000_000, 45 + i, threads)
            bt.append(to_dev(dd, oo))
            del dd, oo
        ex["E_1gpu"] = run_extra(T, ctx, "E", "config E on one GPU: the first 8.0e9 B of the 64e9 B corpus' shard 0 "
                                 "(config-C generator, seeds 45, 46), 2 slices of 4e9 B per step", bt, den, st, 3, 1,
                                 traffic_workload="config E: %.1e B synthetic OCR-noisy markdown corpus, one record-range "
                                                  "shard per GPU, %d slice(s) of <= 4e9 B per shard" % (8_000_000_000, 2))
        del bt
        torch.cuda.empty_cache()
        out["configs"] = ex
        data, offs = keep_host

    # ---- CPU baseline: the oracle (port of the reference algorithm), bounded sample, rank 0 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.normalize:
        import oracle_lib as O
        N, R = batches[0][3], batches[0][2]     # the first batch of the shard
        if nb > 1:
            first = ctx.preprocess_batch_device(batches[0][0].data_ptr(), batches[0][1].data_ptr(), R, N, den, st)
            stats = {"D": int(first.n_denoised), "T": int(first.n_tokens), "C": int(first.n_chunks)}
        cfg = O.make_config(den.enabled, 0.35, True, 1024, (64,), 512)
        cut = int(np.searchsorted(offs, min(N, 1_000_000_000), side="right")) - 1   # about 10 s on one core
        cut = max(cut, 1)
        sub_n = int(offs[cut])
        L = O.lib()
        r1 = O.OrcResult()
        t0 = time.perf_counter()
        L.orc_batch_mt(data.ctypes.data, offs.ctypes.data, cut, C.byref(cfg), 1, C.byref(r1))
        t1 = time.perf_counter() - t0
        L.orc_result_free(C.byref(r1))
        cores = min(16, os.cpu_count() or 1)
        r2 = O.OrcResult()
        t0 = time.perf_counter()
        L.orc_batch_mt(data.ctypes.data, offs.ctypes.data, R, C.byref(cfg), cores, C.byref(r2))
        t2 = time.perf_counter() - t0
        ok = int(r2.n_tokens) == stats["T"] and int(r2.n_denoised) == stats["D"] and int(r2.n_chunks) == stats["C"]
        L.orc_result_free(C.byref(r2))
        out["cpu_baseline"] = {"value": round(sub_n / t1 / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": "first %d records (%.0f MB) of the same batch, CPU restatement of the "
                                         "reference algorithm, 1 thread" % (cut, sub_n / 1e6)}
        out["cpu_baseline_all_cores"] = {"value": round(N / t2 / 1e9, 4), "unit": "GB/s", "cores": cores,
                                         "kind": "port", "sample": "whole batch, record-parallel static partition",
                                         "totals_match_gpu": bool(ok)}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if ctx2 is not None:
        ctx2.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
