#!/usr/bin/env python3
"""Experiment: two contexts, each on a stream restricted to a set of compute units, one batch in flight on each.

k_plan is bound by instruction issue and k_emit by memory bandwidth; two batches whose phases are offset could
share the GPU better than one batch that owns it.  Streams made with hipExtStreamCreateWithCUMask are handed to
trp_create_on_stream; the library is unchanged.

  python tools/cu_mask_inflight.py [--config B] [--steps 10] [--split 128]   (CUs of the first context)

Prints GB/s for: blocking calls on one unmasked context, two unmasked contexts in flight, two masked contexts in
flight (contiguous halves and interleaved CUs)."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-triplets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools", "corpusgen"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--split", type=int, default=128)
    ap.add_argument("--bytes", type=int, default=1_000_000_000)
    ap.add_argument("--no-masks", action="store_true", help="only the runs on unmasked streams")
    args = ap.parse_args()
    import numpy as np
    import torch
    import corpusgen
    import triplets_amd as T

    torch.cuda.set_device(0)
    hip = C.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32

    def masked_stream(cus):
        m = (C.c_uint32 * words)()
        for cu in cus:
            m[cu // 32] |= 1 << (cu % 32)
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), m)
        assert rc == 0, rc
        return s.value

    gen = {"B": corpusgen.config_b, "C": corpusgen.config_c, "D": corpusgen.config_d}[args.config]
    seed = {"B": 42, "C": 43, "D": 44}[args.config]
    batches = []
    for i in range(2):   # two different batches
        data, offs = gen(args.bytes, seed + 7 * i, 16)
        batches.append((torch.from_numpy(data).cuda(), torch.from_numpy(offs.astype(np.int64)).cuda(),
                        int(len(offs) - 1), int(len(data))))
    torch.cuda.synchronize()
    den = T.DenoiserConfig(enabled=True, max_digit_ratio=0.35, strip_markdown=True)
    st = T.ChunkingStrategy()
    nbytes = sum(b[3] for b in batches)

    def blocking(ctx):
        for _ in range(3):
            for b in batches:
                ctx.preprocess_batch_device(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3], den, st)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for b in batches:
                ctx.preprocess_batch_device(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3], den, st)
        return nbytes * args.steps / (time.perf_counter() - t0) / 1e9

    def in_flight(c0, c1, stagger_s=0.0):
        cs = (c0, c1)
        for j in (0, 1):   # warm-up: buffers
            for _ in range(2):
                cs[j].preprocess_batch_device(batches[j][0].data_ptr(), batches[j][1].data_ptr(), batches[j][2],
                                              batches[j][3], den, st)
        t0 = time.perf_counter()
        busy = [False, False]
        for s in range(args.steps):
            for j in (0, 1):
                if busy[j]:
                    cs[j].wait_batch()
                elif j == 1 and stagger_s > 0:   # the second context starts out of phase with the first
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < stagger_s:
                        pass
                b = batches[j]
                cs[j].submit_batch_device(b[0].data_ptr(), b[1].data_ptr(), b[2], b[3], den, st)
                busy[j] = True
        for j in (0, 1):
            cs[j].wait_batch()
        return nbytes * args.steps / (time.perf_counter() - t0) / 1e9

    one = T.Context(0)
    print("blocking, one context, all CUs: %.1f GB/s" % blocking(one), flush=True)
    two = T.Context(0)
    print("in flight, two contexts, all CUs: %.1f GB/s" % in_flight(one, two), flush=True)
    print("in flight, two contexts, all CUs, second context 0.6 ms out of phase: %.1f GB/s" % in_flight(one, two, 0.0006),
          flush=True)
    same = T.Context(0, stream=one.stream)   # a second context on the first one's stream: the batches run one after
    print("in flight, two contexts on one stream: %.1f GB/s" % in_flight(one, same), flush=True)   # the other, no idle GPU between them
    same.close()
    k = args.split
    for name, a, b in () if args.no_masks else (("contiguous", range(0, k), range(k, n_cu)),
                       ("interleaved", [c for c in range(n_cu) if (c * k // n_cu) != ((c + 1) * k // n_cu)],
                        [c for c in range(n_cu) if (c * k // n_cu) == ((c + 1) * k // n_cu)])):
        a, b = list(a), list(b)
        ca, cb = T.Context(0, stream=masked_stream(a)), T.Context(0, stream=masked_stream(b))
        rate = blocking(ca)
        print("blocking, one context on %d CUs (%s): %.1f GB/s" % (len(a), name, rate), flush=True)
        print("in flight, %d + %d CUs (%s): %.1f GB/s" % (len(a), len(b), name, in_flight(ca, cb)), flush=True)
        half = 0.5 * batches[0][3] / (rate * 1e9)
        print("in flight, %d + %d CUs (%s), second context %.2f ms out of phase: %.1f GB/s"
              % (len(a), len(b), name, half * 1e3, in_flight(ca, cb, half)), flush=True)
        ca.close()
        cb.close()
    one.close()
    two.close()


if __name__ == "__main__":
    main()
