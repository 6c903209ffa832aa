#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (average per dispatch).

  pmc_summary.py DIR                         print every counter of every kernel
  pmc_summary.py --traffic FETCH_DIR WRITE_DIR OUT.json
        HBM bytes per launch of each kernel from two separate passes, one with FETCH_SIZE and one
        with WRITE_SIZE.  Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests
        of wide coalesced streaming reads (16 bytes per lane) at 64 bytes: the read side is doubled for the
        kernels whose reads are of that kind (STREAMING below: they load the corpus, and k_emit the plan,
        with one 16-byte load per lane).  The other kernels (binary searches, gathers of offsets, 8-byte scan
        loads) are reported as counted, marked "read_calibrated": false - their absolute read volume is
        uncalibrated and small.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


STREAMING = ("k_seed", "k_summary", "k_plan", "k_count_tokens", "k_emit")


def short_name(full):
    k = full.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return k.strip()


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short_name(row["Kernel_Name"])
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}, cnt


def main(argv):
    if argv[1] == "--traffic":
        fetch, _ = load(argv[2])
        write, _ = load(argv[3])
        out = {}
        for k in sorted(set(fetch) | set(write)):
            base = k.split("<")[0]
            streaming = base in STREAMING
            rd = (2.0 if streaming else 1.0) * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
            wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
            e = out.setdefault(base, {"read_bytes": 0.0, "write_bytes": 0.0, "read_calibrated": streaming})
            # template instances of one kernel that both ran are reported under their full names too
            e["read_bytes"] = max(e["read_bytes"], rd)
            e["write_bytes"] = max(e["write_bytes"], wr)
            out[k] = {"read_bytes": rd, "write_bytes": wr, "read_calibrated": streaming}
        for e in out.values():
            e["bytes"] = e["read_bytes"] + e["write_bytes"]
        with open(argv[4], "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        for k, e in sorted(out.items()):
            print("%-28s read %14.0f  write %14.0f  total %14.0f" % (k, e["read_bytes"], e["write_bytes"], e["bytes"]))
        return
    avg, cnt = load(argv[1])
    for k in sorted(avg):
        print(k)
        for c in sorted(avg[k]):
            print("   %-28s %16.0f  (avg of %d dispatches)" % (c, avg[k][c], cnt[k][c]))


if __name__ == "__main__":
    main(sys.argv)
