#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (average per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print("   %-28s %16.0f  (avg of %d dispatches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))


if __name__ == "__main__":
    main(sys.argv[1])
