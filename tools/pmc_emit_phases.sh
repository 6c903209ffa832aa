#!/bin/bash
# Dynamic instruction counts of k_emit per phase (cumulative), config B.
export TMPDIR=/tmp
for d in 2 3 4 5 0; do
  OUT=gpurun_out/pmc_ep$d
  rm -rf $OUT
  TRP_PROFILE_EMIT_STOP=$d timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -f csv -d $OUT -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT.json 2> $OUT.log || exit 1
  python3 - $OUT $d <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_emit" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# the profiled (phase-stopped) dispatches are the last ones: bench runs warmup/timed calls unprofiled first
w=acc["SQ_WAVES"]
print("stop", sys.argv[2], {k: round(min(v)/min(w)) for k,v in acc.items() if k!="SQ_WAVES"}, "per wave (min over dispatches)", "max:", {k: round(max(v)/max(w)) for k,v in acc.items() if k!="SQ_WAVES"})
PY
done
