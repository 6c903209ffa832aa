#!/bin/bash
# Dynamic instruction counts of k_plan up to the given stop points (cumulative), per 4 KiB wave:
#   tools/pmc_plan_stops.sh "2 21 22 3" [bench args...]     (a library built with -DTRP_FINE_STOPS has the two-digit stops)
export TMPDIR=/tmp
STOPS="$1"; shift
for d in $STOPS; do
  OUT=gpurun_out/pmc_ps$d
  rm -rf $OUT
  TRP_PROFILE_PLAN_STOP=$d timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -f csv -d $OUT -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT.json 2> $OUT.log || exit 1
  python3 - $OUT $d <<'PY'
import csv,glob,sys
from collections import defaultdict
rows=defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_plan" in r["Kernel_Name"]:
            d=rows[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
last=rows[max(rows)]
w=last["SQ_WAVES"]
print("stop",sys.argv[2],{k.replace("SQ_INSTS_",""):round(v/w) for k,v in last.items() if k!="SQ_WAVES"})
PY
done
