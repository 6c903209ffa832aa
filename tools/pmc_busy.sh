#!/bin/bash
# Where the waves of the main kernels spend their cycles (SQ counters, one pass).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_busy
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -f csv -d $OUT -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT.json 2> $OUT.log || exit 1
python3 - $OUT <<'PY'
import csv,glob,sys
from collections import defaultdict
rows=defaultdict(lambda: defaultdict(dict))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        for k in ("k_plan<","k_emit<","k_seed<"):
            if k in n:
                d=rows[k][int(r["Dispatch_Id"])]
                d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k,v in rows.items():
    last=v[max(v)]
    w=last["SQ_WAVES"]
    print(k, "waves %d"%w, {c.replace("SQ_",""):round(x/w) for c,x in last.items() if c!="SQ_WAVES"})
PY
