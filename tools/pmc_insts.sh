#!/bin/bash
# Dynamic instruction counts per kernel for one bench configuration.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_insts
rm -rf $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $OUT -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT.json 2> $OUT.log || exit 1
python3 tools/pmc_summary.py $OUT > $OUT.txt
python3 - <<'PY'
import re
t=open("gpurun_out/pmc_insts.txt").read().split("\n")
cur=None; d={}
for l in t:
    if l and not l.startswith(" "): cur=l.strip(); d[cur]={}
    elif l.strip(): k,v=l.split()[:2]; d[cur][k]=float(v)
for k,v in d.items():
    if v.get("SQ_INSTS_VALU",0)>1e6:
        w=v["SQ_WAVES"]
        print("%-18s waves %7d  per wave: VALU %6.0f SALU %5.0f LDS %4.0f BR %4.0f VMEM %3.0f/%3.0f"%(k,w,v["SQ_INSTS_VALU"]/w,v["SQ_INSTS_SALU"]/w,v["SQ_INSTS_LDS"]/w,v["SQ_INSTS_BRANCH"]/w,v["SQ_INSTS_VMEM_RD"]/w,v["SQ_INSTS_VMEM_WR"]/w))
PY
