#!/bin/bash
# A/B timing of library variants: tools/ab.sh "<bench args>" lib1.so lib2.so ...
# Prints value (GB/s), ms per step and the per-kernel durations (HIP events) of every variant.
ARGS="$1"; shift
for lib in "$@"; do
  TRIPLETS_AMD_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras $ARGS > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$lib FAILED"; tail -5 gpurun_out/ab_tmp.err; exit 1; }
  python3 - "$lib" "$ARGS" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().split("\n")[-1])
k=d["roofline"]["kernels_ms"]
print("%-28s %-10s %7.1f GB/s %7.4f ms | "%(sys.argv[1].split("/")[-1],sys.argv[2][:10],d["value"],d["ms_per_step"])+" ".join("%s %.3f"%(n.replace("k_",""),v) for n,v in list(k.items())[:6]), flush=True)
PY
done
