#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box:
#   kernel-trace statistics of a bench run, then FETCH_SIZE and WRITE_SIZE in separate counter passes,
#   then the per-kernel HBM traffic summary.  Usage: tools/collect_profiles.sh TAG [bench args...]
# Output: gpurun_out/prof_TAG/{bench.json,kernel_stats.csv,traffic.json,...}
set -eo pipefail
TAG=$1; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- \
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > "$OUT/bench.json" 2> "$OUT/stats.log"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/fetch" -o fetch -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/fetch.json" 2> "$OUT/fetch.log"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/write" -o write -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/write.json" 2> "$OUT/write.log"
python3 tools/pmc_summary.py --traffic "$OUT/fetch" "$OUT/write" "$OUT/traffic.json" > "$OUT/traffic.txt"
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
t = json.load(open(out + "/traffic.json"))
b = json.load(open(out + "/bench.json"))
t["_meta"] = {"bytes_per_gpu": b["config"]["bytes_per_gpu"], "workload": b["config"]["workload"],
              "denoiser": b["config"]["denoiser"]}
json.dump(t, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
PY
cat "$OUT/traffic.txt"
