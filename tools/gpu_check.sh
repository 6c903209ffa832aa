#!/bin/bash
# GPU box: parity tests, then the bench on configs B, C, D (each step under its own time limit; stops at
# the first failure).  Usage: tools/gpu_check.sh [quick]   -- quick skips the full-size property tests.
set -o pipefail
mkdir -p gpurun_out
SEL="gpu"
[ "$1" = "quick" ] && SEL="gpu and not full_size and not one_gigabyte and not skewed"
timeout -k 10 600 python -u -m pytest tests -m gpu -k "$SEL" -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for c in B C D; do
  timeout -k 10 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --config $c > gpurun_out/b$c.json 2> gpurun_out/b$c.err || { tail -5 gpurun_out/b$c.err; exit 1; }
  python - gpurun_out/b$c.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1][-7:-5], d['value'],'GB/s', d['ms_per_step'],'ms', {k:v for k,v in list(d['roofline']['kernels_ms'].items())[:6]})
PY
done
