#!/bin/bash
# Times the phases of k_emit on the bench workload (cumulative ms after each phase; 0 = whole kernel).
for d in ${PHASES:-1 2 3 4 5 0}; do
  TRP_PROFILE_EMIT_STOP=$d timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/phase$d.json 2>/dev/null || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/phase$d.json'));print('after phase $d:', d['roofline']['kernels_ms'].get('k_emit'))"
done
