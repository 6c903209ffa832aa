#!/bin/bash
# Times the phases of k_plan on the bench workload (cumulative ms after each phase; 0 = whole kernel).
for d in 1 2 3 4 5 6 0; do
  TRP_PROFILE_PLAN_STOP=$d timeout -k 10 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/pphase$d.json 2>/dev/null || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/pphase$d.json'));print('after phase $d:', d['roofline']['kernels_ms'].get('k_plan'))"
done
