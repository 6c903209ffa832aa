# Does the headline figure depend on the number of warm-up and timed steps?  (clock ramp, first-touch effects)
for wk in "3 10" "10 30" "20 100" "3 10" "20 100"; do
  set -- $wk
  timeout -k 10 120 python bench.py --warmup $1 --steps $2 --no-cpu-baseline --no-extras > gpurun_out/sens.json 2>/dev/null || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/sens.json'));k=d['roofline']['kernels_ms'];print('warmup %3d steps %4d: %7.1f GB/s %8.4f ms/step, kernels sum %.4f ms, k_emit %.4f k_plan %.4f'%($1,$2,d['value'],d['ms_per_step'],sum(k.values()),k['k_emit'],k['k_plan']))"
done
