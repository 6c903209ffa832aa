for n in 1000000 4000000 16000000 64000000 256000000 1000000000; do
  timeout -k 10 120 python bench.py --bytes $n --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/curve_$n.json 2>/dev/null || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/curve_$n.json'));k=d['roofline']['kernels_ms'];print('%11d B: %7.1f GB/s %8.4f ms/step, kernels sum %.4f ms'%($n,d['value'],d['ms_per_step'],sum(k.values())))"
done
