for cfg in "--config cfg5" "--precision f16x3" "--config cfg2"; do
  for dbg in 0 2; do
    WN_HWGRAD_DBG=$dbg WN_FLAG_CHECK=off python bench.py $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown --no-second-line > gpurun_out/hw_probe.json 2>/dev/null
    python -c "
import json; b=json.loads(open('gpurun_out/hw_probe.json').read().strip().splitlines()[-1]); k=b['kernels']['hwgrad_kernel']; print('$cfg', 'WN_HWGRAD_DBG=$dbg', 'step', b['ms_per_step'], 'hwgrad us', round(1e3*k['avg_ms'],1), 'TFLOP/s', k['tflops'])"
  done
done
