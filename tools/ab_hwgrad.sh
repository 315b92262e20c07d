#!/bin/bash
# A/B of two builds of the library on hwgrad_kernel (base = libwavenet_amd_base.so next to the product library), one box, alternating
for i in 1 2; do for lib in libwavenet_amd_base.so libwavenet_amd.so; do
  for cfg in "--config cfg5" "--precision f16x3" "--config cfg2"; do
    WN_LIB=$lib python tools/run_bench_with_lib.py $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown --no-second-line > gpurun_out/hw_probe.json 2>/dev/null
    python -c "
import json; b=json.loads(open('gpurun_out/hw_probe.json').read().strip().splitlines()[-1]); k=b['kernels']['hwgrad_kernel']; print('$lib', '$cfg', 'step', b['ms_per_step'], 'hwgrad us', round(1e3*k['avg_ms'],1), 'TFLOP/s', k['tflops'])"
  done
done; done
