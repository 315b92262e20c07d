for i in 1 2; do
  for lib in libwavenet_amd_base.so libwavenet_amd.so; do
    WN_LIB=$lib python tools/run_bench_with_lib.py --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-breakdown > gpurun_out/ab_$lib.$i.json 2>/dev/null
    python -c "
import json,sys; b=json.load(open('gpurun_out/ab_$lib.$i.json')); print('$lib cfg3 f16x3', b['ms_per_step'], {k:v['avg_ms'] for k,v in b['kernels'].items() if k.startswith('hgemm')})"
    WN_LIB=$lib python tools/run_bench_with_lib.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown > gpurun_out/ab5_$lib.$i.json 2>/dev/null
    python -c "
import json,sys; b=json.load(open('gpurun_out/ab5_$lib.$i.json')); print('$lib cfg5 f16', b['ms_per_step'], {k:v['avg_ms'] for k,v in b['kernels'].items() if k.startswith('hgemm')})"
  done
done
