for b in 4 8 12 16 32; do
  python bench.py --steps 6 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('batch $b', 'samples/s', d['value'], 'ms', d['ms_per_step'], ' '.join('%s=%.1f'%(n.split('<')[-1].rstrip('>'),v['tflops']) for n,v in k.items() if v['tflops']))"
done
