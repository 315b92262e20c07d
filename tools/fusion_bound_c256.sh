#!/bin/bash
# What could a fused block forward win at 256 channels in f16x3 (second half)?  In training the fused kernel
# still has to write z and sigmoid(g); what it saves is the residual product's READS of z and x.  Upper bound, measured: the same
# step with the residual launches reading their B operand from an L2-resident window (WN_HGEMM_DBG=8 on the HEPI_STORE class:
# results are garbage, times are what the launch costs without its HBM reads).  Needs the diagnostic library:
#   make -C wavenet_speech_amd/csrc l2operands   (here), then   gpurun -- 'bash tools/fusion_bound_c256.sh'
for i in 1 2; do
  WN_LIB=libwavenet_amd_l2ops.so python tools/run_bench_with_lib.py --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-breakdown > gpurun_out/fb_base.$i.json 2>/dev/null
  WN_HGEMM_DBG=8 WN_HGEMM_DBG_EPI=0 WN_FLAG_CHECK=off WN_LIB=libwavenet_amd_l2ops.so python tools/run_bench_with_lib.py --precision f16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-breakdown > gpurun_out/fb_l2.$i.json 2>gpurun_out/fb_l2.$i.err
  for f in fb_base fb_l2; do python -c "
import json; b=json.loads(open('gpurun_out/$f.$i.json').read().strip().splitlines()[-1]); k=b['kernels']; print('$f', b['ms_per_step'], {n: round(1e3*v['avg_ms'],1) for n,v in k.items() if n in ('hgemm_kernel<gate>','hgemm_kernel<res>','hgemm_kernel<dx>')})"; done
done
