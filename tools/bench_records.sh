#!/bin/bash
# the committed bench records of the round (profiles/<round>/bench_*.json): gpurun -- 'bash tools/bench_records.sh r03'
R=${1:-r03}
mkdir -p gpurun_out/$R
python bench.py --steps 10 --warmup 3 > gpurun_out/$R/bench_cfg3_f32.json 2> gpurun_out/$R/bench_cfg3_f32.err
python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_cfg3_f16x3.json 2> gpurun_out/$R/bench_cfg3_f16x3.err
python bench.py --config cfg2 --steps 50 --warmup 5 > gpurun_out/$R/bench_cfg2_bf16.json 2> gpurun_out/$R/bench_cfg2_bf16.err
python bench.py --config cfg2 --graph off --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/$R/bench_cfg2_bf16_eager.json 2> gpurun_out/$R/bench_cfg2_bf16_eager.err
WN_COL_BWD=0 python bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/$R/bench_cfg2_bf16_tiled_bwd.json 2> gpurun_out/$R/bench_cfg2_bf16_tiled_bwd.err
WN_COL_PAIR=0 python bench.py --config cfg2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/$R/bench_cfg2_bf16_unpaired_bwd.json 2> gpurun_out/$R/bench_cfg2_bf16_unpaired_bwd.err
python bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/$R/bench_cfg5_f16.json 2> gpurun_out/$R/bench_cfg5_f16.err
python bench.py --config cfg5 --precision f32 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/$R/bench_cfg5_f32.json 2> gpurun_out/$R/bench_cfg5_f32.err
grep -h "timed" gpurun_out/$R/bench_*.err
