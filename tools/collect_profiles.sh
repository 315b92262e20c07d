#!/bin/bash
# Regenerates the judged profile artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh r01'
# 1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/<round>/kernel_stats_*.csv
# 2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (HBM traffic) -> gpurun_out/<round>/pmc_hbm_traffic.csv
# Copy the results from gpurun_out/<round>/ into profiles/<round>/ afterwards (gpurun_out is scratch).
set -e
ROUND=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/$ROUND
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 3 --warmup 1 --no-breakdown \
    > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps3_warmup1.csv
echo "[collect] kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-breakdown --no-cpu-baseline \
        > $OUT/bench_under_pmc_$c.json 2> $OUT/pmc_$c.err
    echo "[collect] pmc $c done"
done
python3 tools/summarize_pmc.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_hbm_traffic.csv
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cat $OUT/pmc_hbm_traffic.csv
