set -e
export TMPDIR=/tmp
OUT=gpurun_out/r03_mid/cfg2_bf16
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --config cfg2 --graph off --steps 20 --warmup 1 --no-breakdown --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
