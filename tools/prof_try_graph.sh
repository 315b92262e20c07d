export TMPDIR=/tmp
OUT=gpurun_out/r03_try
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --config cfg2 --steps 10 --warmup 1 --no-breakdown --no-cpu-baseline > $OUT/bench.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -3 $OUT/stats.err
head -5 $OUT/kernel_stats.csv
