#!/bin/bash
# Regenerates the judged profile artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r03/cfg2_bf16 --config cfg2 --precision bf16'
# <round> may be a path (one directory per configuration and precision, so that no record quotes another run's counters)
# 1. rocprofv3 --kernel-trace --stats of the bench command             -> gpurun_out/<round>/kernel_stats_*.csv
# 2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (HBM traffic)  -> gpurun_out/<round>/pmc_hbm_traffic.csv
# 3. separate --pmc passes for the MFMA counters                        -> gpurun_out/<round>/pmc_mfma.csv
#    (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE: busy fraction; SQ_INSTS_MFMA / SQ_BUSY_CU_CYCLES: instruction counts)
# Counter passes carry no trace options (gpurun refuses --pmc together with the hip/hsa trace domains); the program sits
# directly after `--`.  Copy the results from gpurun_out/<round>/ into profiles/<round>/ afterwards (gpurun_out is scratch).
set -e
ROUND=${1:-r03}
shift || true
EXTRA="$@"
TAG=${WN_PROFILE_TAG:-}
STEPS=${WN_PROFILE_STEPS:-3}      # timed steps of the kernel-stats pass (short steps: use more)
export TMPDIR=/tmp
OUT=gpurun_out/$ROUND
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps $STEPS --warmup 1 --no-breakdown $EXTRA \
    > $OUT/bench_under_rocprof$TAG.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_steps${STEPS}_warmup1$TAG.csv
if [ -n "$WN_PROFILE_KEEP_TRACE" ]; then cp $(find $OUT/stats -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace$TAG.csv; fi
echo "[collect] kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-breakdown --no-cpu-baseline $EXTRA \
        > $OUT/bench_under_pmc_$c.json 2> $OUT/pmc_$c.err
    echo "[collect] pmc $c done"
done
python3 tools/summarize_pmc.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_hbm_traffic$TAG.csv
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cat $OUT/pmc_hbm_traffic$TAG.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_busy -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-breakdown --no-cpu-baseline $EXTRA \
    > $OUT/bench_under_pmc_mfma_busy.json 2> $OUT/pmc_mfma_busy.err
echo "[collect] pmc mfma busy done"
# instruction-count pass: counter names differ between ROCm releases, so this pass may be refused -- keep going without it
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma_insts -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-breakdown --no-cpu-baseline $EXTRA \
    > $OUT/bench_under_pmc_mfma_insts.json 2> $OUT/pmc_mfma_insts.err || echo "[collect] instruction-count pass failed (see pmc_mfma_insts.err)"
python3 tools/summarize_mfma.py $OUT/pmc_mfma_busy $OUT/pmc_mfma_insts > $OUT/pmc_mfma$TAG.csv
rm -rf $OUT/pmc_mfma_busy $OUT/pmc_mfma_insts
cat $OUT/pmc_mfma$TAG.csv
