#!/bin/bash
# every judged profile of the round, one directory per (config, precision): gpurun --timeout 1200 -- 'bash tools/prof_all.sh r03'
set -e
R=${1:-r03}
bash tools/collect_profiles.sh $R/cfg3_f32 --config cfg3 --precision f32 --no-second-line
echo "== cfg3 f32 done"
bash tools/collect_profiles.sh $R/cfg3_f16x3 --config cfg3 --precision f16x3
echo "== cfg3 f16x3 done"
WN_PROFILE_STEPS=20 bash tools/collect_profiles.sh $R/cfg2_bf16 --config cfg2 --precision bf16
echo "== cfg2 bf16 done"
bash tools/collect_profiles.sh $R/cfg5_f16 --config cfg5 --precision f16
echo "== cfg5 f16 done"
bash tools/collect_profiles.sh $R/cfg5_f32 --config cfg5 --precision f32
echo "== cfg5 f32 done"
