set -e
WN_PROFILE_STEPS=20 WN_PROFILE_KEEP_TRACE=1 bash tools/collect_profiles.sh r03_before/cfg2_bf16 --config cfg2 --precision bf16
echo "== cfg2 done"
bash tools/collect_profiles.sh r03_before/cfg3_f16x3 --config cfg3 --precision f16x3
echo "== cfg3 f16x3 done"
bash tools/collect_profiles.sh r03_before/cfg5_f16 --config cfg5 --precision f16
echo "== cfg5 f16 done"
bash tools/collect_profiles.sh r03_before/cfg5_f32 --config cfg5 --precision f32
echo "== cfg5 f32 done"
