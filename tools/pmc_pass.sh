#!/bin/bash
# One extra rocprofv3 counter pass over the bench (no trace options beside --pmc):
#   bash tools/pmc_pass.sh <out name> "<counters>" [bench.py args]      -> gpurun_out/<round>/<out name>.csv (per-kernel averages)
set -e
NAME=$1; CTRS=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/${WN_ROUND:-r02}
mkdir -p $OUT
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_$NAME -o pmc -- python3 bench.py --steps 1 --warmup 1 --no-breakdown --no-cpu-baseline "$@" \
    > $OUT/bench_under_pmc_$NAME.json 2> $OUT/pmc_$NAME.err
python3 - $OUT/pmc_$NAME > $OUT/$NAME.csv <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "wn::" in row["Kernel_Name"]:
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
names = sorted({c for k in acc for c in acc[k]})
w = csv.writer(sys.stdout); w.writerow(["kernel", "dispatches"] + names)
for k in sorted(acc):
    n = max(v[0] for v in acc[k].values())
    w.writerow([k[:110], n] + ["%.0f" % (acc[k][c][1] / acc[k][c][0]) if c in acc[k] else "" for c in names])
PY
rm -rf $OUT/pmc_$NAME
cat $OUT/$NAME.csv
