#!/bin/bash
# Samples GPU power and shader clock (rocm-smi) every 0.5 s while bench.py runs; prints min/median/max.
# Usage (GPU box): bash tools/power_clock_trace.sh
# Usage: power_clock_trace.sh [out file] [bench.py arguments...]   (default: the headline configuration, 40 steps)
OUT=${1:-gpurun_out/power_clock.txt}
shift || true
ARGS=${@:---steps 40 --warmup 2}
mkdir -p $(dirname $OUT)
python3 bench.py $ARGS > gpurun_out/bench_power.json 2> gpurun_out/bench_power.err &
BP=$!
: > $OUT
while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks --json 2>/dev/null >> $OUT
    echo >> $OUT
    sleep 0.5
done
wait $BP
python3 - "$OUT" <<'PY'
import json, sys, statistics
pw, sclk = [], []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    for card, v in d.items():
        if not isinstance(v, dict):
            continue
        for k, val in v.items():
            kl = k.lower()
            try:
                if "power" in kl and "(w)" in kl: pw.append(float(val))
                if kl.startswith("sclk clock speed"): sclk.append(float(str(val).strip("()Mhz ")))
            except Exception:
                pass
def s(x): return "n=%d min %.0f median %.0f max %.0f" % (len(x), min(x), statistics.median(x), max(x)) if x else "none"
print("power W:", s(pw)); print("sclk MHz:", s(sclk))
PY
tail -c 600 gpurun_out/bench_power.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
