#!/usr/bin/env python3
"""Print value / ms per step / per-kernel-class averages of a bench.py JSON line (file argument or stdin)."""
import json, sys
d = json.loads(open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read())
print("%s  %.1f samples/s  %.2f ms/step  %s" % (d["dtype"][:6], d["value"], d["ms_per_step"], d.get("breakdown")))
steps = d["steps"]
for k, v in d["kernels"].items():
    print("  %-36s %8.3f ms/launch  %7.2f ms/step  %s" % (k, v["avg_ms"], v["ms_total"] / steps, v["tflops"]))
