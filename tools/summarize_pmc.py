#!/usr/bin/env python3
"""Average rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel into profiles/<round>/pmc_hbm_traffic.csv.

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB (1024 B);
on gfx950 FETCH_SIZE under-reports 16-byte-per-lane streams by 2x, so fetches are doubled; WRITE_SIZE is exact.
Usage: summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import collections, csv, glob, os, sys

print("counter,kernel,dispatches,avg_counter_value_KB,avg_bytes_corrected,note")
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name, ctr = row["Kernel_Name"], row["Counter_Name"]
                if "wn::" not in name:
                    continue
                a = acc[(ctr, name)]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    for (ctr, name), (n, tot) in sorted(acc.items()):
        avg_kb = tot / n
        if ctr == "FETCH_SIZE":
            w.writerow([ctr, name, n, "%.1f" % avg_kb, int(avg_kb * 1024 * 2), "x2 gfx950 FETCH_SIZE correction for 16B/lane streams"])
        else:
            w.writerow([ctr, name, n, "%.1f" % avg_kb, int(avg_kb * 1024), "exact"])
