#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from rocprofv3 --pmc passes -> profiles/<round>/pmc_mfma.csv.

  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
      SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over every SIMD (64 per v_mfma_f32_32x32x2_f32, 32 per
      v_mfma_f32_32x32x16_f16: /opt/skills/guides/MI355X_MICROARCH.md cycle constants); GRBM_GUI_ACTIVE is reported as
      the sum over the 8 XCDs of the cycles the dispatch was active.  1.0 = every SIMD's matrix pipe busy all the time.
  mfma_insts / valu_insts: wave-level instruction counts (second pass; blank when that pass was not available).
Usage: summarize_mfma.py <dir of the busy pass> [<dir of the instruction-count pass>]"""
import collections, csv, glob, os, sys

N_XCD, N_SIMD = 8, 1024


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "wn::" not in name:
                    continue
                a = acc[name][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


busy = load(sys.argv[1])
insts = load(sys.argv[2]) if len(sys.argv) > 2 and os.path.isdir(sys.argv[2]) else {}
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "avg_mfma_busy_cycles", "avg_gui_active_sum8xcd", "mfma_busy_frac", "eff_clock_note",
            "avg_mfma_insts", "avg_valu_insts"])
for name in sorted(busy):
    c = busy[name]
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = c["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    b = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n
    g = c["GRBM_GUI_ACTIVE"][1] / c["GRBM_GUI_ACTIVE"][0]
    frac = b / (g / N_XCD * N_SIMD) if g > 0 else 0.0
    i = insts.get(name, {})
    mi = "%.0f" % (i["SQ_INSTS_MFMA"][1] / i["SQ_INSTS_MFMA"][0]) if "SQ_INSTS_MFMA" in i else ""
    vi = "%.0f" % (i["SQ_INSTS_VALU"][1] / i["SQ_INSTS_VALU"][0]) if "SQ_INSTS_VALU" in i else ""
    w.writerow([name, n, "%.0f" % b, "%.0f" % g, "%.4f" % frac, "gui_active/8 = active cycles per XCD", mi, vi])
