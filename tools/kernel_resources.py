#!/usr/bin/env python3
"""Print VGPR/AGPR/spill/LDS/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "wavenet_speech_amd", "csrc")
files = sys.argv[1:] or ["wn_gemm.hip", "wn_wgrad.hip", "wn_pack.hip"]
# per-file flags of csrc/Makefile (FLAGS_<file>): report the product's code
extra = {"wn_half.hip": ["-fno-slp-vectorize"], "wn_fused.hip": ["-fno-slp-vectorize"], "wn_col.hip": ["-fno-slp-vectorize"],
         "wn_col_conv.hip": ["-fno-slp-vectorize"], "wn_col_skip.hip": ["-fno-slp-vectorize"], "wn_col2.hip": ["-fno-slp-vectorize"]}
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "--offload-arch=gfx950"] + extra.get(f, []) + ["-c", os.path.join(src, f),
                          "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:70]}
            continue
        m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*): (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
            if m.group(1).startswith("LDS Size"):
                print("%-72s VGPR %4s AGPR %4s SGPR %4s spill %s/%s scratch %s LDS %6s occ %s" % (
                    cur["name"], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("VGPRs Spill"),
                    cur.get("SGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("LDS Size [bytes/block]"),
                    cur.get("Occupancy [waves/SIMD]")))
                cur = None
