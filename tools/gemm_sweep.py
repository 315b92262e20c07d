#!/usr/bin/env python3
"""Series-GEMM microbenchmark (GPU): time the conv-forward instantiation for K = k*C, k = 1..6 taps, to split
the kernel time into a per-wave fixed cost and a per-k-block cost.  Usage: python tools/gemm_sweep.py [C] [B] [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import functional as HF

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
dev = "cuda:0"
x = torch.randn(B, C, L, device=dev)
res = []
for k in (1, 2, 3, 4, 5, 6):
    w = torch.randn(C, C, k, device=dev) * 0.05
    b = torch.randn(C, device=dev)
    with torch.no_grad():
        for _ in range(2):
            HF.dilated_conv(x, w, b, 1, True)
        torch.cuda.synchronize()
        HF.profile_reset(); HF.profile_enable(True)
        for _ in range(5):
            HF.dilated_conv(x, w, b, 1, True)
        torch.cuda.synchronize()
        HF.profile_enable(False)
    ms, n, fl = HF.profile_read()["series_gemm_kernel<conv_fwd>"]
    avg = ms / n
    res.append((k, avg))
    print("k=%d K=%5d  %.4f ms  %.1f TFLOP/s" % (k, k * C, avg, fl / n / avg / 1e9), flush=True)
# linear fit
import numpy as np
ks = np.array([r[0] for r in res], float); ts = np.array([r[1] for r in res])
slope, icpt = np.polyfit(ks, ts, 1)
nslab = max(1, (C + 127) // 128); waves = nslab * B * ((L + 127) // 128); rounds = waves / 1024.0
print("fit: %.4f ms per tap (%d k-blocks) + %.4f ms fixed;  waves=%d (%.2f per SIMD)" % (slope, C // 8, icpt, waves, rounds))
print("per wave: %.2f us per k-block (ideal 64 MFMA x 64 cyc @2.4GHz = 1.707 us at MT=4), fixed %.1f us per wave"
      % (slope * 1e3 / rounds / (C / 8.0), icpt * 1e3 / rounds))
