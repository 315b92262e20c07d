#!/usr/bin/env python3
"""Steady-state time of one series-GEMM launch (the conv-forward instantiation): WARM launches first, then 50 timed.
Usage: conv_time.py taps [C B L]   (env WARM, default 1500: the chip needs ~0.1 s of load to reach its steady clock)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import functional as HF
k = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16; L = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
x = torch.randn(B, C, L, device="cuda:0"); w = torch.randn(C, C, k, device="cuda:0") * 0.05; b = torch.randn(C, device="cuda:0")
with torch.no_grad():
    for _ in range(int(os.environ.get("WARM", "1500"))):
        HF.dilated_conv(x, w, b, 1, True)
    torch.cuda.synchronize()
    HF.profile_reset(); HF.profile_enable(True)
    for _ in range(50):
        HF.dilated_conv(x, w, b, 1, True)
    torch.cuda.synchronize()
    HF.profile_enable(False)
for name, (ms, n, fl) in HF.profile_read().items():
    if n and fl:
        print("k=%d %s: %.4f ms/launch  %.1f TFLOP/s" % (k, name, ms / n, fl / (ms * 1e-3) / 1e12))
