#!/usr/bin/env python3
"""Run the conv-forward series GEMM a few times (for rocprofv3 --pmc).  Usage: gemm_one.py taps [C B L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import functional as HF
k = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16; L = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
x = torch.randn(B, C, L, device="cuda:0"); w = torch.randn(C, C, k, device="cuda:0") * 0.05; b = torch.randn(C, device="cuda:0")
with torch.no_grad():
    for _ in range(4):
        HF.dilated_conv(x, w, b, 1, True)
torch.cuda.synchronize()
