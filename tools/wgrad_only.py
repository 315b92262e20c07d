#!/usr/bin/env python3
"""Run a few residual-block backward passes (for rocprofv3 --pmc on wgrad_kernel).  Usage: wgrad_only.py [libname]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavenet_speech_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(ROOT, "wavenet_speech_amd", sys.argv[1])
from wavenet_speech_amd.modules.block import ResidualBlock
dev = "cuda:0"
blk = ResidualBlock(256, 256, 2, 64).to(dev)
x = torch.randn(16, 256, 16000, device=dev, requires_grad=True)
for _ in range(3):
    r, s = blk(x); (r.sum() + s.sum()).backward()
torch.cuda.synchronize()
