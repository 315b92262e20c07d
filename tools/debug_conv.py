#!/usr/bin/env python3
"""Debug: stand-alone HIP dilated conv fwd/bwd vs oracle at several sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
from wavenet_speech_amd import functional as HF
torch.set_num_threads(16)
dev = "cuda:0"
for (ci, co, k, d, causal, L, B) in [(64, 64, 1, 1, True, 2000, 2), (64, 64, 2, 1, True, 2000, 2), (512, 512, 2, 1, True, 4800, 1),
                                      (256, 256, 1, 1, True, 16000, 1), (256, 256, 1, 1, True, 1000, 1), (256, 256, 1, 1, True, 4000, 1),
                                      (128, 128, 1, 1, True, 4000, 1), (256, 256, 2, 64, True, 4000, 2)]:
    torch.manual_seed(1)
    w = torch.randn(co, ci, k) * 0.1; b = torch.randn(co); x = torch.randn(B, ci, L); cot = torch.randn(B, co, L)
    wl, bl, xl = w.clone().requires_grad_(True), b.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y0 = O.dilated_conv(xl, wl, bl, d, causal, impl="aten"); (y0 * cot).sum().backward()
    wg, bg, xg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), x.to(dev).requires_grad_(True)
    y1 = HF.dilated_conv(xg, wg, bg, d, causal); (y1 * cot.to(dev)).sum().backward()
    print("ci=%d co=%d k=%d d=%d L=%d B=%d: y %.1e dx %.1e dw %.1e db %.1e" % (ci, co, k, d, L, B, O.rel_err(y1.detach().cpu(), y0),
          O.rel_err(xg.grad.cpu(), xl.grad), O.rel_err(wg.grad.cpu(), wl.grad), O.rel_err(bg.grad.cpu(), bl.grad)), flush=True)
