#!/usr/bin/env python3
"""Time the fused NLL head vs torch's cross_entropy at [16,256,16000] and against the HBM roofline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from wavenet_speech_amd import functional as HF
B, C, L = 16, 256, 16000
dev = "cuda:0"
pred = torch.randn(B, C, L, device=dev, requires_grad=True); tg = torch.randint(0, C, (B, L), device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def hip_f(): return HF.sequence_nll(pred, tg)
def hip_fb(): pred.grad = None; HF.sequence_nll(pred, tg).backward()
def th_f(): return F.cross_entropy(pred, tg, reduction="sum") / B
def th_fb(): pred.grad = None; (F.cross_entropy(pred, tg, reduction="sum") / B).backward()
nbytes = B * C * L * 4
for name, f, passes in (("HIP fwd", hip_f, 1), ("torch fwd", th_f, 1), ("HIP fwd+bwd", hip_fb, 3), ("torch fwd+bwd", th_fb, 3)):
    ms = timeit(f)
    print("%-14s %.3f ms   algorithmic %.0f GB/s (%d passes of %.0f MB) = %.1f%% of 8 TB/s" % (name, ms, passes * nbytes / ms / 1e6, passes, nbytes / 1e6, 100 * passes * nbytes / ms / 1e6 / 8000))
