#!/usr/bin/env python3
"""Diagnostic: which torch ops (not the library's kernels) run in one training step, by count and device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from wavenet_speech_amd.modules.wavenet import WaveNet
from wavenet_speech_amd.parallel import FlatGradAllReduce
dev = "cuda:0"
C, L, B = 256, 16000, 16
layers = [(C, C, 2, 2 ** i) for _ in range(3) for i in range(10)]
net = WaveNet(C, 2, layers, C, softmax=False).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4); sync = FlatGradAllReduce(net.parameters())
x = torch.zeros(B, C, L, device=dev).scatter_(1, torch.randint(0, C, (B, 1, L), device=dev), 1.0); cot = torch.randn(B, C, L, device=dev)
def step():
    sync.zero(); out = net(x); (out * cot).sum().backward(); sync.reduce(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, "device_time_total", None)
    if dt is None: dt = getattr(e, "cuda_time_total", 0)
    rows.append((e.key, e.count, dt / 1e3, e.cpu_time_total / 1e3))
rows.sort(key=lambda r: -r[2])
print("%-60s %7s %10s %10s" % ("op", "count", "dev ms", "cpu ms"))
for k, n, d, c in rows[:45]:
    print("%-60s %7d %10.3f %10.3f" % (k[:60], n, d, c))
print("---- CPU-side ops that launch fill kernels")
for e in prof.key_averages(group_by_stack_n=0):
    if any(s in e.key for s in ("zero", "fill", "zeros")):
        print("%-50s count %5d cpu ms %.3f" % (e.key[:50], e.count, e.cpu_time_total / 1e3))
