#!/usr/bin/env python3
"""Debug: per-parameter gradient error of the HIP WaveNet vs the oracle (mid-size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
from wavenet_speech_amd.modules.wavenet import WaveNet
torch.set_num_threads(16)
dev = "cuda:0"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.manual_seed(7)
layers = [(C, C, 2, 2 ** (i % 10)) for i in range(nb)]
net = WaveNet(C, 2, layers, C, softmax=False)
sd = {k: v.clone() for k, v in net.state_dict().items()}
x = O.one_hot_encoding(torch.randint(0, C, (B, L)), C); cot = torch.randn(B, C, L)
sdl = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
y0 = O.wavenet(x, sdl, layers, False, impl="aten"); (y0 * cot).sum().backward()
net = net.to(dev)
y1 = net(x.to(dev)); (y1 * cot.to(dev)).sum().backward()
print("forward", O.rel_err(y1.detach().cpu(), y0))
errs = sorted(((O.rel_err(p.grad.cpu(), sdl[k].grad), k, float(sdl[k].grad.abs().max())) for k, p in net.named_parameters() if sdl[k].grad is not None), reverse=True)
for e, k, m in errs[:12]: print("%.2e  %-50s |ref|max %.3g" % (e, k, m))
print("...")
for e, k, m in errs[-3:]: print("%.2e  %-50s |ref|max %.3g" % (e, k, m))
