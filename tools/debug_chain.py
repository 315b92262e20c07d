#!/usr/bin/env python3
"""Debug: x -> convA(k=1) -> leaky -> convB(k=1) -> loss ; check intermediate grads vs CPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import wavenet_oracle as O
from wavenet_speech_amd import functional as HF
torch.set_num_threads(16)
dev = "cuda:0"
C, L, B = 64, 2000, 2
torch.manual_seed(3)
wA, bA, wB, bB = torch.randn(C, C, 1) * 0.2, torch.randn(C), torch.randn(C, C, 1) * 0.2, torch.randn(C)
x, cot = torch.randn(B, C, L), torch.randn(B, C, L)
def run(conv, dev_):
    ps = [t.clone().to(dev_).requires_grad_(True) for t in (wA, bA, wB, bB)]
    xg = x.clone().to(dev_).requires_grad_(True)
    z1 = conv(xg, ps[0], ps[1]); z1.retain_grad()
    a1 = F.leaky_relu(z1, 0.01); a1.retain_grad()
    z2 = conv(a1, ps[2], ps[3])
    (z2 * cot.to(dev_)).sum().backward()
    return dict(z2=z2.detach(), da1=a1.grad, dz1=z1.grad, dx=xg.grad, dwA=ps[0].grad, dbA=ps[1].grad, dwB=ps[2].grad, dbB=ps[3].grad)
ref = run(lambda t, w, b: O.dilated_conv(t, w, b, 1, True, "aten"), "cpu")
got = run(lambda t, w, b: HF.dilated_conv(t, w, b, 1, True), dev)
for k in ref:
    print("%-5s err %.2e" % (k, O.rel_err(got[k].cpu(), ref[k])))
