#!/usr/bin/env python3
"""Soak test of the ring kernels' hand-counted waits: the cfg2 model (fused forward, column-owner dz / dx / series convs, group
weight gradients) stepped N times on the same input; every output and gradient must be bitwise equal to the first step's.  A
wait that is one piece too loose shows up here as a rare mismatch long before it shows up in a parity tolerance.
usage: soak_determinism.py [steps] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavenet_speech_amd as W
from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
from wavenet_speech_amd.modules.wavenet import WaveNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = "cuda:0"
torch.manual_seed(0)
bad = 0
for name in ("rawctc", "wavenet96"):
    if name == "rawctc":
        C = 128
        net = RawCTCNet(C, 3, 5, [(C, C, 2, 2 ** i) for i in range(10)], C, softmax=False, causal=False).to(dev)
        x = torch.randn(32, 1, 4096, device=dev)
        cot = torch.randn(32, 5, 4098, device=dev)
    else:
        C = 96
        net = WaveNet(C, 2, [(C, C, 2, 2 ** i) for i in range(8)], C, softmax=False).to(dev)
        x = torch.randn(5, C, 3001, device=dev)
        cot = torch.randn(5, C, 3001, device=dev)
    with torch.no_grad():
        for blk in net.convolutions:       # keep the random-init stack inside fp16's range
            blk.residual_proj.weight.copy_(torch.eye(C, device=dev) + 0.02 * torch.randn(C, C, device=dev))
            blk.conv1x1_residual.weight.mul_(0.3)
    W.set_precision(net, prec)
    ref = None
    t0 = time.time()
    for i in range(steps):
        for p in net.parameters():
            p.grad = None
        y = net(x)
        (y * cot).sum().backward()
        cur = [y.detach()] + [p.grad for p in net.parameters() if p.grad is not None]
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for j, (a, b) in enumerate(zip(cur, ref)):
                if not torch.equal(a, b):
                    bad += 1
                    print("MISMATCH %s step %d tensor %d: max |diff| %.3e" % (name, i, j, float((a.float() - b.float()).abs().max())))
                    break
        if i % 100 == 99:
            print("%s %s: %d steps, %d mismatches, %.1f s" % (name, prec, i + 1, bad, time.time() - t0), flush=True)
W.check_device_flags()
print("soak: %d mismatching steps" % bad)
sys.exit(1 if bad else 0)
