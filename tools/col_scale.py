#!/usr/bin/env python3
"""How do the column-owner kernels (fused forward, dz, dx) scale with the number of units?  Latency-bound launches show steps
at multiples of the slot count, throughput-bound ones a straight line.   usage: col_scale.py [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import functional as HF
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused import _Stack, _eval

dev = "cuda:0"
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
c, dil, L = 128, (1, 2, 4, 512), 4098
net = _Stack(c, dil, c, False, seed=1).to(dev)
net.stack_state.precision = prec
for B in (2, 4, 8, 12, 16, 20, 24, 28, 31, 32, 33, 40, 48, 64):
    torch.manual_seed(0)
    x = torch.randn(B, c, L, device=dev)
    cot = torch.randn(B, c, L, device=dev)
    for _ in range(3):
        _eval(net, x, cot)
    HF.profile_reset(); HF.profile_enable(True)
    for _ in range(10):
        _eval(net, x, cot)
    HF.profile_enable(False)
    k = HF.profile_read()
    units = (B * L + 31) // 32
    print("B %3d units %5d (%.3f x 2048)  " % (B, units, units / 2048.0) +
          "  ".join("%s %.1f us" % (n.replace("_kernel", "").replace("hgemm<dx>", "hgemm<dx,dense>"), 1e3 * v[0] / v[1]) for n, v in k.items()
                    if v[1] and ("dz" in n or "dx" in n or "fused" in n or "hwgrad" in n)))
