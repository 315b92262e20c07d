#!/usr/bin/env python3
"""Debug: (1) accuracy/determinism of torch's (MIOpen) 1x1 Conv1d stack on the GPU vs CPU at [1,256,16000];
(2) accuracy of the HIP residual stack alone (no torch convs) vs the oracle at 30 blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
torch.set_num_threads(16)
dev = "cuda:0"
C, L = 256, 16000
torch.manual_seed(0)
stack = torch.nn.Sequential(torch.nn.LeakyReLU(0.01), torch.nn.Conv1d(C, C, 1), torch.nn.LeakyReLU(0.01), torch.nn.Conv1d(C, C, 1))
x = torch.randn(1, C, L); cot = torch.randn(1, C, L)
xc = x.clone().requires_grad_(True)
(stack(xc) * cot).sum().backward()
ref = {"dx": xc.grad.clone(), **{k: p.grad.clone() for k, p in stack.named_parameters()}}
stack.zero_grad(set_to_none=True)
stack = stack.to(dev)
res = []
for it in range(2):
    stack.zero_grad(set_to_none=True)
    xg = x.to(dev).requires_grad_(True)
    y = stack(xg)
    (y * cot.to(dev)).sum().backward()
    res.append({"dx": xg.grad.clone(), **{k: p.grad.clone() for k, p in stack.named_parameters()}})
for k in ref:
    print("torch/MIOpen output stack %-10s vs CPU %.2e   run1==run2 %s" % (k, O.rel_err(res[0][k].cpu(), ref[k]), torch.equal(res[0][k], res[1][k])))

# (2) HIP stack alone
from wavenet_speech_amd.modules.wavenet import WaveNet
from wavenet_speech_amd.modules.block import run_stack
layers = [(C, C, 2, 2 ** i) for _ in range(3) for i in range(10)]
net = WaveNet(C, 2, layers, C, softmax=False)
with torch.no_grad():
    for blk in net.convolutions:
        blk.residual_proj.weight.copy_(torch.eye(C) + 0.02 * torch.randn(C, C)); blk.conv1x1_residual.weight.mul_(0.3)
sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
x0 = torch.randn(1, C, L); cot = torch.randn(1, C, L)
xo = x0.clone().requires_grad_(True)
_, S0 = O.block_stack(xo, torch.zeros(1, C, L), sd, layers, True, impl="aten")
(S0 * cot).sum().backward()
net = net.to(dev)
outs = []
for it in range(2):
    net.zero_grad(set_to_none=True)
    xg = x0.to(dev).requires_grad_(True)
    S1 = run_stack(xg, net.convolutions, net.bottlenecks)
    (S1 * cot.to(dev)).sum().backward()
    outs.append((S1.detach().clone(), xg.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
print("HIP stack: skips_sum vs oracle %.2e, dx %.2e" % (O.rel_err(outs[0][0].cpu(), S0), O.rel_err(outs[0][1].cpu(), xo.grad)))
worst = max(((O.rel_err(g.cpu(), sd[k].grad), k) for k, g in outs[0][2].items()))
print("HIP stack: worst parameter gradient vs oracle: %.2e (%s)" % worst)
print("HIP stack deterministic:", torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and all(torch.equal(outs[0][2][k], outs[1][2][k]) for k in outs[0][2]))
