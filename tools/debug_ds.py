#!/usr/bin/env python3
"""Debug: compare dS (grad wrt skips_sum) and stack grads, HIP vs oracle, feeding the HIP stack with the ORACLE's dS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import wavenet_oracle as O
from wavenet_speech_amd.modules.wavenet import WaveNet
from wavenet_speech_amd.modules.block import run_stack
from wavenet_speech_amd.modules.pointwise import run_sequential
torch.set_num_threads(16)
dev = "cuda:0"
C, L, B, nb = 64, 2000, 2, 10
torch.manual_seed(7)
layers = [(C, C, 2, 2 ** (i % 10)) for i in range(nb)]
net = WaveNet(C, 2, layers, C, softmax=False)
sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
x = O.one_hot_encoding(torch.randint(0, C, (B, L)), C); cot = torch.randn(B, C, L)
# oracle with intermediates
out0 = O.dilated_conv(x, sd["entry_conv1d.conv1d.weight"], sd["entry_conv1d.conv1d.bias"], 1, True, "aten")
_, S0 = O.block_stack(out0, torch.zeros(B, C, L), sd, layers, True, "aten"); S0.retain_grad()
y = F.leaky_relu(S0, 0.01); y = O.conv1x1(y, sd["output_stack.1.weight"], sd["output_stack.1.bias"]); y = F.leaky_relu(y, 0.01)
y0 = O.conv1x1(y, sd["output_stack.3.weight"], sd["output_stack.3.bias"]); (y0 * cot).sum().backward()
net = net.to(dev)
out1 = net.entry_conv1d(x.to(dev))
S1 = run_stack(out1, net.convolutions, net.bottlenecks); S1.retain_grad()
y1 = run_sequential(net.output_stack, S1); (y1 * cot.to(dev)).sum().backward()
print("S fwd err %.2e ; dS err %.2e ; |dS| max %.3g" % (O.rel_err(S1.detach().cpu(), S0), O.rel_err(S1.grad.cpu(), S0.grad), float(S0.grad.abs().max())))
g_model = {k: p.grad.clone() for k, p in net.named_parameters()}
# now feed the oracle's dS into the HIP stack alone
net.zero_grad(set_to_none=True)
out1 = net.entry_conv1d(x.to(dev))
S2 = run_stack(out1, net.convolutions, net.bottlenecks)
S2.backward(S0.grad.to(dev))
for name in ("bottlenecks.2.weight", "convolutions.6.conv_tanh.conv1d.weight", "convolutions.0.conv1x1_skip.weight", "entry_conv1d.conv1d.weight"):
    p = dict(net.named_parameters())[name]
    print("%-45s in-model err %.2e   stack fed with oracle dS err %.2e" % (name, O.rel_err(g_model[name].cpu(), sd[name].grad), O.rel_err(p.grad.cpu(), sd[name].grad)))
