#!/usr/bin/env python3
"""Debug: retain every intermediate grad of stack -> leaky -> convA -> leaky -> convB, HIP vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import wavenet_oracle as O
from wavenet_speech_amd.modules.wavenet import WaveNet
from wavenet_speech_amd.modules.block import run_stack
from wavenet_speech_amd import functional as HF
torch.set_num_threads(16)
dev = "cuda:0"
C, L, B, nb = 64, 2000, 2, 10
torch.manual_seed(7)
layers = [(C, C, 2, 2 ** (i % 10)) for i in range(nb)]
net = WaveNet(C, 2, layers, C, softmax=False)
sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
x = O.one_hot_encoding(torch.randint(0, C, (B, L)), C); cot = torch.randn(B, C, L)
def chain(S, conv, w1, b1, w3, b3, cot_):
    S.retain_grad()
    a0 = F.leaky_relu(S, 0.01); a0.retain_grad()
    z1 = conv(a0, w1, b1); z1.retain_grad()
    a1 = F.leaky_relu(z1, 0.01); a1.retain_grad()
    z2 = conv(a1, w3, b3)
    (z2 * cot_).sum().backward()
    return dict(S=S.detach(), a0=a0.detach(), z1=z1.detach(), z2=z2.detach(), dS=S.grad, da0=a0.grad, dz1=z1.grad, da1=a1.grad)
out0 = O.dilated_conv(x, sd["entry_conv1d.conv1d.weight"], sd["entry_conv1d.conv1d.bias"], 1, True, "aten")
_, S0 = O.block_stack(out0, torch.zeros(B, C, L), sd, layers, True, "aten")
ref = chain(S0, lambda t, w, b: O.dilated_conv(t, w, b, 1, True, "aten"), sd["output_stack.1.weight"], sd["output_stack.1.bias"], sd["output_stack.3.weight"], sd["output_stack.3.bias"], cot)
net = net.to(dev)
S1 = run_stack(net.entry_conv1d(x.to(dev)), net.convolutions, net.bottlenecks)
got = chain(S1, lambda t, w, b: HF.dilated_conv(t, w, b, 1, True), net.output_stack[1].weight, net.output_stack[1].bias, net.output_stack[3].weight, net.output_stack[3].bias, cot.to(dev))
for k in ref:
    print("%-4s err %.2e   (|ref| max %.3g)" % (k, O.rel_err(got[k].cpu(), ref[k]), float(ref[k].abs().max())))
# same chain but fed with a detached copy of S1 (no stack function upstream)
S1d = S1.detach().clone().requires_grad_(True)
got2 = chain(S1d, lambda t, w, b: HF.dilated_conv(t, w, b, 1, True), net.output_stack[1].weight, net.output_stack[1].bias, net.output_stack[3].weight, net.output_stack[3].bias, cot.to(dev))
print("detached-input chain: dS err %.2e da0 err %.2e" % (O.rel_err(got2["dS"].cpu(), ref["dS"]), O.rel_err(got2["da0"].cpu(), ref["da0"])))
# where is the error?
e = (got["dS"].cpu() - ref["dS"]).abs()
print("dS abs err by batch:", e.amax(dim=(1, 2)).tolist())
print("dS abs err by time quarter:", [float(e[:, :, i * 500:(i + 1) * 500].max()) for i in range(4)])
print("fraction of elements with err > 1e-3:", float((e > 1e-3).float().mean()))
