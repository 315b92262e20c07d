import os, sys, torch
sys.path.insert(0, os.getcwd())
import wavenet_speech_amd as W
from wavenet_speech_amd import functional as HF
from wavenet_speech_amd.modules.wavenet import WaveNet
from oracle import wavenet_oracle as O
torch.manual_seed(0)
dev = torch.device("cuda:0")
for C, L, B in ((128, 1000, 3), (96, 517, 2), (40, 300, 2)):
    layers = [(C, C, 2, d) for d in (1, 2, 4, 512)]
    net = WaveNet(C, 2, layers, C, softmax=False).to(dev)
    with torch.no_grad():
        for blk in net.convolutions:
            blk.residual_proj.weight.copy_(torch.eye(C, device=dev) + 0.02 * torch.randn(C, C, device=dev))
    x = torch.randn(B, C, L, device=dev)
    y32 = net(x)
    W.set_precision(net, "bf16")
    HF.profile_reset(); HF.profile_enable(True)
    y = net(x)
    (y * torch.randn_like(y)).sum().backward()
    with torch.no_grad():
        yi = net(x)
    HF.profile_enable(False)
    k = HF.profile_read()
    print(C, L, B, "bf16 vs f32 fwd rel err %.3e  infer vs train %.3e" % (float((y - y32).abs().max() / y32.abs().max()), float((yi - y).abs().max() / y.abs().max())),
          {n: v[1] for n, v in k.items() if v[1]})
