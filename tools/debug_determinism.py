#!/usr/bin/env python3
"""Debug: run one residual block fwd+bwd twice at full scale and report which tensors differ bitwise / vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
from wavenet_speech_amd.modules.block import ResidualBlock
torch.set_num_threads(16)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
d = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = "cuda:0"
torch.manual_seed(0)
blk = ResidualBlock(C, C, 2, d)
sd = {k: v.clone() for k, v in blk.state_dict().items()}
x = torch.randn(B, C, L); cr = torch.randn(B, C, L); cs = torch.randn(B, C, L)
blk = blk.to(dev)
def run():
    blk.zero_grad(set_to_none=True)
    xg = x.to(dev).requires_grad_(True)
    r, s = blk(xg)
    ((r * cr.to(dev)).sum() + (s * cs.to(dev)).sum()).backward()
    out = {"r": r.detach().clone(), "s": s.detach().clone(), "dx": xg.grad.clone()}
    out.update({k: p.grad.clone() for k, p in blk.named_parameters()})
    return out
a = run(); b = run(); c = run()
for k in a:
    same_ab = torch.equal(a[k], b[k]); same_ac = torch.equal(a[k], c[k])
    md = float((a[k] - b[k]).abs().max()) / (float(a[k].abs().max()) + 1e-30)
    print("%-28s run1==run2 %-5s run1==run3 %-5s  max rel diff %.2e" % (k, same_ab, same_ac, md))
if B <= 2:
    sdl = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    r0, s0 = O.residual_block(xo, sdl, d, True, impl="aten")
    ((r0 * cr).sum() + (s0 * cs).sum()).backward()
    ref = {"r": r0, "s": s0, "dx": xo.grad}; ref.update({k: v.grad for k, v in sdl.items()})
    for k in a:
        print("%-28s vs oracle %.2e" % (k, O.rel_err(a[k].cpu(), ref[k])))
