import os, sys, torch
sys.path.insert(0, os.getcwd())
import wavenet_speech_amd as W
from tests.test_gpu_half import _cond_wavenet
DEV="cuda:0"
c, dims, L, B = 256, (1,4), 256, 2
layers = [(c, c, 2, d) for d in dims]
net = _cond_wavenet(c, layers, seed=c + L).to(DEV)
g = torch.Generator().manual_seed(L + B)
x, cot = torch.randn(B, c, L, generator=g).to(DEV), torch.randn(B, c, L, generator=g).to(DEV)
W.set_precision(net, sys.argv[1])
def grads():
    for p in net.parameters(): p.grad=None
    xg=x.clone().requires_grad_(True)
    y=net(xg); (y*cot).sum().backward()
    W.check_device_flags()
    return y.detach(), xg.grad, {k:p.grad.clone() for k,p in net.named_parameters() if p.grad is not None}
y1,dx1,g1=grads()
os.environ["WN_SERIES_HEAD"]="0"
y0,dx0,g0=grads()
print("fwd equal", torch.equal(y1,y0), "dx rel", float((dx1-dx0).abs().max()/dx0.abs().max()))
for k in g0: print(k, "%.2e"%float((g1[k]-g0[k]).abs().max()/g0[k].abs().max()))
if len(sys.argv) > 2:
    from wavenet_speech_amd import functional_half as FH
    FH._grad_scale = lambda cot, mode: (None, None)
    os.environ.pop("WN_SERIES_HEAD")
    y1,dx1,g1=grads()
    os.environ["WN_SERIES_HEAD"]="0"
    y0,dx0,g0=grads()
    print("NO DYN: dx rel", float((dx1-dx0).abs().max()/dx0.abs().max()), "os1w %.2e" % float((g1["output_stack.1.weight"]-g0["output_stack.1.weight"]).abs().max()/g0["output_stack.1.weight"].abs().max()))
