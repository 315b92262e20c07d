"""HIP-graph capture of a training step (wavenet_speech_amd.graphs.GraphedStep): replayed steps must do exactly what eager
steps do -- the same kernels through the same C ABI -- so parameters after N steps are bitwise equal."""
import copy

import pytest
import torch

import wavenet_speech_amd as W
from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
from wavenet_speech_amd.modules.wavenet import WaveNet
from wavenet_speech_amd.parallel import FlatGradAllReduce

pytestmark = pytest.mark.gpu


def _train(net, x, cot, steps, graphed):
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
    sync = FlatGradAllReduce(net.parameters())
    losses = []
    if graphed:
        xs, cs = x.clone(), cot.clone()
        g = W.GraphedStep(lambda: (net(xs) * cs).sum(), net.parameters(), optimizer=opt, sync=sync, warmup=2)
        for _ in range(steps):
            losses.append(float(g()))
        g.check()
        return losses, 2
    for _ in range(steps):
        sync.zero()
        loss = (net(x) * cot).sum()
        loss.backward()
        sync.reduce()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, 0


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16x3"])
def test_graphed_steps_equal_eager_steps(precision):
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    layers = [(32, 32, 2, d) for d in (1, 2, 4, 8)]
    base = WaveNet(32, 2, layers, 32, softmax=False).to(dev)
    x = torch.randn(2, 32, 700, device=dev)
    cot = torch.randn(2, 32, 700, device=dev)
    nets = [copy.deepcopy(base) for _ in range(2)]
    for n in nets:
        W.set_precision(n, precision)
    # the graphed run performs `warm` extra eager steps before capture: give the eager run the same head start
    lg, warm = _train(nets[1], x, cot, 3, graphed=True)
    le, _ = _train(nets[0], x, cot, 3 + warm, graphed=False)
    assert lg == le[warm:], (lg, le)
    for (k, a), (_, b) in zip(nets[0].state_dict().items(), nets[1].state_dict().items()):
        assert torch.equal(a, b), k


def test_graphed_rawctcnet_new_inputs_each_replay():
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    layers = [(32, 32, 2, d) for d in (1, 2, 4)]
    net = RawCTCNet(32, 3, 5, layers, 32, softmax=False, causal=False).to(dev)
    W.set_precision(net, "bf16")
    ref = copy.deepcopy(net)
    W.set_precision(ref, "bf16")
    xs = torch.zeros(2, 1, 500, device=dev)
    cot = torch.randn(2, 5, 502, device=dev)
    g = W.GraphedStep(lambda: (net(xs) * cot).sum(), net.parameters(), warmup=1)
    for seed in (1, 2):
        x = torch.randn(2, 1, 500, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        xs.copy_(x)
        loss = float(g())
        for p in ref.parameters():
            p.grad = None
        want = (ref(x) * cot).sum()
        want.backward()
        assert loss == float(want.detach())
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            assert (p.grad is None) == (q.grad is None), k
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), k
