"""
GPU: the reference's own test scripts (tests/test_conv_ops.py, test_block.py, test_wavenet.py, test_classifier.py are
print-and-eyeball scripts: shape asserts plus "you should see a gradually decreasing loss") replayed on the HIP modules,
with the eyeballing turned into assertions and the oracle as numeric checker.
"""
import pytest
import torch

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _M():
    import wavenet_speech_amd.modules as M
    return M


def _overfit(module, source, target, combine, steps=150, lr=1e-2):
    """the reference's constant-sequence overfit loop (tests/test_block.py:44-57) -- returns (first, last) loss"""
    opt = torch.optim.Adam(module.parameters(), lr=lr)
    loss_fn = torch.nn.MSELoss()
    first = last = None
    for _ in range(steps):
        opt.zero_grad()
        loss = loss_fn(combine(module(source)), target)
        loss.backward()
        opt.step()
        last = float(loss.detach())
        first = last if first is None else first
    return first, last


def test_conv_ops_script():
    """tests/test_conv_ops.py: CausalConv1d keeps the sequence length (k=5, d=3) and can overfit a constant"""
    M = _M()
    torch.manual_seed(0)
    for cls in (M.CausalConv1d, M.NonCausalConv1d):
        conv = cls(4, 5, 5, dilation=3).to(DEV)
        x = torch.randn(3, 4, 12, device=DEV)
        assert conv(x).shape == (3, 5, 12)
        src = torch.full((3, 4, 12), 2.0, device=DEV)
        tgt = torch.full((3, 5, 12), 3.0, device=DEV)
        first, last = _overfit(conv, src, tgt, lambda y: y)
        assert last < 0.2 * first


def test_block_script():
    """tests/test_block.py: 4 -> 5 channels, k=2, d=2, L=12, B=3; causal and non-causal; overfit outs+skips to 3"""
    M = _M()
    torch.manual_seed(0)
    for causal in (True, False):
        blk = M.ResidualBlock(4, 5, 2, 2, causal=causal).to(DEV)
        out, skip = blk(torch.randn(3, 4, 12, device=DEV))
        assert out.shape == (3, 5, 12) and skip.shape == (3, 5, 12)
        src = torch.full((3, 4, 12), 2.0, device=DEV)
        tgt = torch.full((3, 5, 12), 3.0, device=DEV)
        first, last = _overfit(blk, src, tgt, lambda o: o[0] + o[1])
        assert last < 0.1 * first


def test_wavenet_script_40_blocks():
    """tests/test_wavenet.py: 11-dim, 40 blocks (4 cycles of dilation 1..512), L=14, B=5 -- every dilation >= L reads
    only the zero history.  Shape + parity with the oracle (LeakyReLU pattern pinned, see oracle._leaky)."""
    M = _M()
    torch.manual_seed(1)
    layers = [(11, 11, 2, 2 ** (i % 10)) for i in range(40)]
    net = M.WaveNet(11, 2, layers, 11, softmax=True)
    with torch.no_grad():                      # conditioned residual path: 40 kaiming projections amplify rounding noise
        for blk in net.convolutions:
            blk.residual_proj.weight.copy_(torch.eye(11) + 0.05 * torch.randn(11, 11))
            blk.conv1x1_residual.weight.mul_(0.3)
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x, cot = torch.randn(5, 11, 14), torch.randn(5, 11, 14)
    net = net.to(DEV)
    slopes, remove = O.capture_leaky_slopes(net)
    y1 = net(x.to(DEV))
    remove()
    assert y1.shape == (5, 11, 14)
    assert torch.allclose(y1.sum(dim=1), torch.ones(5, 14, device=DEV), atol=1e-5)      # softmax over channels
    (y1 * cot.to(DEV)).sum().backward()
    y0 = O.wavenet(x, sd, layers, True, slopes=slopes)
    (y0 * cot).sum().backward()
    assert O.rel_err(y1.detach().cpu(), y0) < TOL
    for k, p in net.named_parameters():
        if sd[k].grad is not None:
            assert O.rel_err(p.grad.cpu(), sd[k].grad) < TOL, k


def test_classifier_script_shape():
    """tests/test_classifier.py: WaveNetClassifier 256-dim in, 30 blocks, L=10000, pool 3 -> [B, labels, 3333]"""
    M = _M()
    torch.manual_seed(2)
    layers = [(256, 256, 2, 2 ** (i % 10)) for i in range(30)]
    clf = M.WaveNetClassifier(256, 5, layers, 256, pool_kernel_size=3, softmax=False)
    with torch.no_grad():
        for blk in list(clf.convolutions) + [clf.input_block]:
            blk.residual_proj.weight.copy_(torch.eye(256) + 0.02 * torch.randn(256, 256))
            blk.conv1x1_residual.weight.mul_(0.3)
    sd = {k: v.clone() for k, v in clf.state_dict().items()}
    x = torch.randn(1, 256, 10000)
    with torch.no_grad():
        y0 = O.wavenet_classifier(x, sd, layers, 3, 1, False, impl="aten")
        y1 = clf.to(DEV)(x.to(DEV))
    assert y1.shape == (1, 5, 3333)
    assert O.rel_err(y1.cpu(), y0) < TOL
