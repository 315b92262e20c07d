"""GPU: behaviours the round-1 review asked to be pinned down -- the inference branch really runs under no_grad, the
last block's unused residual path gets no gradient, mis-chained stacks and bad NLL targets are errors (not silent
out-of-bounds reads), the lease pool never hands a buffer to another stream, bench.py refuses to run on fewer GPUs
than requested, and the reference's one CTC known answer holds on the device."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(c=32, nblk=4, seed=3, softmax=False):
    from wavenet_speech_amd.modules.wavenet import WaveNet
    torch.manual_seed(seed)
    layers = [(c, c, 2, 2 ** i) for i in range(nblk)]
    net = WaveNet(16, 2, layers, c, softmax=softmax).to(DEV)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return net


def test_no_grad_takes_the_inference_branch_and_reuses_packed_weights():
    """needs_input_grad is True for nn.Parameters even under no_grad: grad mode must be captured outside the Function.
    The inference branch keeps ONE z scratch buffer and saves nothing, so it must lease far fewer series buffers."""
    from wavenet_speech_amd import series
    net = _net(nblk=8)
    x = torch.randn(2, 16, 700, device=DEV)
    made = []
    orig = series.Lease.__init__

    def counting(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    series.Lease.__init__ = counting
    try:
        y_train = net(x)
        n_train = len(made)
        del made[:]
        import wavenet_speech_amd as W
        W.freeze_for_inference(net)          # keeping packed weights across forwards is opt-in (p.data updates are invisible)
        with torch.no_grad():
            y_eval = net(x)
            n_eval = len(made)
            hits0 = net.stack_state.cache.hits
            y_eval2 = net(x)
            assert net.stack_state.cache.hits == hits0 + 8, "second no_grad forward must reuse all 8 packed blocks"
    finally:
        series.Lease.__init__ = orig
    # training: x + (ta, sg, z, r) per block; inference: x + one z + r per block
    assert n_eval < 0.6 * n_train, (n_eval, n_train)
    assert O.rel_err(y_eval.cpu(), y_train.detach().cpu()) < 1e-6   # per-block accumulation vs one long-K product
    assert torch.equal(y_eval, y_eval2)
    # an in-place update invalidates the cache
    with torch.no_grad():
        net.convolutions[0].conv_tanh.conv1d.weight.mul_(1.5)
        y3 = net(x)
    assert not torch.equal(y3, y_eval)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    layers = net.layers
    assert O.rel_err(y3.cpu(), O.wavenet(x.cpu(), sd, layers, False)) < 1e-4


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16x3"])
def test_data_updates_are_seen_by_unfrozen_inference(precision):
    """p.data.mul_() bumps no autograd version: by default every forward packs the current weights, so the
    output must follow; only freeze_for_inference keeps packed weights."""
    import wavenet_speech_amd as W
    net = _net(nblk=4)
    W.set_precision(net, precision)
    x = torch.randn(2, 16, 300, device=DEV)
    with torch.no_grad():
        y0 = net(x)
        y0b = net(x)
        net.convolutions[1].conv_sigmoid.conv1d.weight.data.mul_(2.0)
        net.bottlenecks[2].weight.data.mul_(0.5)
        y1 = net(x)
        net.convolutions[1].conv_sigmoid.conv1d.weight.data.mul_(0.5)
        net.bottlenecks[2].weight.data.mul_(2.0)
        y2 = net(x)
    assert torch.equal(y0, y0b)
    assert not torch.equal(y0, y1)
    assert torch.equal(y0, y2)


def test_last_block_residual_path_gets_no_gradient():
    """the reference's autograd leaves conv1x1_residual / residual_proj of the LAST block without a gradient (their
    output is unused, modules/wavenet.py:99): None here too, not zeros (weight decay would otherwise move them)."""
    net = _net()
    x = torch.randn(2, 16, 300, device=DEV)
    net(x).sum().backward()
    last = net.convolutions[-1]
    assert last.conv1x1_residual.weight.grad is None and last.conv1x1_residual.bias.grad is None
    assert last.residual_proj.weight.grad is None and last.residual_proj.bias.grad is None
    assert last.conv_tanh.conv1d.weight.grad is not None and last.conv1x1_skip.weight.grad is not None
    first = net.convolutions[0]
    assert first.conv1x1_residual.weight.grad is not None and first.residual_proj.weight.grad is not None
    # the data-parallel wrapper keeps them None too (only their slice of the flat buffer is zeroed for the all-reduce): a
    # weight-decay optimizer then treats them exactly as on the single-process path
    from wavenet_speech_amd.parallel import FlatGradAllReduce
    sync = FlatGradAllReduce(net.parameters())
    sync.zero()
    net(x).sum().backward()
    sync.reduce()
    assert last.residual_proj.weight.grad is None and last.conv1x1_residual.bias.grad is None
    assert first.residual_proj.weight.grad is not None
    i = [id(p) for p in sync.params].index(id(last.residual_proj.weight))
    assert float(sync.views[i].abs().max()) == 0.0


def test_mis_chained_stack_is_an_error():
    from wavenet_speech_amd.modules.wavenet import WaveNet
    net = WaveNet(16, 2, [(32, 32, 2, 1), (24, 32, 2, 2)], 32, softmax=False).to(DEV)
    with pytest.raises(RuntimeError, match="expects 24 input channels"):
        net(torch.randn(1, 16, 64, device=DEV))


def test_nll_rejects_out_of_range_targets():
    from wavenet_speech_amd import functional as HF
    logits = torch.randn(2, 8, 40, device=DEV)
    tg = torch.randint(0, 8, (2, 40), device=DEV)
    HF.sequence_nll(logits, tg)
    for bad in (8, -1, 1 << 40):
        t2 = tg.clone()
        t2[1, 17] = bad
        with pytest.raises(RuntimeError, match="outside"):
            HF.sequence_nll(logits, t2)


def test_lease_pool_never_crosses_streams():
    from wavenet_speech_amd.series import POOL, Lease, SeriesLayout
    lay = SeriesLayout(333, 4)
    dev = torch.device(DEV)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(s1):
        a = Lease(1, 16, lay, dev)
        pa = a.ptr
        del a                      # back to the pool, possibly with work still queued on s1
    with torch.cuda.stream(s2):
        b = Lease(1, 16, lay, dev)
        assert b.ptr != pa, "a buffer released on one stream was handed to another stream"
    with torch.cuda.stream(s1):
        c = Lease(1, 16, lay, dev)
        assert c.ptr == pa         # same stream: stream order makes reuse safe
    del b, c
    POOL.clear()


def test_bench_refuses_fewer_gpus_than_requested():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single visible GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "only 1 GPU" in out.stderr and not out.stdout.strip()
    # a launcher whose world size disagrees with --gpus is an error too
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_ctc_known_answer_on_the_device():
    """reference tests/test_classifier.py:53-59: warp-ctc on a 2-step, 5-class toy -> 'approximately 2.4628'"""
    from wavenet_speech_amd.training import ctc_total
    acts = torch.tensor([[[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1]]], device=DEV)   # (B=1, T=2, C=5)
    trans = acts.permute(0, 2, 1).contiguous().requires_grad_(True)                             # [B, labels, T]
    labels = torch.tensor([[1, 2]], device=DEV)
    loss = ctc_total(trans, labels, torch.tensor([2], device=DEV))
    assert loss.is_cuda and abs(float(loss) - 2.4628) < 2e-4
    loss.backward()
    assert trans.grad is not None and bool(torch.isfinite(trans.grad).all())
