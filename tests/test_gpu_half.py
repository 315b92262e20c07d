"""GPU parity of the half-precision-MFMA modes of the residual stack (wavenet_speech_amd.set_precision).

f16x3 (three-product fp16 split, fp32 accumulate) is held to the SAME 1e-4 bar as the exact-fp32 path, forward and every
gradient, on the golden fixtures and against the oracle.  Plain f16 / bf16 (BASELINE configs[4] / configs[1]) are checked
against the oracle with the error their storage format implies; the measured errors are printed."""
import pytest
import torch

import wavenet_speech_amd as W
from oracle import wavenet_oracle as O
from tests import goldenio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
LOOSE = {"f16": 2e-2, "bf16": 1.2e-1}     # max-norm relative error bounds of the plain modes on the small nets below


def _cond_wavenet(c, layers, in_dim=None, seed=0):
    from wavenet_speech_amd.modules.wavenet import WaveNet
    torch.manual_seed(seed)
    net = WaveNet(in_dim or c, 2, layers, c, softmax=False)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn(p.shape))
        for blk in net.convolutions:
            blk.residual_proj.weight.copy_(torch.eye(blk.out_channels, blk.in_channels)
                                           + 0.02 * torch.randn(blk.out_channels, blk.in_channels))
            blk.conv1x1_residual.weight.mul_(0.3)
    return net


def _run(net, x, cot, layers, precision, tol, replay_slopes=True):
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.to(DEV)
    W.set_precision(net, precision)
    slopes, remove = O.capture_leaky_slopes(net)
    xg = x.to(DEV).requires_grad_(True)
    y1 = net(xg)
    remove()
    (y1 * cot.to(DEV)).sum().backward()
    xr = x.clone().requires_grad_(True)
    y0 = O.wavenet(xr, sd, layers, False, slopes=slopes if replay_slopes else None)
    (y0 * cot).sum().backward()
    errs = {"forward": O.rel_err(y1.detach().cpu(), y0), "dx": O.rel_err(xg.grad.cpu(), xr.grad)}
    for k, p in net.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None, k
            continue
        errs[k] = O.rel_err(p.grad.cpu(), sd[k].grad)
    worst = max(errs, key=errs.get)
    print("%s: forward %.2e, dx %.2e, worst gradient %s %.2e" % (precision, errs["forward"], errs["dx"], worst, errs[worst]))
    assert errs[worst] < tol, (worst, errs[worst])
    return errs


@pytest.mark.parametrize("shape", [(32, 4, 300, 2), (48, 3, 1000, 1), (20, 2, 77, 3), (64, 5, 515, 2)])
def test_f16x3_small_wavenets_vs_oracle(shape):
    c, nblk, L, B = shape
    layers = [(c, c, 2, 2 ** i) for i in range(nblk)]
    net = _cond_wavenet(c, layers, seed=c)
    g = torch.Generator().manual_seed(L)
    x, cot = torch.randn(B, c, L, generator=g), torch.randn(B, c, L, generator=g)
    _run(net, x, cot, layers, "f16x3", TOL)


def test_f16x3_mixed_widths_k3_and_dilation_beyond_length():
    layers = [(24, 40, 3, 1), (40, 40, 2, 7), (40, 72, 2, 300), (72, 72, 3, 2)]
    from wavenet_speech_amd.modules.wavenet import WaveNet
    torch.manual_seed(9)
    net = WaveNet(10, 2, layers, 36, softmax=False)
    g = torch.Generator().manual_seed(10)
    x, cot = torch.randn(2, 10, 260, generator=g), torch.randn(2, 36, 260, generator=g)
    _run(net, x, cot, layers, "f16x3", TOL)


@pytest.mark.parametrize("name", goldenio.names("wavenet_") + goldenio.names("rawctc_") + goldenio.names("classifier_"))
def test_f16x3_golden_models(name):
    """the reference's own outputs and autograd gradients (tests/golden) through the f16x3 stack, same tolerance as fp32"""
    from tests.test_gpu_parity import _mods, _run_golden
    g = goldenio.load(name)
    m = g.meta
    M = _mods()
    if m["kind"] == "wavenet":
        net = M.WaveNet(m["in_dim"], m["entry_kwidth"], m["layers"], m["out_dim"], softmax=m["softmax"])
    elif m["kind"] == "rawctc":
        net = M.RawCTCNet(m["num_features"], m["feature_kwidth"], m["num_labels"], m["layers"], m["out_dim"],
                          input_kernel_size=m["input_kernel_size"], input_dilation=m["input_dilation"],
                          positions=m["positions"], softmax=m["softmax"], causal=m["causal"])
    else:
        net = M.WaveNetClassifier(m["in_dim"], m["num_labels"], m["layers"], m["out_dim"],
                                  pool_kernel_size=m["pool_kernel_size"], input_kernel_size=m["input_kernel_size"],
                                  input_dilation=m["input_dilation"], softmax=m["softmax"])
    W.set_precision(net, "f16x3")
    _run_golden(g, net)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_plain_half_modes_vs_oracle(precision):
    c, L, B = 64, 600, 2
    layers = [(c, c, 2, 2 ** i) for i in range(6)]
    net = _cond_wavenet(c, layers, seed=5)
    g = torch.Generator().manual_seed(6)
    x, cot = torch.randn(B, c, L, generator=g), torch.randn(B, c, L, generator=g)
    errs = _run(net, x, cot, layers, precision, LOOSE[precision])
    print(precision, {k: "%.1e" % v for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:6]})


@pytest.mark.parametrize("precision,c,dims,L,B", [("f16x3", 256, (1, 2, 64), 384, 2), ("f16x3", 128, (1, 3), 256, 3),
                                                  ("f16x3", 256, (3, 1), 400, 2), ("f16x3", 256, (1,), 16, 1),
                                                  ("f16", 256, (1, 4), 256, 2), ("bf16", 256, (2, 1), 384, 1),
                                                  ("f16x3", 512, (1, 130), 128, 1)])
def test_full_tile_shapes_take_the_16x16x32_gemm(precision, c, dims, L, B):
    """rows a multiple of 256 (gate: channels a multiple of 128) and L a multiple of 16 select hgemm8_kernel in the f16x3 mode
    (256 x 256 tiles on v_mfma_f32_16x16x32, 32-channel k-steps, its own weight packing; L = 400 and 16 leave a partial last
    tile): all four epilogues against the oracle.  By default that is the gate and skips_sum GEMMs; test_every_gemm_and_mode_on_the_16x16x32_kernel runs the same
    cases with WN_HGEMM16=2 (every GEMM, every mode) in a child process; WN_HGEMM16=0 keeps every shape on hgemm_kernel."""
    layers = [(c, c, 2, d) for d in dims]
    net = _cond_wavenet(c, layers, seed=c + L)
    g = torch.Generator().manual_seed(L + B)
    x, cot = torch.randn(B, c, L, generator=g), torch.randn(B, c, L, generator=g)
    _run(net, x, cot, layers, precision, TOL if precision == "f16x3" else LOOSE[precision])
    # inference (no saved tanh / sigmoid, per-block skip accumulation on the other kernel) agrees with the training forward
    net = net.to(DEV)
    with torch.no_grad():
        y_eval = net(x.to(DEV))
    y_train = net(x.to(DEV)).detach()
    assert O.rel_err(y_eval.cpu(), y_train.cpu()) < (1e-5 if precision == "f16x3" else 2e-2)


def test_f16x3_two_skips_sum_groups_on_full_tiles():
    """34 blocks = two long-K skips_sum launches (32 + 2 blocks), the second accumulating into the first's output, at a
    shape that runs on hgemm8_kernel"""
    c, L = 256, 128
    layers = [(c, c, 2, 1 + (i % 3)) for i in range(34)]
    net = _cond_wavenet(c, layers, seed=77)
    g = torch.Generator().manual_seed(78)
    x, cot = torch.randn(1, c, L, generator=g), torch.randn(1, c, L, generator=g)
    _run(net, x, cot, layers, "f16x3", TOL)


def test_every_gemm_and_mode_on_the_16x16x32_kernel():
    """by default only the f16x3 gate and skips_sum GEMMs of full-tile shapes run on hgemm8_kernel (measured: the others gain
    nothing or lose); WN_HGEMM16=2 in a child process puts every GEMM of every mode there: all epilogues, all three modes"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WN_HGEMM16="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_half.py"), "-q", "-x", "-k",
                        "full_tile or two_skips_sum"], env=env, cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "8 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


def test_half_inference_matches_training_forward_and_is_deterministic():
    c = 64
    layers = [(c, c, 2, 2 ** i) for i in range(5)]
    net = _cond_wavenet(c, layers, seed=11).to(DEV)
    W.set_precision(net, "f16x3")
    x = torch.randn(2, c, 400, device=DEV)
    y_train = net(x).detach()
    with torch.no_grad():
        y_eval = net(x)
        y_eval2 = net(x)
    assert O.rel_err(y_eval.cpu(), y_train.cpu()) < 1e-5
    assert torch.equal(y_eval, y_eval2)
    W.set_precision(net, "f32")
    with torch.no_grad():
        y32 = net(x)
    assert O.rel_err(y_eval.cpu(), y32.cpu()) < 1e-5


def test_fp16_overflow_is_loud():
    """fp16 holds the residual stream up to 16 * 65504: beyond that the call must fail, not return inf/garbage -- at once
    for an inference call, by the next call into the stack (or check_fp16_overflow()) for a training call, whose flag is
    read without stalling the stream"""
    c = 32
    layers = [(c, c, 2, 1), (c, c, 2, 2)]
    net = _cond_wavenet(c, layers, seed=13).to(DEV)
    W.set_precision(net, "f16x3")
    x = torch.randn(1, c, 100, device=DEV) * 3e6
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="overflow"):
            net(x)
    # training call: the flag travels asynchronously; whichever library call first finds it landed raises (that can be a later
    # kernel launch of the same model call -- the output-stack convs poll too), at the latest check_device_flags()
    with pytest.raises(RuntimeError, match="overflow"):
        net(x)
        W.check_fp16_overflow()
    for _ in range(4):                                   # flags of later launches of the same call may still be in flight
        try:
            W.check_fp16_overflow()
            break
        except RuntimeError:
            pass
    W.check_fp16_overflow()                              # nothing left pending
    W.set_precision(net, "bf16")
    assert bool(torch.isfinite(net(x)).all())


def test_f16x3_cfg3_one_utterance_vs_oracle():
    """the full configs[2] utterance (256 ch x 30 blocks x 16000 steps) at the fp32 path's own 1e-4 bar"""
    from tests.test_gpu_fullsize import _layers, _onehot, _wavenet
    c, L = 256, 16000
    layers = _layers(c, 3)
    net = _wavenet(c, layers)
    x, cot = _onehot(1, c, L, 1)
    errs = _run(net, x, cot, layers, "f16x3", TOL)
    assert errs["forward"] < 2e-5


def test_f16x3_survives_the_reference_init_at_30_blocks():
    """reference init: the residual stream reaches ~7e4 and gradients grow ~2^15 on the way back -- inside the range the
    built-in power-of-two scales leave (no overflow error).  This map is ill-conditioned (two CPU fp32 summation orders
    differ by 4.5e-4 themselves); hi+lo carries 22-23 significant bits against fp32's 24, and the
    amplification multiplies that: the f16x3 result is allowed 10x the CPU fp32 path's own distance from fp64."""
    from tests.test_gpu_fullsize import _layers, _onehot, _wavenet
    c, L = 256, 4000
    layers = _layers(c, 3)
    net = _wavenet(c, layers, seed=7, conditioned=False)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, cot = _onehot(1, c, L, 8)
    with torch.no_grad():
        y64 = O.wavenet(x.double(), {k: v.double() for k, v in sd.items()}, layers, False, impl="taps")
        y_cpu = O.wavenet(x, sd, layers, False, impl="aten")
    net = net.to(DEV)
    W.set_precision(net, "f16x3")
    y = net(x.to(DEV))
    (y * cot.to(DEV)).sum().backward()            # must not raise
    e_cpu, e_hip = O.rel_err(y_cpu.double(), y64), O.rel_err(y.detach().cpu().double(), y64)
    print("reference init, 30 blocks: CPU fp32 vs fp64 %.2e, f16x3 vs fp64 %.2e" % (e_cpu, e_hip))
    assert e_hip < max(TOL, 10.0 * e_cpu)
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)


def test_cfg2_bf16_full_batch():
    """BASELINE configs[1] as stated: RawCTCNet 128 ch x (10 + input) blocks, L=4096, batch 32, bf16.  Size-independent
    properties at the full size plus one utterance against the oracle; the bf16 error is the storage format's (8 bits)."""
    from tests.test_gpu_fullsize import _properties
    from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
    torch.manual_seed(12)
    layers = [(128, 128, 2, 2 ** i) for i in range(10)]
    net = RawCTCNet(128, 3, 5, layers, 128, softmax=False, causal=False).to(DEV)
    W.set_precision(net, "bf16")
    g = torch.Generator().manual_seed(13)
    x = torch.randn(32, 1, 4096, generator=g).to(DEV)
    cot = torch.randn(32, 5, 4098, generator=g).to(DEV)
    y = _properties(net, x, cot, None, 16, causal_prefix=False, additivity_tol=2e-2)
    assert tuple(y.shape) == (32, 5, 4098)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        y0 = O.raw_ctcnet(x[7:8].cpu(), sd, layers, 3, 1, False, False, False, impl="aten")
    err = O.rel_err(y[7:8].cpu(), y0)
    print("cfg2 bf16 forward error vs oracle: %.2e" % err)
    assert err < 0.1


def test_cfg5_f16_full_depth_and_length():
    """BASELINE configs[4]'s shape on one GPU in its stated dtype: WaveNet 512 ch x 60 blocks x L=48000, batch 2, fp16
    (conditioned residual path: 60 random-init blocks amplify by ~2^30, beyond fp16 whatever the kernel)."""
    from tests.test_gpu_fullsize import _layers, _properties, _wavenet
    c, L, B = 512, 48000, 2
    layers = _layers(c, 6)
    net = _wavenet(c, layers, seed=21).to(DEV)
    W.set_precision(net, "f16")
    g = torch.Generator().manual_seed(22)
    x = torch.randn(B, c, L, generator=g).to(DEV)
    cot = torch.randn(B, c, L, generator=g).to(DEV)
    _properties(net, x, cot, 8192, 1, causal_prefix=True, additivity_tol=2e-2)
    # all 60 blocks against the oracle on a shorter utterance
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    xs = x[:1, :, :4000].contiguous()
    with torch.no_grad():
        y1 = net(xs).cpu()
        y0 = O.wavenet(xs.cpu(), sd, layers, False, impl="aten")
    err = O.rel_err(y1, y0)
    print("cfg5 f16 forward error vs oracle (60 blocks): %.2e" % err)
    assert err < 5e-2


def test_f16x3_wide_stack_multi_slab():
    """512 channels: eight 128-row gate slabs, four-slab block GEMMs, two-tile weight gradients"""
    c = 512
    layers = [(c, c, 2, d) for d in (1, 64, 4)]
    net = _cond_wavenet(c, layers, seed=31)
    g = torch.Generator().manual_seed(32)
    x, cot = torch.randn(1, c, 700, generator=g), torch.randn(1, c, 700, generator=g)
    _run(net, x, cot, layers, "f16x3", TOL)


def test_fp16_weight_beyond_range_is_loud():
    """weights are packed as 256 * 16 * w in the fp16 modes, so |w| >= 16 becomes inf and inf * 0 = NaN downstream; the
    store-side checks are written !(|v| <= 65504) so that the NaN trips the overflow flag instead of passing silently"""
    layers = [(32, 32, 2, 1), (32, 32, 2, 2)]
    net = _cond_wavenet(32, layers, seed=3).to(DEV)
    W.set_precision(net, "f16")
    with torch.no_grad():
        net.convolutions[0].conv_tanh.conv1d.weight[3, 5, 1] = 32.0
        with pytest.raises(RuntimeError, match="fp16 overflow"):
            net(torch.randn(1, 32, 200, device=DEV))
            W.check_device_flags()


def test_grad_scale_kernel_matches_the_torch_expression():
    from wavenet_speech_amd import functional_half as FH
    mode = FH._Mode("f16")
    for n, mag in ((1000003, 3.7), (64, 1e-12), (5, 7e9), (4096, 0.0)):
        x = torch.randn(n, device=DEV) * mag
        dyn, inv = FH._grad_scale(x, mode)
        amax = x.abs().amax().clamp_min(1e-30)
        want = torch.exp2(torch.floor(torch.log2(FH.GRAD_TARGET / amax)).clamp(-100.0, 100.0))
        assert float(dyn) == float(want), (n, mag, float(dyn), float(want))
        assert float(inv) == 1.0 / float(want)
