"""
GPU parity tests (run on a real MI355X with `-m gpu`): the HIP path, called through the C ABI of
libwavenet_amd.so, against (1) the golden vectors captured from the reference modules and (2) the CPU oracle
on seeded inputs.  Tolerance: north_star states 1e-4 relative (max |diff| / max |ref|) for this fp32 path.
"""
import pytest
import torch

from oracle import wavenet_oracle as O
from tests import goldenio

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _mods():
    import wavenet_speech_amd.modules as M
    return M


def _compare_grads(g, module, tol=TOL):
    for k, p in module.named_parameters():
        if not g.hasgrad[k]:
            continue
        assert p.grad is not None, k
        err = O.rel_err(p.grad.cpu(), g.grads[k])
        assert err < tol, (g.name, k, err)


def _run_golden(g, module, tol=TOL):
    missing = module.load_state_dict(g.sd, strict=True)  # reference checkpoints must load unchanged
    assert not missing.missing_keys and not missing.unexpected_keys
    module = module.to(DEV)
    x = g.inputs["x"].to(DEV).requires_grad_(True)
    outs = module(x)
    if not isinstance(outs, tuple):
        outs = (outs,)
    assert len(outs) == len(g.outs)
    loss = 0
    for o, ref, cot in zip(outs, g.outs, g.cots):
        assert tuple(o.shape) == tuple(ref.shape)
        err = O.rel_err(o.detach().cpu(), ref)
        assert err < tol, (g.name, "forward", err)
        loss = loss + (o * cot.to(DEV)).sum()
    loss.backward()
    _compare_grads(g, module, tol)
    err = O.rel_err(x.grad.cpu(), g.grad_inputs["x"])
    assert err < tol, (g.name, "dx", err)


@pytest.mark.parametrize("name", goldenio.names("conv_"))
def test_golden_conv(name):
    g = goldenio.load(name)
    m = g.meta
    M = _mods()
    cls = M.CausalConv1d if m["causal"] else M.NonCausalConv1d
    _run_golden(g, cls(m["cin"], m["cout"], m["k"], dilation=m["d"]))


@pytest.mark.parametrize("name", goldenio.names("block_"))
def test_golden_block(name):
    g = goldenio.load(name)
    m = g.meta
    _run_golden(g, _mods().ResidualBlock(m["cin"], m["cout"], m["k"], m["d"], causal=m["causal"]))


@pytest.mark.parametrize("name", goldenio.names("wavenet_"))
def test_golden_wavenet(name):
    g = goldenio.load(name)
    m = g.meta
    _run_golden(g, _mods().WaveNet(m["in_dim"], m["entry_kwidth"], m["layers"], m["out_dim"], softmax=m["softmax"]))


@pytest.mark.parametrize("name", goldenio.names("rawctc_"))
def test_golden_raw_ctcnet(name):
    g = goldenio.load(name)
    m = g.meta
    net = _mods().RawCTCNet(m["num_features"], m["feature_kwidth"], m["num_labels"], m["layers"], m["out_dim"],
                            input_kernel_size=m["input_kernel_size"], input_dilation=m["input_dilation"],
                            positions=m["positions"], softmax=m["softmax"], causal=m["causal"])
    _run_golden(g, net)


@pytest.mark.parametrize("name", goldenio.names("classifier_"))
def test_golden_classifier(name):
    g = goldenio.load(name)
    m = g.meta
    net = _mods().WaveNetClassifier(m["in_dim"], m["num_labels"], m["layers"], m["out_dim"],
                                    pool_kernel_size=m["pool_kernel_size"], input_kernel_size=m["input_kernel_size"],
                                    input_dilation=m["input_dilation"], softmax=m["softmax"])
    _run_golden(g, net)


# ---------------------------------------------------------------------------------------------------------------
# seeded comparisons against the CPU oracle at sizes it finishes in seconds
# ---------------------------------------------------------------------------------------------------------------
BLOCK_CASES = [
    # ci, co, k, d, causal, L, B
    (64, 64, 2, 1, True, 515, 2),
    (128, 128, 2, 512, True, 1300, 2),     # cfg2 width, largest dilation, L not a multiple of 128
    (256, 256, 2, 64, True, 640, 1),       # cfg3 width
    (256, 256, 2, 512, True, 1024, 1),
    (96, 160, 3, 7, False, 333, 2),        # Ci != Co, k=3 non-causal, odd length
    (33, 65, 2, 5, False, 127, 3),         # awkward channel counts, L < 128
    (8, 8, 2, 2, True, 1, 1),              # single time step
    (300, 260, 2, 3, True, 200, 1),        # > 256 channels: multi-slab / multi-tile paths
]


@pytest.mark.parametrize("case", BLOCK_CASES)
def test_block_vs_oracle(case):
    ci, co, k, d, causal, L, B = case
    torch.manual_seed(hash(case) % 10000)
    blk = _mods().ResidualBlock(ci, co, k, d, causal=causal)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape))
    sd = {kk: v.clone() for kk, v in blk.state_dict().items()}
    x = torch.randn(B, ci, L)
    cr, cs = torch.randn(B, co, L), torch.randn(B, co, L)
    # oracle (CPU)
    sdl = {kk: v.clone().requires_grad_(True) for kk, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    r0, s0 = O.residual_block(xo, sdl, d, causal)
    ((r0 * cr).sum() + (s0 * cs).sum()).backward()
    # HIP
    blk = blk.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    r1, s1 = blk(xg)
    ((r1 * cr.to(DEV)).sum() + (s1 * cs.to(DEV)).sum()).backward()
    assert O.rel_err(r1.detach().cpu(), r0) < TOL
    assert O.rel_err(s1.detach().cpu(), s0) < TOL
    assert O.rel_err(xg.grad.cpu(), xo.grad) < TOL
    for kk, p in blk.named_parameters():
        err = O.rel_err(p.grad.cpu(), sdl[kk].grad)
        assert err < TOL, (case, kk, err)


def test_wavenet_vs_oracle_mid_size():
    """64 channels, one full 1..512 dilation cycle, L=2000 (receptive field 1024 < L), batch 2."""
    torch.manual_seed(7)
    layers = [(64, 64, 2, 2 ** i) for i in range(10)]
    net = _mods().WaveNet(64, 2, layers, 64, softmax=False)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    q = torch.randint(0, 64, (2, 2000))
    x = O.one_hot_encoding(q, 64)
    cot = torch.randn(2, 64, 2000)
    net = net.to(DEV)
    slopes, remove = O.capture_leaky_slopes(net)     # replay the HIP model's LeakyReLU pattern in the oracle
    y1 = net(x.to(DEV))
    remove()
    (y1 * cot.to(DEV)).sum().backward()
    sdl = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y0 = O.wavenet(x, sdl, layers, False, impl="aten", slopes=slopes)
    (y0 * cot).sum().backward()
    assert O.rel_err(y1.detach().cpu(), O.wavenet(x, sd, layers, False, impl="aten")) < TOL   # plain forward parity
    assert O.rel_err(y1.detach().cpu(), y0) < TOL
    worst = 0.0
    for k, p in net.named_parameters():
        if sdl[k].grad is None:
            continue
        worst = max(worst, O.rel_err(p.grad.cpu(), sdl[k].grad))
    assert worst < TOL, worst


def test_inference_mode_matches_training_forward():
    torch.manual_seed(3)
    layers = [(32, 32, 2, 2 ** i) for i in range(4)]
    net = _mods().WaveNet(16, 2, layers, 32, softmax=True).to(DEV)
    x = torch.randn(2, 16, 300, device=DEV)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    y_train = net(x)
    with torch.no_grad():
        y_eval = net(x)
        y_eval2 = net(x)
    # training forms skips_sum with one long-K product, inference accumulates it block by block: same math, a
    # different association of the per-block sums and biases
    assert O.rel_err(y_train.detach().cpu(), y_eval.cpu()) < 1e-6
    assert torch.equal(y_eval, y_eval2)


def test_two_forwards_in_flight_do_not_alias():
    """buffers are leased from a pool; a second forward before the first backward must not clobber saved tensors"""
    torch.manual_seed(5)
    blk = _mods().ResidualBlock(16, 16, 2, 4).to(DEV)
    x1 = torch.randn(2, 16, 200, device=DEV, requires_grad=True)
    x2 = torch.randn(2, 16, 200, device=DEV, requires_grad=True)
    r1, s1 = blk(x1)
    r2, s2 = blk(x2)
    (r1.sum() + s1.sum()).backward()
    g_interleaved = x1.grad.clone()
    x1.grad = None
    blk.zero_grad()
    r1b, s1b = blk(x1)
    (r1b.sum() + s1b.sum()).backward()
    assert torch.equal(g_interleaved, x1.grad)
    del r2, s2


def test_errors_are_loud():
    M = _mods()
    blk = M.ResidualBlock(8, 8, 2, 1).to(DEV)
    with pytest.raises(RuntimeError):
        blk(torch.randn(1, 8, 10))                    # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        blk(torch.randn(1, 9, 10, device=DEV))        # wrong channel count
    with pytest.raises(RuntimeError):
        M.CausalConv1d(4, 4, 9).to(DEV)(torch.randn(1, 4, 32, device=DEV))  # kernel_width 9 > WN_MAX_TAPS (8)


def test_dense_degenerate_layout_does_not_alias_pool_buffers():
    """k=1, C%8==0, L%128==0: the padded layout equals the dense one; outputs must still be private copies."""
    import wavenet_speech_amd.functional as HF
    torch.manual_seed(9)
    w = (torch.randn(16, 16, 1) * 0.2).to(DEV).requires_grad_(True)
    b = torch.randn(16).to(DEV).requires_grad_(True)
    x = torch.randn(2, 16, 256)
    cot = torch.randn(2, 16, 256)
    y = HF.dilated_conv(x.to(DEV), w, b, 1, True)
    y_before = y.detach().clone()
    (y * cot.to(DEV)).sum().backward()          # backward recycles pooled buffers of the same shape
    y2 = HF.dilated_conv(torch.randn(2, 16, 256, device=DEV), w, b, 1, True)
    assert torch.equal(y.detach(), y_before)
    ref = O.dilated_conv(x, w.detach().cpu(), b.detach().cpu(), 1, True)
    assert O.rel_err(y.detach().cpu(), ref) < TOL
    del y2


def test_gate_activation_accuracy():
    """the gate epilogue's exp2/rcp based tanh and sigmoid: absolute error vs torch (fp64) over a wide input sweep"""
    M = _mods()
    c = 32
    blk = M.ResidualBlock(c, c, 2, 1).to(DEV)
    with torch.no_grad():
        for p in blk.parameters():
            p.zero_()
        # a = x (identity on tap 1), g = x  -> ta = tanh(x), sg = sigmoid(x); read them back through z = ta*sg via skip = I z
        blk.conv_tanh.conv1d.weight[:, :, 1] = torch.eye(c)
        blk.conv_sigmoid.conv1d.weight[:, :, 1] = torch.eye(c)
        blk.conv1x1_skip.weight[:, :, 0] = torch.eye(c)
    x = torch.cat([torch.linspace(-30, 30, 4096), torch.linspace(-0.3, 0.3, 4096), torch.tensor([0.0, 1e-6, -1e-6, 88.0, -88.0, 0.124999, 0.125001])])
    x = x[: (x.numel() // c) * c].view(1, c, -1).contiguous()
    with torch.no_grad():
        _, s = blk(x.to(DEV))
    ref = (torch.tanh(x.double()) * torch.sigmoid(x.double()))
    err = float((s.cpu().double() - ref).abs().max())
    assert err < 3e-7, err


# ---------------------------------------------------------------------------------------------------------------
# seeded fuzz: many small awkward shapes (1..8 taps, dilation beyond the sequence, 1..70 channels, 1..300 steps)
# ---------------------------------------------------------------------------------------------------------------
def _fuzz_cases(n, seed):
    import random
    rng = random.Random(seed)
    cases = []
    for _ in range(n):
        k = rng.choice([1, 2, 2, 2, 3, 4, 5, 8])
        d = rng.choice([1, 2, 3, 7, 16, 64, 300, 512])
        ci, co = rng.choice([1, 3, 8, 9, 31, 32, 40, 64, 70]), rng.choice([1, 2, 8, 15, 32, 33, 64, 70])
        L = rng.choice([1, 2, 5, 31, 32, 127, 128, 129, 200, 300])
        cases.append((ci, co, k, d, rng.random() < 0.5, L, rng.choice([1, 2, 3])))
    return cases


def test_block_fuzz_vs_oracle():
    worst = (0.0, None)
    for i, (ci, co, k, d, causal, L, B) in enumerate(_fuzz_cases(40, 2024)):
        torch.manual_seed(1000 + i)
        blk = _mods().ResidualBlock(ci, co, k, d, causal=causal)
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape))
        sdl = {kk: v.clone().requires_grad_(True) for kk, v in blk.state_dict().items()}
        x, cr, cs = torch.randn(B, ci, L), torch.randn(B, co, L), torch.randn(B, co, L)
        xo = x.clone().requires_grad_(True)
        r0, s0 = O.residual_block(xo, sdl, d, causal)
        ((r0 * cr).sum() + (s0 * cs).sum()).backward()
        blk = blk.to(DEV)
        xg = x.to(DEV).requires_grad_(True)
        r1, s1 = blk(xg)
        ((r1 * cr.to(DEV)).sum() + (s1 * cs.to(DEV)).sum()).backward()
        errs = {"r": O.rel_err(r1.detach().cpu(), r0), "s": O.rel_err(s1.detach().cpu(), s0),
                "dx": O.rel_err(xg.grad.cpu(), xo.grad)}
        for kk, p in blk.named_parameters():
            ref = sdl[kk].grad
            if float(ref.abs().max()) == 0.0:      # e.g. a tap that only ever sees the zero padding
                assert float(p.grad.abs().max()) == 0.0, ((ci, co, k, d, causal, L, B), kk)
                continue
            errs[kk] = O.rel_err(p.grad.cpu(), ref)
        for name, e in errs.items():
            assert e < TOL, ((ci, co, k, d, causal, L, B), name, e)
            if e > worst[0]:
                worst = (e, ((ci, co, k, d, causal, L, B), name))
    print("worst fuzz error", worst)


def test_stack_fuzz_vs_oracle():
    """Short stacks of blocks with changing widths / taps / dilations plus bottlenecks (the run_stack entry point)."""
    import random
    from wavenet_speech_amd.modules.block import run_stack
    rng = random.Random(77)
    for i in range(12):
        torch.manual_seed(500 + i)
        causal = rng.random() < 0.5
        widths = [rng.choice([5, 8, 24, 40]) for _ in range(rng.choice([2, 3, 4]) + 1)]
        out_dim = rng.choice([3, 8, 20])
        L, B = rng.choice([7, 64, 130, 257]), rng.choice([1, 2])
        layers = [(widths[j], widths[j + 1], rng.choice([2, 2, 3]), rng.choice([1, 2, 4, 9, 200])) for j in range(len(widths) - 1)]
        blocks = torch.nn.ModuleList([_mods().ResidualBlock(ci, co, k, d, causal=causal) for ci, co, k, d in layers])
        botts = torch.nn.ModuleList([torch.nn.Conv1d(co, out_dim, 1) for _ci, co, _k, _d in layers])
        sd = {}
        for j, (b, t) in enumerate(zip(blocks, botts)):
            sd.update({"convolutions.%d.%s" % (j, n): v.clone().requires_grad_(True) for n, v in b.state_dict().items()})
            sd.update({"bottlenecks.%d.%s" % (j, n): v.clone().requires_grad_(True) for n, v in t.state_dict().items()})
        x, cot = torch.randn(B, widths[0], L), torch.randn(B, out_dim, L)
        xo = x.clone().requires_grad_(True)
        _, S0 = O.block_stack(xo, torch.zeros(B, out_dim, L), sd, layers, causal)
        (S0 * cot).sum().backward()
        blocks, botts = blocks.to(DEV), botts.to(DEV)
        xg = x.to(DEV).requires_grad_(True)
        S1 = run_stack(xg, blocks, botts)
        (S1 * cot.to(DEV)).sum().backward()
        assert O.rel_err(S1.detach().cpu(), S0) < TOL, (layers, "S")
        assert O.rel_err(xg.grad.cpu(), xo.grad) < TOL, (layers, "dx")
        named = [("convolutions.%d.%s" % (j, n), p) for j, b in enumerate(blocks) for n, p in b.named_parameters()]
        named += [("bottlenecks.%d.%s" % (j, n), p) for j, t in enumerate(botts) for n, p in t.named_parameters()]
        for key, p in named:
            ref = sd[key].grad
            if ref is None or float(ref.abs().max()) == 0.0:   # the last block's residual branch feeds nothing
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, (layers, key)
                continue
            e = O.rel_err(p.grad.cpu(), ref)
            assert e < TOL, (layers, key, e)
