"""The fused block forward (hfused_fwd_kernel: gate -> z -> res [+ skip] in one launch, z in registers between the products;
reference span modules/block.py:65-79) against the two-launch form of the same block (WN_FUSED_FWD=0: gate GEMM, residual GEMM,
z through HBM) and against the oracle; the composite / block-group weight gradients against the per-block form.

The stack is evaluated WITHOUT an output block (run_stack returns skips_sum), so the function under test is smooth: no LeakyReLU
decision can flip between two evaluations and the comparison measures rounding only."""
import copy
import ctypes

import pytest
import torch
import torch.nn as nn

from oracle import wavenet_oracle as O
from wavenet_speech_amd import _lib
from wavenet_speech_amd import functional as HF
from wavenet_speech_amd.modules.block import ResidualBlock, StackState, run_stack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# max-norm relative distance of two evaluations that round to the storage format at different points / in another order
PAIR = {"bf16": 8e-3, "f16": 1e-3}      # observed: 2.7e-3 / 3.7e-4 (worst: dx)
# distance from the fp32 oracle (the bounds of tests/test_gpu_half.py for the plain modes)
LOOSE = {"f16": 2e-2, "bf16": 1.2e-1}


class _Stack(nn.Module):
    def __init__(self, c, dilations, out_dim, causal, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.layers = [(c, c, 2, d) for d in dilations]
        self.causal = causal
        self.convolutions = nn.ModuleList([ResidualBlock(ci, co, k, d, causal=causal) for ci, co, k, d in self.layers])
        self.bottlenecks = nn.ModuleList([nn.Conv1d(co, out_dim, 1) for _ci, co, _k, _d in self.layers])
        self.stack_state = StackState()
        with torch.no_grad():
            for p in self.parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn(p.shape))
            for blk in self.convolutions:       # a conditioned residual path, as in a trained network
                blk.residual_proj.weight.copy_(torch.eye(c) + 0.02 * torch.randn(c, c))
                blk.conv1x1_residual.weight.mul_(0.3)

    def forward(self, x):
        return run_stack(x, self.convolutions, self.bottlenecks, self.stack_state)


def _eval(net, x, cot, training=True):
    for p in net.parameters():
        p.grad = None
    if not training:
        with torch.no_grad():
            return net(x), None, None
    xg = x.clone().requires_grad_(True)
    s = net(xg)
    (s * cot).sum().backward()
    return s.detach(), xg.grad.clone(), {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}


CASES = [  # channels, dilations, out_dim, causal, B, L
    (128, (1, 2, 512), 128, False, 2, 1030),     # cfg2's width, non-causal (RawCTCNet), a dilation beyond half the tile width
    (128, (1, 4), 96, True, 3, 257),             # skip rows != channels
    (96, (2, 8), 96, True, 2, 517),              # three z tiles (NZT = 3)
    (64, (1, 16), 64, False, 1, 1000),           # NZT = 2
    (24, (1, 2), 40, True, 2, 77),               # padded channels (24 -> 32), one ragged unit
    (8, (1,), 8, True, 1, 1),                    # a single time step
    (40, (4, 1), 16, False, 5, 33),              # more utterances than columns per unit
]


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("case", CASES)
def test_fused_forward_equals_two_launch_form_and_oracle(precision, case, monkeypatch):
    c, dil, out_dim, causal, B, L = case
    net = _Stack(c, dil, out_dim, causal, seed=11)
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    torch.manual_seed(3)
    x = torch.randn(B, c, L)
    cot = torch.randn(B, out_dim, L)
    net = net.to(DEV)
    net.stack_state.precision = precision
    other = copy.deepcopy(net)                  # its own pack tables: the two forms pack different weight streams

    monkeypatch.setenv("WN_FUSED_FWD", "1")
    HF.profile_reset()
    HF.profile_enable(True)
    s1, dx1, g1 = _eval(net, x.to(DEV), cot.to(DEV))
    s1i, _, _ = _eval(net, x.to(DEV), cot.to(DEV), training=False)
    HF.profile_enable(False)
    launched = {k: v[1] for k, v in HF.profile_read().items() if v[1]}
    assert launched.get("hfused_fwd_kernel", 0) == 2 * len(dil), launched          # training + inference forward of every block
    assert not any(k.startswith("hgemm_kernel<gate>") or k.startswith("hgemm_kernel<res>") for k in launched), launched

    monkeypatch.setenv("WN_FUSED_FWD", "0")
    HF.profile_reset()
    HF.profile_enable(True)
    s0, dx0, g0 = _eval(other, x.to(DEV), cot.to(DEV))
    HF.profile_enable(False)
    launched = {k: v[1] for k, v in HF.profile_read().items() if v[1]}
    assert "hfused_fwd_kernel" not in launched and launched.get("hgemm_kernel<gate>", 0) == len(dil), launched
    assert not any(k.startswith("hcol_kernel") for k in launched), launched     # (the two-launch form also runs dz / dx on hgemm_kernel)

    # oracle (fp32, CPU)
    xr = x.clone().requires_grad_(True)
    _, s_ref = O.block_stack(xr, torch.zeros(B, out_dim, L), sd, net.layers, causal)
    (s_ref * cot).sum().backward()

    pair, loose = PAIR[precision], LOOSE[precision]
    e_pair = {"forward": O.rel_err(s1.cpu(), s0.cpu()), "inference": O.rel_err(s1i.cpu(), s0.cpu()), "dx": O.rel_err(dx1.cpu(), dx0.cpu())}
    e_ref = {"forward": O.rel_err(s1.cpu(), s_ref.detach()), "inference": O.rel_err(s1i.cpu(), s_ref.detach()),
             "dx": O.rel_err(dx1.cpu(), xr.grad)}
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        ref = sd[k].grad
        if g0[k] is None:
            continue
        e_pair[k] = O.rel_err(g1[k].cpu(), g0[k].cpu())
        if ref is not None:
            e_ref[k] = O.rel_err(g1[k].cpu(), ref)
    wp, wr = max(e_pair, key=e_pair.get), max(e_ref, key=e_ref.get)
    print("%s %s: fused vs two-launch: forward %.2e, worst %s %.2e | vs oracle: forward %.2e, worst %s %.2e"
          % (precision, case, e_pair["forward"], wp, e_pair[wp], e_ref["forward"], wr, e_ref[wr]))
    assert e_pair[wp] <= pair, (wp, e_pair[wp])
    assert e_ref[wr] <= loose, (wr, e_ref[wr])


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_fused_forward_is_deterministic_and_batch_independent(precision, monkeypatch):
    monkeypatch.setenv("WN_FUSED_FWD", "1")
    net = _Stack(128, (1, 2, 4, 64), 128, False, seed=2).to(DEV)
    net.stack_state.precision = precision
    torch.manual_seed(1)
    x = torch.randn(4, 128, 700, device=DEV)
    cot = torch.randn(4, 128, 700, device=DEV)
    a = _eval(net, x, cot)
    b = _eval(net, x, cot)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        if a[2][k] is not None:
            assert torch.equal(a[2][k], b[2][k]), k
    one = _eval(net, x[2:3].contiguous(), cot[2:3].contiguous())
    assert torch.equal(one[0][0], a[0][2])           # an utterance's result does not depend on its neighbours in the batch
    if precision == "bf16":                           # (fp16 carries one dynamic gradient scale per call: another batch, another scale)
        assert torch.equal(one[1][0], a[1][2])


def test_which_blocks_take_the_fused_forward():
    lib = _lib.load()

    def fused(c, out_dim, precision, k=2):
        shape = _lib.BlockShape(2, 300, c, c, out_dim, k, 1, 1, 16 + 512, 8)        # batch, length, Ci, Co, skip rows, k, d, causal, ld, halo
        return lib.wn_hblock_forward_is_fused(ctypes.byref(shape), _lib.PRECISIONS[precision])

    assert fused(128, 128, "bf16") == 1 and fused(8, 8, "f16") == 1 and fused(96, 40, "bf16") == 1
    assert fused(128, 128, "f16x3") == 0        # two planes: the x fragments alone would need 256 registers
    assert fused(256, 256, "bf16") == 0         # a wave cannot own 256 output channels
    assert fused(128, 256, "bf16") == 0         # more than 128 skip rows
    assert fused(64, 64, "bf16", k=3) == 0      # three taps


@pytest.mark.parametrize("precision", ["bf16", "f16", "f16x3"])
def test_block_group_weight_gradients_equal_per_block_form(precision, monkeypatch):
    """wn_hblocks_backward_weights (up to eight blocks per split-K launch, composite 256x256 tiles) against one launch per block
    (WN_WGRAD_GROUP=1) and against separate tiles per gradient (WN_HWGRAD_COMPOSITE=0): the same products summed in the same
    slab order must be bitwise equal within a split plan; plans with other split counts agree to fp32 summation rounding."""
    c = 64 if precision != "f16x3" else 32
    net = _Stack(c, (1, 2, 4, 8, 16), c, True, seed=4).to(DEV)
    net.stack_state.precision = precision
    torch.manual_seed(9)
    x = torch.randn(3, c, 900, device=DEV)
    cot = torch.randn(3, c, 900, device=DEV)
    g_group = _eval(net, x, cot)
    monkeypatch.setenv("WN_WGRAD_GROUP", "1")
    g_single = _eval(copy.deepcopy(net), x, cot)
    monkeypatch.setenv("WN_HWGRAD_COMPOSITE", "0")
    g_plain = _eval(copy.deepcopy(net), x, cot)
    assert torch.equal(g_group[0], g_single[0]) and torch.equal(g_group[1], g_single[1])       # forward and dx do not depend on it
    for k, g in g_group[2].items():
        if g is None:
            assert g_single[2][k] is None and g_plain[2][k] is None, k
            continue
        scale = max(float(g.abs().max()), 1e-30)
        assert float((g - g_single[2][k]).abs().max()) <= 2e-6 * scale, k
        assert float((g - g_plain[2][k]).abs().max()) <= 2e-6 * scale, k


@pytest.mark.parametrize("pair", ["1", "0"])
@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("case", [(128, (1, 2, 512), 128, False, 3, 1030), (128, (4, 1), 128, True, 2, 257), (96, (2, 8, 1), 96, True, 2, 517),
                                  (64, (1, 16), 64, False, 1, 1000), (24, (1, 2), 24, True, 2, 77), (8, (1,), 8, True, 1, 1),
                                  (40, (4, 1), 40, False, 5, 33), (128, (1, 2), 96, True, 2, 300), (120, (2, 1, 4), 100, True, 2, 130)])
def test_column_owner_dz_dx_equal_the_tiled_gemms(precision, case, pair, monkeypatch):
    """hcol_kernel (dz and dx of blocks of <= 128 channels as column-owner streaming kernels, wn_col.hip) against hgemm_kernel on
    the same packed weights (WN_COL_BWD=0): the same products in the same k order with fp32 accumulation; the dgate epilogue
    uses a reciprocal where the tiled kernel divides, so da / dg agree to rounding of the storage format."""
    c, dil, out_dim, causal, B, L = case
    net = _Stack(c, dil, out_dim, causal, seed=21).to(DEV)
    net.stack_state.precision = precision
    other = copy.deepcopy(net)
    torch.manual_seed(4)
    x = torch.randn(B, c, L, device=DEV)
    cot = torch.randn(B, out_dim, L, device=DEV)
    monkeypatch.setenv("WN_COL_BWD", "1")
    monkeypatch.setenv("WN_COL_PAIR", pair)      # 1: dx of a block and dz of the block below it in one launch (hcol2_kernel, wn_col2.hip)
    HF.profile_reset()
    HF.profile_enable(True)
    s1, dx1, g1 = _eval(net, x, cot)
    HF.profile_enable(False)
    launched = {k: v[1] for k, v in HF.profile_read().items() if v[1]}
    same_width = (out_dim + 31) // 32 == (c + 31) // 32
    n = len(dil)
    if same_width and pair == "1" and n > 1:
        # dz of the top block, n - 1 paired launches, the bottom block's dense dx (the stack's input is dense here) on hgemm_kernel
        assert launched.get("hcol_kernel<dz,dgate>", 0) == 1 and launched.get("hcol2_kernel<dx+dz>", 0) == n - 1 and \
            "hcol_kernel<dx>" not in launched and launched.get("hgemm_kernel<dx>", 0) == 1, launched
    elif same_width:
        assert launched.get("hcol_kernel<dz,dgate>", 0) == n and launched.get("hcol_kernel<dx>", 0) == n - 1 and \
            "hcol2_kernel<dx+dz>" not in launched, launched
    else:
        assert not any(k.startswith("hcol_kernel") for k in launched), launched       # a narrower skip path keeps the tiled kernels
    monkeypatch.setenv("WN_COL_BWD", "0")
    HF.profile_reset()
    HF.profile_enable(True)
    s0, dx0, g0 = _eval(other, x, cot)
    HF.profile_enable(False)
    launched = {k: v[1] for k, v in HF.profile_read().items() if v[1]}
    assert not any(k.startswith("hcol_kernel") for k in launched) and launched.get("hgemm_kernel<dz,dgate>", 0) == len(dil), launched
    assert torch.equal(s1, s0)                                   # the forward is the same code
    errs = {"dx": O.rel_err(dx1.cpu(), dx0.cpu())}
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is not None:
            errs[k] = O.rel_err(g1[k].cpu(), g0[k].cpu())
    w = max(errs, key=errs.get)
    print("%s %s: column-owner vs tiled: dx %.2e, worst %s %.2e" % (precision, case, errs["dx"], w, errs[w]))
    assert errs[w] <= PAIR[precision], (w, errs[w])


def test_column_owner_kernels_on_flattened_columns_of_many_short_utterances():
    """units are 32 consecutive valid columns of the flattened (utterance, time) space: utterances shorter than a unit share one"""
    net = _Stack(32, (1, 2), 32, True, seed=3).to(DEV)
    net.stack_state.precision = "bf16"
    torch.manual_seed(8)
    x = torch.randn(37, 32, 5, device=DEV)                       # 185 columns: 6 units, each spanning 6-7 utterances
    cot = torch.randn(37, 32, 5, device=DEV)
    full = _eval(net, x, cot)
    for b in (0, 17, 36):
        one = _eval(net, x[b:b + 1].contiguous(), cot[b:b + 1].contiguous())
        assert torch.equal(one[0][0], full[0][b]) and torch.equal(one[1][0], full[1][b]), b


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("model", ["wavenet", "rawctc"])
def test_series_convs_and_masked_dx_column_owner_equal_tiled(precision, model, monkeypatch):
    """the 1x1 convs of the output block / feature layer (LeakyReLU fused forward, its derivative fused backward) and the first
    block's masked dx as hcol_kernel against hgemm_kernel<HEPI_LEAKY> (WN_COL_BWD=0): the forward is the same arithmetic in the same
    order -> bitwise; the gradients differ only through dz's reciprocal (rounding of the storage format)."""
    import wavenet_speech_amd as W
    from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
    from wavenet_speech_amd.modules.wavenet import WaveNet
    torch.manual_seed(6)
    if model == "wavenet":
        net = WaveNet(64, 2, [(64, 64, 2, d) for d in (1, 2, 4)], 64, softmax=False).to(DEV)
        x = torch.randn(2, 64, 333, device=DEV)
        cot = torch.randn(2, 64, 333, device=DEV)
    else:
        net = RawCTCNet(96, 3, 5, [(96, 96, 2, d) for d in (1, 2, 8)], 96, softmax=False, causal=False).to(DEV)
        x = torch.randn(3, 1, 500, device=DEV)
        cot = torch.randn(3, 5, 502, device=DEV)
    W.set_precision(net, precision)
    other = copy.deepcopy(net)

    def run(n):
        for p in n.parameters():
            p.grad = None
        HF.profile_reset()
        HF.profile_enable(True)
        y = n(x)
        (y * cot).sum().backward()
        HF.profile_enable(False)
        return y.detach(), {k: (None if p.grad is None else p.grad.clone()) for k, p in n.named_parameters()}, \
            {k: v[1] for k, v in HF.profile_read().items() if v[1]}

    monkeypatch.setenv("WN_COL_BWD", "1")
    y1, g1, k1 = run(net)
    monkeypatch.setenv("WN_COL_BWD", "0")
    y0, g0, k0 = run(other)
    assert any(k.startswith("hcol_kernel") for k in k1) and not any(k.startswith("hcol_kernel") for k in k0), (k1, k0)
    assert torch.equal(y1, y0)
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is not None:
            assert O.rel_err(g1[k].cpu(), g0[k].cpu()) <= PAIR[precision], k
