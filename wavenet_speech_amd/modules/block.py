"""
Drop-in for the reference's modules/block.py::ResidualBlock (and GatedActivationUnit): identical constructor,
attributes, parameter names/shapes/registration order; forward = one call into the HIP library.
"""
import torch
import torch.nn as nn

from .. import functional as HF
from .conv_ops import CausalConv1d, NonCausalConv1d


class GatedActivationUnit(nn.Module):
    """tanh(x) * sigmoid(y)  (reference modules/block.py:177-188).  Kept for API compatibility; inside
    ResidualBlock the gate is fused into the dilated-conv kernel's epilogue."""

    def forward(self, x, y):
        return torch.tanh(x) * torch.sigmoid(y)

    def __repr__(self):
        return self.__class__.__name__ + ' ()'


class ResidualBlock(nn.Module):
    """(residual_out, skip_out) = block(seq)   -- reference modules/block.py:15-82.

    a = conv_tanh(seq), g = conv_sigmoid(seq), z = tanh(a) sigmoid(g)
    residual_out = conv1x1_residual(z) + residual_proj(seq)      (the projection is a learned Linear, not identity)
    skip_out     = conv1x1_skip(z)
    """

    def __init__(self, in_channels, out_channels, kernel_width, dilation, causal=True, conditioning=None):
        super(ResidualBlock, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_width, self.dilation = kernel_width, dilation
        self.causal = causal
        self.conditioning = conditioning is not None
        conv = CausalConv1d if causal else NonCausalConv1d
        self.conv_tanh = conv(in_channels, out_channels, kernel_width, dilation=dilation)
        self.conv_sigmoid = conv(in_channels, out_channels, kernel_width, dilation=dilation)
        self.conv1x1_residual = nn.Conv1d(out_channels, out_channels, kernel_size=1)
        self.conv1x1_skip = nn.Conv1d(out_channels, out_channels, kernel_size=1)
        self.gated_activation = GatedActivationUnit()
        self.residual_proj = nn.Linear(in_channels, out_channels)
        self.receptive_field = self.conv_tanh.receptive_field

    def hip_params(self, skip_weight=None, skip_bias=None):
        """the ten tensors in C-ABI order; the skip projection may be replaced by a folded bottleneck*skip pair"""
        return [self.conv_tanh.conv1d.weight, self.conv_tanh.conv1d.bias,
                self.conv_sigmoid.conv1d.weight, self.conv_sigmoid.conv1d.bias,
                self.conv1x1_residual.weight, self.conv1x1_residual.bias,
                self.conv1x1_skip.weight if skip_weight is None else skip_weight,
                self.conv1x1_skip.bias if skip_bias is None else skip_bias,
                self.residual_proj.weight, self.residual_proj.bias]

    def spec(self, skip_rows=None):
        return HF.BlockSpec(self.in_channels, self.out_channels,
                            self.out_channels if skip_rows is None else skip_rows,
                            self.kernel_width, self.dilation, self.causal)

    def forward(self, seq):
        return HF.residual_block(seq, self.spec(), self.hip_params())


def fold_bottleneck(block, bottleneck):
    """bottleneck(conv1x1_skip(z)) = (Wb Wk) z + (Wb bk + bb): two tiny [D,C]x[C,C] torch products per step
    (autograd carries the chain rule back to both parameters), so the stack kernel accumulates skips_sum
    directly and skip_out never touches HBM."""
    wb = bottleneck.weight.squeeze(2)
    wk = block.conv1x1_skip.weight.squeeze(2)
    return wb @ wk, wb @ block.conv1x1_skip.bias + bottleneck.bias


def fold_bottlenecks(blocks, bottlenecks):
    """fold_bottleneck for a whole stack.  When every block has the same width (all reference configs) the folds are
    batched into one bmm / baddbmm: rocBLAS spends ~63 us on each separate 256^3 product (a single workgroup), i.e.
    5.7 ms per step at 30 blocks, against ~0.1 ms batched.  stack/unbind keep the autograd graph free of per-block
    fill/copy kernels."""
    shapes = set((tuple(b.weight.shape), tuple(k.conv1x1_skip.weight.shape)) for k, b in zip(blocks, bottlenecks))
    if len(shapes) != 1:
        folded = [fold_bottleneck(k, b) for k, b in zip(blocks, bottlenecks)]
        return [w for w, _ in folded], [b for _, b in folded]
    wb = torch.stack([b.weight.squeeze(2) for b in bottlenecks])                   # [n, D, C]
    wk = torch.stack([k.conv1x1_skip.weight.squeeze(2) for k in blocks])           # [n, C, C]
    bk = torch.stack([k.conv1x1_skip.bias for k in blocks]).unsqueeze(2)           # [n, C, 1]
    bb = torch.stack([b.bias for b in bottlenecks]).unsqueeze(2)                   # [n, D, 1]
    wf = torch.bmm(wb, wk)
    bf = torch.baddbmm(bb, wb, bk).squeeze(2)
    return list(wf.unbind(0)), list(bf.unbind(0))


class StackState(object):
    """Per-model state of the fused stack path: the arithmetic mode and, for inference, the folded bottlenecks and
    packed weights kept across forwards while no parameter changes (functional.PackCache)."""

    def __init__(self):
        self.precision = "f32"
        self.cache = HF.PackCache()
        self.folded = None


PRECISIONS = ("f32", "f16x3", "f16", "bf16")


def pointwise_precision(precision):
    """the mode of a model's convolutions OUTSIDE its residual stack (entry conv, feature layer, 1x1 convs of the output
    stacks).  In the plain half modes they follow the stack (wn_hconv_*).  In "f16x3" they stay exact fp32: the mode's claim is
    results at the fp32 path's 1e-4 bar on the reference's own outputs and gradients (conditioned models), and those include LeakyReLU kinks --
    a 1e-7 perturbation of the stack's INPUT flips the sign of an element of skips_sum that happens to lie within 1e-6 of
    zero (measured: 1 of 19 200 in the smoke model, 2e-1 max-norm error in that one gradient element, a golden fixture fails),
    while the stack's own f16x3 rounding does not reach its input.  The half convs themselves are exact to 7e-7 in f16x3
    (tests/test_gpu_half_conv.py) and can be called directly: functional.dilated_conv(..., precision="f16x3")."""
    return precision if precision in ("f16", "bf16") else "f32"


def head_precision(precision):
    """the mode of the 1x1 convs AFTER the stack (output_stack / output_block): they follow the stack in every half mode,
    f16x3 included -- their rounding cannot reach skips_sum, the tensor whose LeakyReLU kinks made the f16x3 ENTRY conv fail a
    golden gradient (pointwise_precision); all golden fixtures hold at the 1e-4 bar with it (tests/test_gpu_half.py) and
    the cfg3 step gains 1.6 % (64.2 -> 63.2 ms).  WN_HEAD_F32=1 keeps them exact."""
    import os
    if precision == "f16x3" and not os.environ.get("WN_HEAD_F32"):
        return "f16x3"
    return pointwise_precision(precision)


def set_precision(module, precision):
    """Select the arithmetic of the residual stacks of `module` (a WaveNet / RawCTCNet / WaveNetClassifier or anything
    containing them): "f32" exact fp32 MFMA (default); "f16x3" fp16 MFMA with every operand split into a high and a low
    half (3 products, fp32 accumulate: 22-bit operands, within 1e-4 of the fp32 path on conditioned models, at 3/16 of the fp32 MFMA cost); "f16" / "bf16" plain
    half-precision storage and MFMA with fp32 accumulation (BASELINE configs[4] / configs[1]).  The 1x1 convs of the output
    stacks follow the stack in every half mode, the entry conv / feature layer in the plain half modes only; see
    pointwise_precision and head_precision."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    n = 0
    from .conv_ops import _DilatedConv1d
    for m in module.modules():
        st = getattr(m, "stack_state", None)
        if isinstance(st, StackState):
            st.precision = precision
            st.cache = HF.PackCache()
            n += 1
        if isinstance(m, _DilatedConv1d):      # the entry / feature convs of the model follow its stack (the convs INSIDE a
            m.precision = pointwise_precision(precision)   # ResidualBlock are parameter containers: the stack reads their weights)
    if n == 0:
        raise ValueError("set_precision: no residual stack in %s" % type(module).__name__)
    return module


def freeze_for_inference(module, on=True):
    """Keep the packed weights and the folded bottlenecks of the residual stacks of `module` across no_grad forwards (by
    default every forward packs the current weights again: one launch per stack in the half-precision modes, four small
    launches per block in the fp32 mode).  The cache is emptied when a parameter's autograd version changes (optimizer step,
    load_state_dict, p.mul_()) or a parameter moves; updates written through `p.data` are NOT seen -- call
    freeze_for_inference(module) again after such an update to drop the cache."""
    n = 0
    for m in module.modules():
        st = getattr(m, "stack_state", None)
        if isinstance(st, StackState):
            tables = st.cache.tables
            st.cache = HF.PackCache()
            st.cache.tables = tables
            st.cache.frozen = bool(on)
            st.folded = None
            n += 1
    if n == 0:
        raise ValueError("freeze_for_inference: no residual stack in %s" % type(module).__name__)
    return module


def fusable_head(head, precision):
    """(slopes, parameters) of an output block that can run inside the half-precision stack function, in the series layout:
    exactly LeakyReLU, Conv1d 1x1, LeakyReLU, Conv1d 1x1 (every reference model's output_stack / output_block), in a half mode
    whose head convs follow the stack (head_precision).  None otherwise: the caller then evaluates the block itself."""
    import os
    import torch.nn as nn
    if head is None or os.environ.get("WN_SERIES_HEAD", "1") == "0":
        return None
    if precision == "f32" or head_precision(precision) != precision:
        return None
    mods = list(head)
    if len(mods) != 4 or not (isinstance(mods[0], nn.LeakyReLU) and isinstance(mods[2], nn.LeakyReLU)):
        return None
    for c in (mods[1], mods[3]):
        if not isinstance(c, nn.Conv1d) or c.kernel_size != (1,) or c.stride != (1,) or c.padding != (0,) or c.groups != 1 \
                or c.bias is None:
            return None
    return (mods[0].negative_slope, mods[2].negative_slope), [mods[1].weight, mods[1].bias, mods[3].weight, mods[3].bias]


def fusable_front(front, precision, x):
    """(slopes, parameters) of a feature layer that can run inside the half-precision stack function: exactly Conv1d(1 -> F, k,
    padding k - 1), LeakyReLU, Conv1d 1x1, LeakyReLU (RawCTCNet.feature_layer, reference modules/raw_ctcnet.py:57-61) on a raw signal
    that needs no gradient, in a mode whose entry convs follow the stack (pointwise_precision).  None otherwise."""
    import os
    import torch.nn as nn
    knob = os.environ.get("WN_SERIES_FRONT", "1")
    if front is None or knob == "0":
        return None
    # (WN_SERIES_FRONT=force: also in f16x3, whose entry convs otherwise stay exact fp32 -- the tests hold the fused feature layer
    # to that mode's 1e-4 bar)
    if precision == "f32" or (pointwise_precision(precision) != precision and knob != "force"):
        return None
    if x.requires_grad or x.dim() != 3 or x.shape[1] != 1:
        return None
    mods = list(front)
    if len(mods) != 4 or not (isinstance(mods[1], nn.LeakyReLU) and isinstance(mods[3], nn.LeakyReLU)):
        return None
    c0, c1 = mods[0], mods[2]
    if not (isinstance(c0, nn.Conv1d) and isinstance(c1, nn.Conv1d)) or c0.bias is None or c1.bias is None:
        return None
    k = c0.kernel_size[0]
    if c0.in_channels != 1 or c0.stride != (1,) or c0.dilation != (1,) or c0.groups != 1 or c0.padding != (k - 1,) or k > 8:
        return None
    if c1.kernel_size != (1,) or c1.stride != (1,) or c1.padding != (0,) or c1.groups != 1:
        return None
    return (mods[1].negative_slope, mods[3].negative_slope), [c0.weight, c0.bias, c1.weight, c1.bias]


def run_stack(out, blocks, bottlenecks, state=None, head=None, front=None, pool=1):
    """skips_sum over `blocks` (reference modules/wavenet.py:98-100) through the fused HIP stack path.
    `pool` > 1: AvgPool1d(pool) of `out` fused into the load of the stack's input (WaveNetClassifier.mean_pool).
    `front`: the result of fusable_front() (the feature layer then runs inside the function and `out` is the raw signal).
    With `head` (the model's output block) returns (tensor, head_done): in the half modes the output block runs inside the same
    function, in the series layout (no dense fp32 skips_sum, no separate LeakyReLU passes), and `tensor` is its output."""
    specs, flat = [], []
    out_dim = bottlenecks[0].out_channels
    blocks, bottlenecks = list(blocks), list(bottlenecks)
    precision = state.precision if state is not None else "f32"
    cache = state.cache if state is not None else None   # pack-job tables; for a frozen model also the packed weights
    if state is not None and cache.frozen and not torch.is_grad_enabled():
        # freeze_for_inference: the folds and the packed weights depend only on the parameters, keep them until one of
        # those is updated in place (its _version changes) or moves to other storage
        params = [p for m in blocks + bottlenecks for p in m.parameters()]
        old_key = cache.key
        cache.validate(params, (precision,))
        if state.folded is None or cache.key != old_key:
            state.folded = fold_bottlenecks(blocks, bottlenecks)
        wfs, bfs = state.folded
    else:
        wfs, bfs = fold_bottlenecks(blocks, bottlenecks)
        if state is not None:
            state.folded = None
    for blk, w, b in zip(blocks, wfs, bfs):
        specs.append(blk.spec(out_dim))
        flat.extend(blk.hip_params(w, b))
    fh = fusable_head(head, precision)
    res = HF.residual_stack(out, specs, flat, precision=precision, pack_cache=cache, head=fh, front=front, pool=pool)
    return res if head is None else (res, fh is not None)
