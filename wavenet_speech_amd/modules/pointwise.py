"""
The small nn.Sequential stacks around the residual stack (output_stack / output_block / feature_layer /
positions_conv1x1 in the reference: modules/wavenet.py:67-71, raw_ctcnet.py:57-67,89-93, classifier.py:71-75).

They stay nn.Sequential containers of nn.Conv1d / LeakyReLU / Hardtanh so parameter names are unchanged, but their
Conv1d members are evaluated by the HIP series-GEMM kernel (exact fp32 fma chains, bitwise reproducible) instead
of MIOpen, whose weight-gradient kernels at [B,256,16000] are not reproducible run to run (atomic accumulation) and
whose auto-tuning ("find") runs dominate a profile of the first step.  With this the whole model runs on the
library's own kernels apart from elementwise activations.
"""
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as HF


def conv1d_same_as_torch(conv, x, precision="f32"):
    """nn.Conv1d(stride 1, dilation 1, padding p in {0, k-1}) through HF.dilated_conv.
    padding k-1 on both sides (RawCTCNet.feature_layer[0]) lengthens the series to L+k-1:
    y[t] = sum_j W[j] x[t + j - (k-1)], i.e. the causal conv of x followed by k-1 trailing zeros."""
    k = conv.kernel_size[0]
    p = conv.padding[0]
    if conv.stride[0] != 1 or conv.dilation[0] != 1 or conv.groups != 1 or p not in (0, k - 1):
        raise RuntimeError("wavenet_speech_amd: unsupported Conv1d configuration %r" % (conv,))
    if k > 1 and p == k - 1:
        x = F.pad(x, (0, k - 1))
    elif k > 1:
        raise RuntimeError("wavenet_speech_amd: unpadded k>1 Conv1d is not part of the path")
    return HF.dilated_conv(x, conv.weight, conv.bias, 1, True, precision)


def run_sequential(seq, x, precision="f32"):
    """`precision`: the arithmetic mode of the model's residual stack (block.set_precision): its Conv1d members follow it"""
    for mod in seq:
        x = conv1d_same_as_torch(mod, x, precision) if isinstance(mod, nn.Conv1d) else mod(x)
    return x
