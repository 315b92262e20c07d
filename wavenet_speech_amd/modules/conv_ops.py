"""
Drop-in for the reference's modules/conv_ops.py: same class names, constructor signatures, attributes and
state_dict keys (`conv1d.weight`, `conv1d.bias`); forward runs the HIP series-GEMM kernel instead of
nn.Conv1d + slice.
"""
import torch
import torch.nn as nn

from .. import functional as HF


def autopad(k, d):
    """padding of the reference's non-causal conv (reference modules/conv_ops.py:104-116); odd totals round up"""
    total = (k - 1) * d
    return (total + 1) // 2


def compute_new_length(seq_len, pad, dil, ker):
    """reference modules/conv_ops.py:85-88 (test helper): Conv1d output length"""
    return float(seq_len + 2 * pad - dil * (ker - 1))


def reshape_in(seq):
    """(N, C, L) -> (N*L, C) plus the (N, L) needed to undo it (reference modules/conv_ops.py:91-95)"""
    n, c, l = seq.shape
    return seq.transpose(1, 2).reshape(n * l, c), (n, l)


def reshape_out(seq, dims):
    """inverse of reshape_in (reference modules/conv_ops.py:98-101)"""
    n, l = dims
    return seq.reshape(n, l, -1).transpose(1, 2).contiguous()


class _DilatedConv1d(nn.Module):
    causal = True

    def __init__(self, in_channels, out_channels, kernel_width, dilation=1):
        super(_DilatedConv1d, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_width, self.dilation = kernel_width, dilation
        self.padding = (kernel_width - 1) * dilation if self.causal else autopad(kernel_width, dilation)
        # parameter container only (never called): keeps the reference's `conv1d.weight/bias` names, shapes, init
        self.conv1d = nn.Conv1d(in_channels, out_channels, kernel_width, stride=1, padding=self.padding,
                                dilation=dilation)
        self.receptive_field = kernel_width + (dilation - 1) * (kernel_width - 1)
        self.precision = "f32"   # arithmetic mode, switched together with the model's residual stack by block.set_precision

    def forward(self, seq):
        return HF.dilated_conv(seq, self.conv1d.weight, self.conv1d.bias, self.dilation, self.causal, self.precision)


class CausalConv1d(_DilatedConv1d):
    """y[t] = b + sum_j W[:,:,j] x[t - (k-1-j) d]   (reference modules/conv_ops.py:8-44)"""
    causal = True


class NonCausalConv1d(_DilatedConv1d):
    """y[t] = b + sum_j W[:,:,j] x[t + j d - autopad(k,d)]   (reference modules/conv_ops.py:47-79)"""
    causal = False
