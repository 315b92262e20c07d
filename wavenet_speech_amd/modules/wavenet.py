"""
Drop-in for the reference's modules/wavenet.py::WaveNet: same constructor, attributes, parameter names
(`entry_conv1d.conv1d.*`, `convolutions.{l}.*`, `bottlenecks.{l}.*`, `output_stack.{1,3}.*`), same init rules.
The layer loop runs as ONE fused autograd function over the HIP library.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as HF
from .block import ResidualBlock, StackState, run_stack, pointwise_precision, head_precision
from .conv_ops import CausalConv1d
from .pointwise import run_sequential


def _init_weights_and_zero_bias(params):
    for p in params:
        if p.dim() > 1:
            nn.init.kaiming_uniform_(p)
        else:
            p.data.zero_()


class WaveNet(nn.Module):
    def __init__(self, in_dim, entry_kwidth, layers, out_dim, softmax=True):
        super(WaveNet, self).__init__()
        self.in_dim, self.entry_kwidth = in_dim, entry_kwidth
        self.layers, self.num_layers = layers, len(layers)
        self.out_dim, self.softmax = out_dim, softmax
        self.stack_state = StackState()   # arithmetic mode + inference weight cache (not a parameter/buffer)

        self.entry_conv1d = CausalConv1d(in_dim, layers[0][0], entry_kwidth, dilation=1)
        self.convolutions = nn.ModuleList([ResidualBlock(ci, co, k, d) for (ci, co, k, d) in layers])
        self.bottlenecks = nn.ModuleList([nn.Conv1d(co, out_dim, 1) for (_ci, co, _k, _d) in layers])
        self.output_stack = nn.Sequential(nn.LeakyReLU(0.01), nn.Conv1d(out_dim, out_dim, kernel_size=1),
                                          nn.LeakyReLU(0.01), nn.Conv1d(out_dim, out_dim, kernel_size=1))
        # reference modules/wavenet.py:74-85.  NB the reference's identity-init of the bottlenecks tests
        # `len(p.size()) == 2`, which a 3-D Conv1d weight never satisfies: bottleneck weights therefore keep
        # PyTorch's default init and only their biases are zeroed.  Reproduced as is.
        _init_weights_and_zero_bias(self.entry_conv1d.parameters())
        _init_weights_and_zero_bias(self.convolutions.parameters())
        for p in self.bottlenecks.parameters():
            if p.dim() == 1:
                p.data.zero_()
        _init_weights_and_zero_bias(self.output_stack.parameters())

    def forward_levels(self, levels):
        """forward(one_hot(levels)) for quantised input levels [B, L] (int64 in [0, in_dim)): the entry conv becomes a gather
        of weight columns and the dense one-hot (262 MB at 16 x 256 x 16000) is never built.  The reference builds the
        one-hot with modules/fns.py:6-15 and multiplies it (modules/wavenet.py:54,93); nn.Module.forward is unchanged."""
        if self.entry_conv1d.dilation != 1:
            raise RuntimeError("wavenet_speech_amd: forward_levels needs the reference's dilation-1 entry conv")
        out = HF.embed_conv(levels, self.entry_conv1d.conv1d.weight, self.entry_conv1d.conv1d.bias)
        return self._after_entry(out)

    def forward(self, signal):
        out = self.entry_conv1d(signal)
        return self._after_entry(out)

    def _after_entry(self, out):
        skips_sum, done = run_stack(out, self.convolutions, self.bottlenecks, self.stack_state, head=self.output_stack)
        output_seq = skips_sum if done else run_sequential(self.output_stack, skips_sum, head_precision(self.stack_state.precision))
        if not self.softmax:
            return output_seq
        return F.softmax(output_seq, dim=1)  # the reference's reshape_in/softmax/reshape_out == softmax over channels
