"""
Drop-in for the reference's modules/raw_ctcnet.py::RawCTCNet (raw 1-channel signal -> label logits).
Same constructor/attributes/parameter names; input block + residual stack run in the fused HIP path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .block import ResidualBlock, StackState, run_stack, pointwise_precision, head_precision, fusable_front
from .pointwise import run_sequential


class RawCTCNet(nn.Module):
    def __init__(self, num_features, feature_kwidth, num_labels, layers, out_dim, input_kernel_size=2,
                 input_dilation=1, positions=False, softmax=True, causal=False):
        super(RawCTCNet, self).__init__()
        self.num_features, self.feature_kwidth, self.num_labels = num_features, feature_kwidth, num_labels
        self.layers, self.num_layers, self.out_dim = layers, len(layers), out_dim
        self.input_kernel_size, self.input_dilation = input_kernel_size, input_dilation
        self.positions, self.softmax, self.causal = positions, softmax, causal
        self.stack_state = StackState()

        # padding = k-1 on both sides: the sequence grows to L + k - 1 (reference modules/raw_ctcnet.py:57-61)
        self.feature_layer = nn.Sequential(
            nn.Conv1d(1, num_features, kernel_size=feature_kwidth, padding=feature_kwidth - 1, dilation=1),
            nn.LeakyReLU(0.01), nn.Conv1d(num_features, num_features, kernel_size=1), nn.LeakyReLU(0.01))
        if positions:
            self.positions_conv1x1 = nn.Sequential(nn.Conv1d(1, num_features, kernel_size=1), nn.Hardtanh())
        self.input_block = ResidualBlock(num_features, layers[0][0], input_kernel_size, input_dilation, causal=causal)
        self.input_skip_bottleneck = nn.Conv1d(layers[0][0], out_dim, kernel_size=1)
        self.convolutions = nn.ModuleList([ResidualBlock(ci, co, k, d, causal=causal) for (ci, co, k, d) in layers])
        self.bottlenecks = nn.ModuleList([nn.Conv1d(co, out_dim, kernel_size=1) for (_ci, co, _k, _d) in layers])
        self.output_block = nn.Sequential(nn.LeakyReLU(0.01), nn.Conv1d(out_dim, out_dim, kernel_size=1),
                                          nn.LeakyReLU(0.01), nn.Conv1d(out_dim, num_labels, kernel_size=1))
        self._reference_init()

    def _reference_init(self, eps=1e-4):
        """reference modules/raw_ctcnet.py:95-115: kaiming-uniform weights, ~0 noisy biases, identity(+noise)
        bottlenecks and position mixer."""
        def noisy_zero(p):
            p.data.zero_().add_(torch.randn(p.size()).mul_(eps))

        def eye_noise(p):
            nn.init.eye_(p.data.view(p.size(0), p.size(1)))
            p.data.add_(torch.randn(p.size()).mul_(eps))

        groups = [(self.feature_layer, False), (self.input_block, False), (self.convolutions, False),
                  (self.bottlenecks, True), (self.output_block, False)]
        if self.positions:
            groups.insert(0, (self.positions_conv1x1, True))
        for mod, identity in groups:
            for p in mod.parameters():
                if p.dim() > 1:
                    eye_noise(p) if identity else nn.init.kaiming_uniform_(p)
                else:
                    noisy_zero(p)

    def forward(self, seq):
        # half modes: feature layer, stack and output block are ONE function in the series layout (no dense round trips, no
        # separate LeakyReLU passes); the position mixer, a signal that needs a gradient or exact-fp32 entry convs keep the op-by-op form
        front = None if self.positions else fusable_front(self.feature_layer, self.stack_state.precision, seq)
        if front is not None:
            out = seq
        else:
            out = run_sequential(self.feature_layer, seq, pointwise_precision(self.stack_state.precision))
        if self.positions:
            steps = torch.arange(0., out.size(2), device=seq.device).view(1, 1, -1)
            out = out + run_sequential(self.positions_conv1x1, steps, pointwise_precision(self.stack_state.precision))
        skips_sum = run_stack(out, [self.input_block] + list(self.convolutions),
                              [self.input_skip_bottleneck] + list(self.bottlenecks), self.stack_state, head=self.output_block,
                              front=front)
        skips_sum, done = skips_sum
        logit_seq = skips_sum if done else run_sequential(self.output_block, skips_sum, head_precision(self.stack_state.precision))
        if not self.softmax:
            return logit_seq
        return F.softmax(logit_seq, dim=1)
