"""
Drop-in for the reference's modules/classifier.py::WaveNetClassifier (AvgPool1d down-sampling followed by the
same non-causal residual stack); consumes WaveNet's output distribution in the joint NLL+CTC model.
"""
import torch.nn as nn
import torch.nn.functional as F

from .block import ResidualBlock, StackState, run_stack, pointwise_precision, head_precision
from .pointwise import run_sequential


class WaveNetClassifier(nn.Module):
    def __init__(self, in_dim, num_labels, layers, out_dim, pool_kernel_size=2, input_kernel_size=2,
                 input_dilation=1, softmax=True):
        super(WaveNetClassifier, self).__init__()
        self.in_dim, self.num_labels = in_dim, num_labels
        self.layers, self.num_layers, self.out_dim = layers, len(layers), out_dim
        self.pool_kernel_size, self.pool_padding = pool_kernel_size, 0
        self.input_kernel_size, self.input_dilation = input_kernel_size, input_dilation
        self.softmax = softmax
        self.stack_state = StackState()

        self.mean_pool = nn.AvgPool1d(kernel_size=pool_kernel_size, padding=self.pool_padding)
        self.input_block = ResidualBlock(in_dim, layers[0][0], input_kernel_size, input_dilation, causal=False)
        self.input_skip_bottleneck = nn.Conv1d(layers[0][0], out_dim, kernel_size=1)
        self.convolutions = nn.ModuleList([ResidualBlock(ci, co, k, d, causal=False) for (ci, co, k, d) in layers])
        self.bottlenecks = nn.ModuleList([nn.Conv1d(co, out_dim, kernel_size=1) for (_ci, co, _k, _d) in layers])
        self.output_block = nn.Sequential(nn.LeakyReLU(0.01), nn.Conv1d(out_dim, out_dim, kernel_size=1),
                                          nn.LeakyReLU(0.01), nn.Conv1d(out_dim, num_labels, kernel_size=1))
        # reference modules/classifier.py:77-88 (the bottleneck identity-init is dead code there too: 3-D weights)
        for mod in (self.input_block, self.convolutions, self.output_block):
            for p in mod.parameters():
                if p.dim() > 1:
                    nn.init.kaiming_uniform_(p)
                else:
                    p.data.zero_()
        for p in self.bottlenecks.parameters():
            if p.dim() == 1:
                p.data.zero_()

    def forward(self, seq):
        # mean_pool (reference modules/classifier.py:53,102: AvgPool1d, no padding, the tail that does not fill a window dropped)
        # is fused into the load of the stack's input series and its backward into the scatter of the input gradient; the
        # nn.AvgPool1d member stays for the reference's attribute surface.  (WN_POOL_FUSED=0: the torch op, then a plain load.)
        import os
        k = self.mean_pool.kernel_size if isinstance(self.mean_pool.kernel_size, int) else self.mean_pool.kernel_size[0]
        fused = os.environ.get("WN_POOL_FUSED", "1") != "0" and seq.is_cuda
        out = seq if fused else self.mean_pool(seq)
        skips_sum = run_stack(out, [self.input_block] + list(self.convolutions),
                              [self.input_skip_bottleneck] + list(self.bottlenecks), self.stack_state, head=self.output_block,
                              pool=k if fused else 1)
        skips_sum, done = skips_sum
        logit_seq = skips_sum if done else run_sequential(self.output_block, skips_sum, head_precision(self.stack_state.precision))
        if not self.softmax:
            return logit_seq
        return F.softmax(logit_seq, dim=1)
