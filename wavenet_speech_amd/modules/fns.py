"""Tensor helper kept for API compatibility with the reference's modules/fns.py (same name, same contract)."""
import torch
import torch.nn.functional as F


def one_hot_encoding(seq, num_indices):
    """Quantised levels -> one-hot channels: int64 (batch, L) in [0, num_indices) -> float (batch, num_indices, L),
    on the device of `seq` (reference: modules/fns.py:6-15)."""
    return F.one_hot(seq, num_indices).transpose(1, 2).to(torch.float32).contiguous()
