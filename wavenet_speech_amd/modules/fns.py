"""Small tensor helpers with the reference's names (reference modules/fns.py)."""
import torch


def one_hot_encoding(seq, num_indices):
    """(batch, L) int64 -> (batch, num_indices, L) float one-hot  (reference modules/fns.py:6-15)"""
    out = torch.zeros(seq.size(0), num_indices, seq.size(1), device=seq.device)
    return out.scatter_(1, seq.unsqueeze(1), 1.)
