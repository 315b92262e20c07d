"""
Synthetic nanopore-like signal, on whatever device the caller asks for: a vectorised restatement
of the reference's on-line generator utils/gaussian_kmer_model.py:

    random nucleotides 1..4 -> sliding 5-mer index sum((nt-1) * [256,64,16,4,1])        (:49, :58-59)
    -> each k-mer held for `upsampling` samples                                          (:62-64)
    -> picoamp sample ~ N(mean[kmer], stdv[kmer])                                         (:67-73)
    -> per-read normalisation (x - mean) / (max - min), mu-law with mu = num_levels,
       digitised against linspace(-1, 1, num_levels)                                      (:36-40, :79-86)
    -> one-hot (num_levels, L)                                                            (:89-97)

The reference reads its 1024-entry mean/stdv table from utils/r9.4_450bps.5mer.template.npz (nanopolish's r9.4 model).
That file is reference data and does not travel; without a table argument a seeded stand-in with the same ranges
(means 59.6-118.5 pA, stdvs 1.34-5.86 pA) is used.  Everything is plain torch ops, so it runs on the GPU.
"""
import math

import torch


def standin_kmer_table(seed=945, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    means = 59.6 + (118.5 - 59.6) * torch.rand(1024, generator=g)
    stdvs = 1.34 + (5.86 - 1.34) * torch.rand(1024, generator=g)
    return means.to(device), stdvs.to(device)


def mu_law(x, mu):
    return torch.sign(x) * torch.log1p(mu * x.abs()) / math.log1p(mu)


def gaussian_kmer_signal(batch, length, num_levels=256, upsampling=3, table=None, generator=None, device="cpu"):
    """Returns (levels [B, L] int64 in 0..num_levels-1, one_hot [B, num_levels, L] float32, bases [B, n] int64 in 1..4)."""
    means, stdvs = table if table is not None else standin_kmer_table(device=device)
    n_kmers = -(-length // upsampling)
    bases = torch.randint(1, 5, (batch, n_kmers + 4), generator=generator).to(device)
    w = torch.tensor([256, 64, 16, 4, 1], device=device)
    windows = (bases - 1).unfold(1, 5, 1)                       # [B, n_kmers, 5]
    kmers = (windows * w).sum(-1)                               # 0..1023
    kmers = kmers.repeat_interleave(upsampling, dim=1)[:, :length]
    noise = torch.randn(batch, length, generator=generator).to(device)
    picoamps = means[kmers] + stdvs[kmers] * noise
    span = picoamps.amax(1, keepdim=True) - picoamps.amin(1, keepdim=True)
    normalised = (picoamps - picoamps.mean(1, keepdim=True)) / span
    mapped = mu_law(normalised, float(num_levels))
    edges = torch.linspace(-1.0, 1.0, num_levels, device=device)
    levels = torch.bucketize(mapped, edges, right=True).clamp_(0, num_levels - 1)   # np.digitize(x, bins)
    one_hot = torch.zeros(batch, num_levels, length, device=device).scatter_(1, levels.unsqueeze(1), 1.0)
    return levels, one_hot, bases[:, 2:-2]
