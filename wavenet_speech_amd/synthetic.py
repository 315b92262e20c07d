"""
Synthetic nanopore-like signal, on whatever device the caller asks for: a vectorised restatement
of the reference's on-line generator utils/gaussian_kmer_model.py, stage by stage:

    kmer_indices   nucleotides 1..4 -> 5-mer index sum((nt-1) * [256,64,16,4,1]) of the window starting two bases in
                   (scipy generic_filter with its centred window, then the [4:-4] trim: n-8 k-mers from n bases), each
                   k-mer held for `upsampling` samples                                               (:49, :58-64)
    gaussian_picoamps   sample ~ N(mean[kmer], stdv[kmer])                                             (:67-73)
    quantize       per-read (x - mean) / (max - min), mu-law with mu = num_levels, np.digitize against
                   linspace(-1, 1, num_levels) -- in float64 like the reference                         (:36-40, :79-86)
    one_hot        (num_levels, L) float32                                                            (:89-97)

Parity: the three deterministic stages are pinned by tests/golden/generator_00.npz, captured from the reference's own
functions (tests/golden/make_generator_golden.py); the Gaussian draw is the only random stage (the reference uses
numpy's global RNG, here the device's generator), checked statistically.

The reference reads its 1024-entry mean/stdv table from utils/r9.4_450bps.5mer.template.npz (nanopolish's r9.4 model).
That file is reference data and does not travel; without a table argument a seeded stand-in with the same ranges
(means 59.6-118.5 pA, stdvs 1.34-5.86 pA) is used.  Everything is plain torch ops on the requested device,
random numbers included.
"""
import math

import torch

KMER_WEIGHTS = (256, 64, 16, 4, 1)


def standin_kmer_table(seed=945, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    means = 59.6 + (118.5 - 59.6) * torch.rand(1024, generator=g)
    stdvs = 1.34 + (5.86 - 1.34) * torch.rand(1024, generator=g)
    return means.to(device), stdvs.to(device)


def mu_law(x, mu):
    return torch.sign(x) * torch.log1p(mu * x.abs()) / math.log1p(mu)


def kmer_indices(bases, upsampling=1):
    """bases [..., n] int64 in 1..4 -> k-mer indices [..., (n - 8) * upsampling] in 0..1023 (utils/gaussian_kmer_model.py:49,
    :58-64).  The reference slides scipy's centred 5-window over the zero-padded sequence and drops the first and last
    four outputs: what is left are the windows bases[p+2 .. p+6], p = 0 .. n-9."""
    w = torch.tensor(KMER_WEIGHTS, device=bases.device, dtype=bases.dtype)
    windows = (bases - 1).unfold(-1, 5, 1)                      # [..., n-4, 5], window i = bases[i .. i+4]
    kmers = (windows * w).sum(-1)[..., 2:-2]                    # windows 2 .. n-7
    if upsampling > 1:
        kmers = kmers.repeat_interleave(upsampling, dim=-1)
    return kmers


def gaussian_picoamps(kmers, table, generator=None):
    """N(mean[kmer], stdv[kmer]) drawn on the device of `kmers` (:67-73)"""
    means, stdvs = table
    noise = torch.randn(kmers.shape, generator=generator, device=kmers.device, dtype=means.dtype)
    return means[kmers] + stdvs[kmers] * noise


def quantize(picoamps, num_levels=256):
    """per-read normalisation, mu-law, np.digitize (:79-86); float64 like the reference's numpy arithmetic.
    Returns int64 levels in 1 .. num_levels-1 (0 is never produced: the normalised signal lies strictly inside (-1, 1))."""
    x = picoamps.double()
    span = x.amax(-1, keepdim=True) - x.amin(-1, keepdim=True)
    normalised = (x - x.mean(-1, keepdim=True)) / span
    mapped = mu_law(normalised, float(num_levels))
    edges = torch.linspace(-1.0, 1.0, num_levels, device=x.device, dtype=torch.float64)
    return torch.bucketize(mapped, edges, right=True)           # == np.digitize(mapped, edges)


def one_hot(levels, num_levels=256):
    """[..., L] int64 -> [..., num_levels, L] float32 (:89-97)"""
    out = torch.zeros(levels.shape[:-1] + (num_levels, levels.shape[-1]), device=levels.device)
    return out.scatter_(-2, levels.unsqueeze(-2), 1.0)


def gaussian_kmer_signal(batch, length, num_levels=256, upsampling=3, table=None, generator=None, device="cpu"):
    """Returns (levels [B, L] int64, one_hot [B, num_levels, L] float32, bases [B, n] int64 in 1..4): `convert_to_signal`
    (:99-104) for a batch of random reads, every random number drawn on `device` (pass a generator of that device)."""
    dev = torch.device(device)
    if generator is not None and torch.device(generator.device).type != dev.type:
        # a CPU generator with a GPU target: draw the seed from it so the call stays reproducible, then go on-device
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,), generator=generator))
        generator = torch.Generator(device=dev).manual_seed(seed)
    means, stdvs = table if table is not None else standin_kmer_table(device=dev)
    n_kmers = -(-length // upsampling)
    bases = torch.randint(1, 5, (batch, n_kmers + 8), generator=generator, device=dev)
    kmers = kmer_indices(bases, upsampling)[:, :length]
    picoamps = gaussian_picoamps(kmers, (means.to(dev), stdvs.to(dev)), generator)
    levels = quantize(picoamps, num_levels).clamp_(0, num_levels - 1)
    return levels, one_hot(levels, num_levels), bases
