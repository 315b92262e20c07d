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
(means 59.6-118.5 pA, stdvs 1.34-5.86 pA) is used.

On a GPU `gaussian_kmer_signal` runs the HIP generator (csrc/wn_synth.hip through the C ABI's wn_synth_*: Philox
nucleotides and Gaussian noise, float64 signal, per-read normalisation, mu-law, digitize, optional one-hot -- three
launches, nothing drawn or computed on the host; it fails loudly if the library is missing).  The stage functions below
are the same arithmetic as plain torch ops: the CPU form that the fixture pins, and the checker of the HIP kernels.
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


def _device_seed(generator, dev):
    """one 63-bit seed for the Philox streams of the HIP generator, drawn from the caller's torch generator (or torch's
    default one) so that a seeded call stays reproducible"""
    g = generator if generator is not None and torch.device(generator.device).type == "cpu" else None
    if generator is not None and g is None:
        return int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=dev).item())
    return int(torch.randint(0, 2 ** 62, (1,), generator=g).item())


def hip_signal(bases, length, num_levels=256, upsampling=3, table=None, seed=0, noise=None, want_one_hot=True, picoamps=None):
    """The HIP stages on given nucleotides (device int64 [B, n]).  `noise` (float64 [B, L]) replaces the Philox Gaussian
    draw; `picoamps` (float64 [B, L]) skips the signal stage altogether (quantize + one-hot of a given signal).
    Returns (levels, one_hot or None, picoamps)."""
    import ctypes
    from . import _lib
    from .functional import _p, _stream
    lib = _lib.load()
    dev = bases.device
    if dev.type != "cuda":
        raise RuntimeError("wavenet_speech_amd: the HIP generator needs device tensors")
    B, nb = bases.shape
    with torch.cuda.device(dev):
        ws_bytes = lib.wn_synth_workspace_bytes(B, length)
        if ws_bytes == 0:
            _lib.check(-1, "wn_synth_workspace_bytes")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        means, stdvs = table if table is not None else standin_kmer_table(device=dev)
        means, stdvs = means.to(dev).double().contiguous(), stdvs.to(dev).double().contiguous()
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        given = picoamps is not None
        if given:
            # statistics of a caller's signal: run the signal stage on it as "noise" with unit tables (x = 0 + 1 * x)
            pico = torch.empty(B, length, dtype=torch.float64, device=dev)
            zeros, ones = torch.zeros(1024, dtype=torch.float64, device=dev), torch.ones(1024, dtype=torch.float64, device=dev)
            _lib.check(lib.wn_synth_signal(_p(bases.contiguous()), B, nb, length, upsampling, _p(zeros), _p(ones), 0,
                                           _p(picoamps.double().contiguous()), _p(pico), _p(ws), ws_bytes, None, _stream()),
                       "wn_synth_signal")
        else:
            pico = torch.empty(B, length, dtype=torch.float64, device=dev)
            nz = None if noise is None else noise.double().contiguous()
            _lib.check(lib.wn_synth_signal(_p(bases.contiguous()), B, nb, length, upsampling, _p(means), _p(stdvs),
                                           ctypes.c_ulonglong(seed), _p(nz), _p(pico), _p(ws), ws_bytes, _p(bad), _stream()),
                       "wn_synth_signal")
        edges = torch.linspace(-1.0, 1.0, num_levels, device=dev, dtype=torch.float64)
        levels = torch.empty(B, length, dtype=torch.int64, device=dev)
        oh = torch.empty(B, num_levels, length, dtype=torch.float32, device=dev) if want_one_hot else None
        _lib.check(lib.wn_synth_quantize(_p(pico), _p(ws), ws_bytes, B, length, num_levels, _p(edges), _p(levels), _p(oh),
                                         _stream()), "wn_synth_quantize")
        if not given and int(bad.item()):
            raise RuntimeError("wavenet_speech_amd: nucleotides outside 1..4 in the generator's input")
    return levels, oh, pico


def hip_bases(batch, nbases, seed, device):
    import ctypes
    from . import _lib
    from .functional import _p, _stream
    lib = _lib.load()
    dev = torch.device(device)
    with torch.cuda.device(dev):
        bases = torch.empty(batch, nbases, dtype=torch.int64, device=dev)
        _lib.check(lib.wn_synth_bases(ctypes.c_ulonglong(seed), batch, nbases, _p(bases), _stream()), "wn_synth_bases")
    return bases


def gaussian_kmer_signal(batch, length, num_levels=256, upsampling=3, table=None, generator=None, device="cpu",
                         want_one_hot=True):
    """Returns (levels [B, L] int64, one_hot [B, num_levels, L] float32, bases [B, n] int64 in 1..4): `convert_to_signal`
    (:99-104) for a batch of random reads, every random number drawn on `device` (pass a generator of that device).
    On a GPU this is the HIP generator (one_hot is None with want_one_hot=False: the level-index entry conv needs none)."""
    dev = torch.device(device)
    if dev.type == "cuda":
        seed = _device_seed(generator, dev)
        n_kmers = -(-length // upsampling)
        bases = hip_bases(batch, n_kmers + 8, seed, dev)
        levels, oh, _ = hip_signal(bases, length, num_levels, upsampling, table, seed, want_one_hot=want_one_hot)
        return levels, oh, bases
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
