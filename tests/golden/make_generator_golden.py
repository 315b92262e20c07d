#!/usr/bin/env python3
"""
Writes tests/golden/generator_00.npz: golden vectors of the DETERMINISTIC stages of the reference's on-line signal
generator, captured by importing /root/reference/utils/gaussian_kmer_model.py in the build container and calling its own
code (the k-mer window lambda through scipy's generic_filter exactly as gaussian_model_fn does at :58-59 incl. the
[4:-4] trim and the :62-64 upsampling; quantize_fn :79-86; one_hot_fn :89-97).  The Gaussian draw (:73) is the only
random stage and is not part of the fixture.  The 1024-entry mean/stdv table used here is synthetic (seeded), not the
reference's nanopolish table.  The reference itself never travels: only these numbers are committed.

    python tests/golden/make_generator_golden.py
"""
import os
import sys
import tempfile
import warnings

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")
from scipy.ndimage import generic_filter  # noqa: E402
from utils.gaussian_kmer_model import GaussianModelLoader  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.RandomState(20260101)
    means = 59.6 + (118.5 - 59.6) * rng.rand(1024)
    stdvs = 1.34 + (5.86 - 1.34) * rng.rand(1024)
    with tempfile.TemporaryDirectory() as tmp:
        table = os.path.join(tmp, "table.npz")
        np.savez(table, means=means, stdvs=stdvs)
        out = {"table.means": means, "table.stdvs": stdvs}
        for case, (n_bases, ups, levels) in enumerate([(40, 3, 256), (173, 1, 256), (64, 4, 64)]):
            ld = GaussianModelLoader(10, 1, 10, table, batch_size=1, num_levels=levels, upsampling=ups)
            bases = rng.randint(1, 5, size=n_bases)
            # gaussian_model_fn :58-64 without the random draw
            kmer_seq = generic_filter(bases, ld.nts_to_kmer, size=(5,), mode='constant')
            kmer_seq = kmer_seq[4:-4].astype(int)
            if ups > 1:
                kmer_seq = kmer_seq.repeat(ups, axis=0)
            k_means = np.array([ld.kmer_means[k] for k in kmer_seq])
            k_stdvs = np.array([ld.kmer_stdvs[k] for k in kmer_seq])
            noise = rng.randn(kmer_seq.shape[0])
            picoamps = k_means + k_stdvs * noise            # a fixed stand-in for the draw of :73
            quantized = ld.quantize_fn(picoamps)
            one_hot = ld.one_hot_fn(quantized)
            pre = "case%d." % case
            out.update({pre + "bases": bases.astype(np.int64), pre + "upsampling": np.int64(ups), pre + "num_levels": np.int64(levels),
                        pre + "kmer_seq": kmer_seq.astype(np.int64), pre + "kmer_means": k_means, pre + "kmer_stdvs": k_stdvs,
                        pre + "noise": noise, pre + "picoamps": picoamps, pre + "quantized": quantized.astype(np.int64),
                        pre + "one_hot": one_hot})
        np.savez_compressed(os.path.join(HERE, "generator_00.npz"), **out)
        print("wrote generator_00.npz:", {k: np.asarray(v).shape for k, v in out.items() if k.startswith("case0")})


if __name__ == "__main__":
    main()
