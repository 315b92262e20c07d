#!/usr/bin/env python3
"""Time the HIP read generator (csrc/wn_synth.hip) against the same stages as torch ops on the GPU.
Usage: synth_bench.py [batch length]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from wavenet_speech_amd import synthetic as S  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16000
dev = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


g = torch.Generator(device=dev).manual_seed(1)
bases = S.hip_bases(B, -(-L // 3) + 8, 11, dev)
full = timed(lambda: S.hip_signal(bases, L, 256, 3, None, seed=3))
lean = timed(lambda: S.hip_signal(bases, L, 256, 3, None, seed=3, want_one_hot=False))


def torch_ops():
    kmers = S.kmer_indices(bases, 3)[:, :L]
    means, stdvs = S.standin_kmer_table(device=dev)
    pico = S.gaussian_picoamps(kmers, (means.double(), stdvs.double()), g)
    lv = S.quantize(pico, 256)
    return S.one_hot(lv, 256)


ref = timed(torch_ops)
onehot_bytes = B * 256 * L * 4
print("generator %d x %d, 256 levels, upsampling 3 (ms per call incl. host launch overhead)" % (B, L))
print("  HIP, levels + dense one-hot : %.3f ms  (%.0f GB/s of one-hot stores)" % (full, onehot_bytes / full / 1e6))
print("  HIP, levels only            : %.3f ms" % lean)
print("  torch ops on the GPU        : %.3f ms" % ref)
