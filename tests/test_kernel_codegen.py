"""CPU (hipcc cross-compiles without a GPU): properties of the generated gfx950 code that the kernels' performance and, for
the LDS-DMA rings, their design rest on -- checked on the assembly, because hipcc is free to change them silently.

* no instantiation of any kernel uses scratch (register spills);
* the K loop of every hgemm / hwgrad instantiation keeps its ring in flight: one counted `s_waitcnt vmcnt(N > 0)` before
  the barrier and no compiler-inserted `vmcnt(0)` (hipcc did insert one in hwgrad_kernel)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wavenet_speech_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


# per-file flags of csrc/Makefile (FLAGS_<file>): the assembly checked here must be the product's
EXTRA_FLAGS = {"wn_half.hip": ["-fno-slp-vectorize"], "wn_fused.hip": ["-fno-slp-vectorize"]}


def _asm(src, tmp_path):
    out = str(tmp_path / (src + ".s"))
    r = subprocess.run([HIPCC, "-O3", "-std=c++20", "--offload-arch=gfx950", "--cuda-device-only"] + EXTRA_FLAGS.get(src, []) +
                       ["-S", os.path.join(CSRC, src), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def test_ring_kernels_keep_their_prefetch_in_flight(tmp_path):
    files = [_asm("wn_half.hip", tmp_path), _asm("wn_half_wgrad.hip", tmp_path), _asm("wn_fused.hip", tmp_path)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rings.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    # 15 hgemm (12 at 128 rows, 3 at 256) + 24 hgemm8 + 3 hwgrad + 24 fused-forward instantiations
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 66, r.stdout


@pytest.mark.parametrize("src", ["wn_gemm.hip", "wn_wgrad.hip", "wn_half.hip", "wn_half_wgrad.hip", "wn_fused.hip", "wn_embed.hip",
                                 "wn_nll.hip", "wn_pack.hip"])
def test_no_kernel_uses_scratch(src, tmp_path):
    text = open(_asm(src, tmp_path)).read()
    sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)
    assert sizes, "no kernel metadata found"
    assert all(int(x) == 0 for x in sizes), "scratch in use: %s" % sizes
    assert "scratch_store" not in text and "scratch_load" not in text
