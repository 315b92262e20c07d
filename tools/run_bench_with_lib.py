#!/usr/bin/env python3
"""A/B helper: run bench.py against another build of the library in the same gpurun call (same box, same clocks).
    WN_LIB=libwavenet_amd_variant.so python tools/run_bench_with_lib.py --precision f16x3 ...
The variant .so must sit next to libwavenet_amd.so (wavenet_speech_amd/); without WN_LIB this is plain bench.py."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wavenet_speech_amd._lib as L  # noqa: E402

if os.environ.get("WN_LIB"):
    L.LIB_PATH = os.path.join(os.path.dirname(L.__file__), os.environ["WN_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
