#!/usr/bin/env python3
"""Diagnostic: per-wave phase timing of series_gemm_kernel from the -DWN_STAMPS build (make -C csrc stamps).
Reports prologue / K-loop / epilogue cycles per wave, the in-kernel clock, SIMD idle gaps between waves.
Usage: wave_stamps.py taps [C B L]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from wavenet_speech_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "wavenet_speech_amd", "libwavenet_amd_stamps.so")
from wavenet_speech_amd import functional as HF

k = int(sys.argv[1]); C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16; L = int(sys.argv[4]) if len(sys.argv) > 4 else 16000
dev = "cuda:0"
lib = _lib.load()
x = torch.randn(B, C, L, device=dev); w = torch.randn(C, C, k, device=dev) * 0.05; b = torch.randn(C, device=dev)
buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(int(os.environ.get("WARM", "3"))):   # WARM=1000 to reach the steady-state clock (DVFS ramps for ~0.1 s)
        HF.dilated_conv(x, w, b, 1, True)
    torch.cuda.synchronize()
    lib.wn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.wn_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    HF.dilated_conv(x, w, b, 1, True)
    torch.cuda.synchronize()
    lib.wn_debug_set_stamp_buffer(None)
grid = ctypes.c_uint.in_dll(lib, "wn_debug_last_grid").value
s = buf.cpu().numpy().reshape(-1, 8)[:grid]
s = s[s[:, 0] != 0]
t0, t1, t2, t3, r0, r1, hw, xcc = [s[:, i].astype(np.int64) for i in range(8)]
issued = xcc >> 32; xcc = xcc & 0xffffffff   # cycles from K-loop end to the last epilogue store's issue
cyc = t3 - t0; ns = (r1 - r0) * 10.0
clock = cyc.sum() / ns.sum()
print("waves %d  K-blocks %d" % (len(s), k * C // 8))
print("in-kernel clock: %.3f GHz" % clock)
for name, d in (("prologue (entry -> first K-block issue)", t1 - t0), ("K loop", t2 - t1), ("epilogue: K-loop end -> last store issued", issued), ("store drain (last issue -> all acknowledged)", t3 - t2 - issued), ("epilogue + store drain", t3 - t2), ("wave total", cyc)):
    print("%-42s median %8.0f cyc (%6.2f us)   p10 %8.0f  p90 %8.0f" % (name, np.median(d), np.median(d) / clock / 1e3, np.percentile(d, 10), np.percentile(d, 90)))
ideal = (k * C // 8) * 64 * 64
print("ideal K loop = %d cyc -> K-loop efficiency %.1f%%" % (ideal, 100.0 * ideal / np.median(t2 - t1)))
# SIMD occupancy: group by (xcc, se, sh, cu, simd)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc & 15) << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd
gaps = []; busy = 0; span = 0
for kk in np.unique(key):
    idx = np.where(key == kk)[0]
    o = idx[np.argsort(r0[idx])]
    st, en = r0[o] * 10.0, r1[o] * 10.0
    gaps.extend((st[1:] - en[:-1]).tolist())
    busy += (en - st).sum()
kernel_ns = (r1.max() - r0.min()) * 10.0
print("distinct SIMD slots seen: %d; waves per slot %.2f" % (len(np.unique(key)), len(s) / len(np.unique(key))))
print("kernel span %.1f us; mean slot busy %.1f%%" % (kernel_ns / 1e3, 100.0 * busy / (len(np.unique(key)) * kernel_ns)))
g = np.array(gaps)
if len(g):
    print("gap between consecutive waves on one SIMD slot: median %.2f us  p90 %.2f us  max %.2f us" % (np.median(g) / 1e3, np.percentile(g, 90) / 1e3, g.max() / 1e3))
print("first wave start spread: %.2f us; last-wave finish spread (p50 -> max of end times): %.1f us" % ((np.percentile(r0, 25) - r0.min()) * 10.0 / 1e3, (r1.max() - np.percentile(r1, 50)) * 10.0 / 1e3))
