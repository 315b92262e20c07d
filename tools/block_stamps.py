#!/usr/bin/env python3
"""Diagnostic: in-situ per-wave phase timing of the four block GEMMs (gate, res+skip, dz, dx) at cfg3 block size,
from the -DWN_STAMPS build.  Usage: block_stamps.py [C B L d]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from wavenet_speech_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "wavenet_speech_amd", "libwavenet_amd_stamps.so")
from wavenet_speech_amd.modules.block import ResidualBlock, run_stack
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
d = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = "cuda:0"
lib = _lib.load()
lib.wn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
blocks = torch.nn.ModuleList([ResidualBlock(C, C, 2, d), ResidualBlock(C, C, 2, 2 * d)]).to(dev)
botts = torch.nn.ModuleList([torch.nn.Conv1d(C, C, 1), torch.nn.Conv1d(C, C, 1)]).to(dev)
x = torch.randn(B, C, L, device=dev, requires_grad=True); cot = torch.randn(B, C, L, device=dev)
buf = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
def step():
    S = run_stack(x, blocks, botts); (S * cot).sum().backward()
for _ in range(2): step()
torch.cuda.synchronize()
names = {1: "gate GEMM", 2: "res GEMM", 9: "skip+= GEMM", 3: "dz GEMM", 4: "dx GEMM"}
for kc, name in names.items():
    buf.zero_()
    lib.wn_debug_set_stamp_class(kc)
    lib.wn_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    step(); torch.cuda.synchronize()
    lib.wn_debug_set_stamp_buffer(None)
    grid = ctypes.c_uint.in_dll(lib, "wn_debug_last_grid").value
    s = buf.cpu().numpy().reshape(-1, 8)[:grid]; s = s[s[:, 0] != 0]
    t0, t1, t2, t3, r0, r1 = [s[:, i].astype(np.int64) for i in range(6)]
    clock = (t3 - t0).sum() / ((r1 - r0).sum() * 10.0)
    span = (r1.max() - r0.min()) * 10.0 / 1e3
    f = lambda a: "%7.0f cyc %6.1f us" % (np.median(a), np.median(a) / clock / 1e3)
    print("%-14s waves %5d clock %.2f GHz | prologue %s | K loop %s (p90 %.0f) | epilogue+drain %s | wave %s | kernel span %.0f us"
          % (name, len(s), clock, f(t1 - t0), f(t2 - t1), np.percentile(t2 - t1, 90), f(t3 - t2), f(t3 - t0), span))
