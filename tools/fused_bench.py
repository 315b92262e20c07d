#!/usr/bin/env python3
"""Time wn_hblock_forward (training form: z, sg, r stored) at BASELINE configs[1]'s block shape, fused vs two launches.
usage: fused_bench.py [C] [L] [B] [d]   (env WN_FUSED_FWD=0 for the two-launch path, WN_FUSED_DBG=bits for ablations)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import _lib
from wavenet_speech_amd import functional as HF
from wavenet_speech_amd import functional_half as FH

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4098
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
d = int(sys.argv[4]) if len(sys.argv) > 4 else 16
prec = os.environ.get("PREC", "bf16")
dev = torch.device("cuda:0")
lib = _lib.load()
mode = FH._Mode(prec)
spec = HF.BlockSpec(C, C, C, 2, d, False)
layout = FH.HalfLayout(L, spec.reach())
shape = FH._shape(spec, B, layout)
torch.manual_seed(0)
prm = [torch.randn(C, C, 2, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1, torch.randn(C, C, 2, device=dev) * 0.05,
       torch.randn(C, device=dev) * 0.1, torch.randn(C, C, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1,
       torch.randn(C, C, device=dev) * 0.05, torch.randn(C, device=dev) * 0.1, torch.randn(C, C, device=dev) * 0.05,
       torch.randn(C, device=dev) * 0.1]
nbytes = lib.wn_hblock_packed_bytes(ctypes.byref(shape), mode.code)
packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
ps = HF._params_struct(prm)
_lib.check(lib.wn_hblock_pack(ctypes.byref(shape), mode.code, ctypes.byref(ps), HF._p(packed), HF._stream()), "pack")
x = FH._hlease(mode, B, C, layout, dev)
FH._load(lib, mode, torch.randn(B, C, L, device=dev), x, layout, float(lib.wn_hseries_residual_scale()), None, None)
r, sg, z = (FH._hlease(mode, B, C, layout, dev) for _ in range(3))

def run():
    _lib.check(lib.wn_hblock_forward(ctypes.byref(shape), mode.code, HF._p(packed), HF._p(x), HF._p(r), None, 0, HF._p(sg), HF._p(z),
                                     None, HF._stream()), "fwd")
for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 200
e0.record()
for _ in range(N):
    run()
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) / N * 1e3
mb = B * L * C * 2 * 4 / 1e6
print("C %d L %d B %d d %d %s fused=%s dbg=%s: %.1f us per block forward  (%.0f MB algorithmic -> %.2f TB/s)" % (
    C, L, B, d, prec, os.environ.get("WN_FUSED_FWD", "1"), os.environ.get("WN_FUSED_DBG", "0"), us, mb, mb / us * 1e-6 * 1e6 / 1e6))
if os.environ.get("WN_FUSED_STAMPS"):
    import numpy as np
    buf = (ctypes.c_ulonglong * (8 * 4096))()
    n = ctypes.CDLL(_lib.LIB_PATH).wn_debug_fused_stamps(buf, 4096)
    a = np.frombuffer(buf, dtype=np.uint64)[:8 * n].reshape(n, 8).astype(np.int64)
    a = a[a[:, 7] > 0]
    d = np.diff(a, axis=1)
    names = ["gate half 0 K loop", "gate 0 epilogue", "gate 1 (K loop + epilogue)", "res K loop", "res epilogue", "-", "drain"]
    print("%d workgroups; wave 0 lifetime %.0f cycles (s_memtime ticks, 100 MHz? see below)" % (len(a), (a[:, 7] - a[:, 0]).mean()))
    for i, nm in enumerate(names):
        print("  %-24s mean %8.0f  median %8.0f" % (nm, d[:, i].mean(), np.median(d[:, i])))
    span = a[:, 7].max() - a[:, 0].min()
    print("  kernel span %d ticks; first start spread %d; starts in second half of span: %d" % (span, a[:, 0].max() - a[:, 0].min(), (a[:, 0] > a[:, 0].min() + span // 2).sum()))
