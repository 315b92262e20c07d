#!/usr/bin/env python3
"""hcol_kernel (column-owner dz / dx) against hgemm_kernel on the same stack: launched kernel classes, differences, timing."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenet_speech_amd import functional as HF
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fused import _Stack, _eval

dev = "cuda:0"
for prec in ("bf16", "f16"):
    for (c, dil, B, L) in ((128, (1, 2, 4, 512), 32, 4098), (96, (1, 8), 3, 700), (32, (1, 2), 2, 100)):
        net = _Stack(c, dil, c, False, seed=1).to(dev)
        net.stack_state.precision = prec
        torch.manual_seed(0)
        x = torch.randn(B, c, L, device=dev)
        cot = torch.randn(B, c, L, device=dev)
        res = {}
        for knob in ("1", "0"):
            os.environ["WN_COL_BWD"] = knob
            n2 = copy.deepcopy(net)
            _eval(n2, x, cot)
            HF.profile_reset(); HF.profile_enable(True)
            for _ in range(5):
                out = _eval(n2, x, cot)
            HF.profile_enable(False)
            k = HF.profile_read()
            res[knob] = out
            print(prec, c, dil, B, L, "WN_COL_BWD=" + knob, {n: "%d x %.1f us" % (v[1], 1e3 * v[0] / v[1]) for n, v in k.items() if v[1] and ("dz" in n or "dx" in n or "fused" in n)})
        a, b_ = res["1"], res["0"]
        err = {"dx": float((a[1] - b_[1]).abs().max() / b_[1].abs().max())}
        for kname in a[2]:
            if a[2][kname] is not None:
                err[kname] = float((a[2][kname] - b_[2][kname]).abs().max() / max(float(b_[2][kname].abs().max()), 1e-30))
        w = max(err, key=err.get)
        print("   col vs hgemm: dx %.2e, worst %s %.2e" % (err["dx"], w, err[w]))
