#!/usr/bin/env python3
"""Time the HIP CTC (csrc/wn_ctc.hip: loss + gradient of the [B, C, T] logits in place) against torch's own ctc_loss on the
GPU (log_softmax + permute + loss + backward) and, as the reference's arrangement, on the CPU after a device->host copy
(pretrain_tnt.py:159 moves the activations to the CPU for warp-ctc every step).  Usage: ctc_bench.py [B C T labels]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from wavenet_speech_amd import training as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Tn = int(sys.argv[3]) if len(sys.argv) > 3 else 4098
LM = int(sys.argv[4]) if len(sys.argv) > 4 else 420
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, C, Tn, generator=g) * 2.0).to(dev)
labels = torch.randint(1, C, (B, LM), generator=g).to(dev)
lens = torch.randint(LM * 3 // 4, LM + 1, (B,), generator=g).to(dev)
in_len = torch.full((B,), Tn, dtype=torch.long, device=dev)


def hip():
    xx = x.clone().requires_grad_(True)
    T.ctc_total(xx, labels, lens).backward()
    return xx.grad


def torch_gpu():
    xx = x.clone().requires_grad_(True)
    F.ctc_loss(F.log_softmax(xx.permute(2, 0, 1), dim=2), labels, in_len, lens, blank=0, reduction="sum").backward()
    return xx.grad


def torch_cpu():
    xx = x.cpu().requires_grad_(True)
    F.ctc_loss(F.log_softmax(xx.permute(2, 0, 1), dim=2), labels.cpu(), in_len.cpu(), lens.cpu(), blank=0, reduction="sum").backward()
    return xx.grad.to(dev)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


a, b_ = hip(), torch_gpu()
print("CTC loss + gradient, %d utterances x %d classes x %d frames, %d..%d labels" % (B, C, Tn, LM * 3 // 4, LM))
print("  HIP (wn_ctc_loss, float64 recursions)       : %8.3f ms" % timed(hip, 10))
print("  torch ctc_loss on the GPU (fp32 recursions) : %8.3f ms   max |grad difference| to HIP %.2e" % (timed(torch_gpu, 10), float((a - b_).abs().max())))
print("  torch ctc_loss on the CPU incl. copies      : %8.3f ms   (%d threads)" % (timed(torch_cpu, 3), torch.get_num_threads()))
