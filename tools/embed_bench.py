"""Timing of the embedding-gather entry conv (wn_embed_*) against the one-hot GEMM path at cfg3's size."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_speech_amd import functional as HF
dev = "cuda:0"
B, C, L = 16, 256, 16000
w = torch.randn(C, 256, 2, device=dev, requires_grad=True); b = torch.randn(C, device=dev, requires_grad=True)
q = torch.randint(0, 256, (B, L), device=dev); cot = torch.randn(B, C, L, device=dev)

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n

y = HF.embed_conv(q, w, b)
print("embed forward  %.3f ms" % t(lambda: HF.embed_conv(q, w, b)))
def fb():
    w.grad = b.grad = None
    (HF.embed_conv(q, w, b)).backward(cot)
print("embed fwd+bwd  %.3f ms" % t(fb))
x = torch.zeros(B, 256, L, device=dev).scatter_(1, q.unsqueeze(1), 1.0)
print("one-hot build  %.3f ms" % t(lambda: torch.zeros(B, 256, L, device=dev).scatter_(1, q.unsqueeze(1), 1.0)))
print("dense forward  %.3f ms" % t(lambda: HF.dilated_conv(x, w, b, 1, True)))
def fb2():
    w.grad = b.grad = None
    HF.dilated_conv(x, w, b, 1, True).backward(cot)
print("dense fwd+bwd  %.3f ms" % t(fb2))
