#!/usr/bin/env python3
"""Context baseline: the reference's op sequence (oracle, ATen conv1d / einsum) executed by stock PyTorch-ROCm eager ON
THE GPU -- what `.cuda()` on the reference modules gives today -- for the cfg3 WaveNet, fwd+bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
C, L = 256, 16000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
layers = [(C, C, 2, 2 ** i) for _ in range(3) for i in range(10)]
sd = {k: v.to(dev).requires_grad_(True) for k, v in O.random_wavenet_state(C, 2, layers, C, seed=0).items()}
x = torch.zeros(B, C, L, device=dev).scatter_(1, torch.randint(0, C, (B, 1, L), device=dev), 1.0)
cot = torch.randn(B, C, L, device=dev)
def wavenet_gpu(x):
    # oracle.wavenet allocates skips on the CPU; same math with device tensors
    out = O.dilated_conv(x, sd["entry_conv1d.conv1d.weight"], sd["entry_conv1d.conv1d.bias"], 1, True, "aten")
    _, skips = O.block_stack(out, torch.zeros(B, C, L, device=dev), sd, layers, True, "aten")
    y = torch.nn.functional.leaky_relu(skips, 0.01)
    y = O.conv1x1(y, sd["output_stack.1.weight"], sd["output_stack.1.bias"])
    y = torch.nn.functional.leaky_relu(y, 0.01)
    return O.conv1x1(y, sd["output_stack.3.weight"], sd["output_stack.3.bias"])
def step():
    for v in sd.values(): v.grad = None
    (wavenet_gpu(x) * cot).sum().backward()
for i in range(2):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("warm-up %d: %.2f s" % (i, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("PyTorch-ROCm eager on MI355X, B=%d: %.1f ms/step = %.2f samples/s, peak memory %.1f GB" % (B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9))
