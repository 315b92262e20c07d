#!/usr/bin/env python3
"""How much of a training step is HOST time?  Runs the cfg2 (or cfg3) model at a tiny sequence length, where every kernel is a
few microseconds, so the step time is what Python + ctypes + the HIP runtime spend issuing it.  Prints ms/step and the
top cProfile entries.   usage: host_time_probe.py [cfg2|cfg3] [precision]"""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wavenet_speech_amd as W

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
prec = sys.argv[2] if len(sys.argv) > 2 else ("bf16" if cfg == "cfg2" else "f16x3")
dev = torch.device("cuda:0")
torch.manual_seed(0)
if cfg == "cfg2":
    from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
    C = 128
    layers = [(C, C, 2, 2 ** i) for i in range(10)]
    net = RawCTCNet(C, 3, 5, layers, C, softmax=False, causal=False).to(dev)
    L, B = 1200, 1
    x = torch.randn(B, 1, L, device=dev)
    cot = torch.randn(B, 5, L + 2, device=dev)
else:
    from wavenet_speech_amd.modules.wavenet import WaveNet
    C = 256
    layers = [(C, C, 2, 2 ** i) for _ in range(3) for i in range(10)]
    net = WaveNet(C, 2, layers, C, softmax=False).to(dev)
    L, B = 1200, 1
    x = torch.randn(B, C, L, device=dev)
    cot = torch.randn(B, C, L, device=dev)
W.set_precision(net, prec)
opt = torch.optim.Adam(net.parameters(), lr=0.0, fused=True)   # lr 0: the random-init model must not drift into fp16 overflow while probing
from wavenet_speech_amd.parallel import FlatGradAllReduce
sync = FlatGradAllReduce(net.parameters())

def step():
    sync.zero()
    out = net(x)
    (out * cot).sum().backward()
    sync.reduce()
    opt.step()

for _ in range(5):
    step()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s %s tiny-L: issue %.3f ms/step, issue+drain %.3f ms/step" % (cfg, prec, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

# which tensors do the small fill / copy kernels touch?  (torch.profiler with shapes, 3 steps; backward runs on the autograd
# thread, so Python stacks are not available there -- shapes identify the call sites well enough)
if os.environ.get("WN_PROBE_FILLS", "1") != "0":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    from collections import Counter
    sites = Counter()
    for ev in prof.events():
        if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::mul_", "aten::cat", "aten::add", "aten::mul",
                       "aten::sum", "aten::bmm", "aten::matmul"):
            own = [s for s in (ev.stack or []) if "wavenet_speech_amd" in s or "host_time_probe" in s]
            sites[(ev.name, str(ev.input_shapes)[:60], own[0][-70:] if own else "-")] += 1
    for (name, shp, site), n in sites.most_common(45):
        print("%6.1f per step  %-12s %-60s %s" % (n / 3.0, name, shp, site))
