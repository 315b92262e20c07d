#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: samples/sec, forward+backward, of the 256-channel,
30-block (3 cycles of dilation 1..512) WaveNet at 16k-sample sequences, batch 16 per GPU
(BASELINE.json configs[2]; configs[3] = the same with N GPUs, weak scaling, global batch 16*N).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched through torch.distributed.run)

A "step" = zero grads, forward, backward of the cotangent loss sum(out*cot), [gradient all-reduce over RCCL],
Adam update -- nothing is skipped inside the timed region.  Inputs are synthetic fixed-length one-hot
mu-law waveforms resident in HBM before timing starts; weights are random-init (reference init rules).

Prints ONE JSON line (rank 0).  Besides the contract's keys it carries
  roofline      -- dominant kernel, achieved algorithmic TFLOP/s from HIP-event timing on the launch stream
                   against the dense fp32 MFMA peak (the path is MFMA-bound)
  roofline_step -- whole-step algorithmic FLOP/s and HBM-byte fractions (both named by north_star)
  kernels       -- per-kernel-class time / launches / achieved TFLOP/s
  cpu_baseline  -- the CPU oracle (same ATen conv1d ops the reference calls) timed on this box's host cores
                   on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense v_mfma_f32_32x32x2_f32 peak
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--cycles", type=int, default=3)
    ap.add_argument("--seq-len", type=int, default=16000)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seq-len", type=int, default=16000)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the extra forward-only / forward+backward runs after the timed region (used when profiling, so "
                         "that rocprofv3's per-kernel shares are those of the timed steps)")
    return ap.parse_args()


def make_layers(channels, cycles):
    return [(channels, channels, 2, 2 ** i) for _ in range(cycles) for i in range(10)]


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: min(affinity mask, cgroup quota, 64)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(channels, cycles, seq_len, budget_s=12.0):
    """Oracle (reference-equivalent ATen CPU ops) fwd+bwd at batch 1 on a bounded sample of the workload:
    the full block stack at a sequence length chosen so one iteration takes ~budget_s; cost is linear in L,
    so samples/s of the full 16k-step utterance = (L_sample / seq_len) / t."""
    from oracle import wavenet_oracle as O
    layers = make_layers(channels, cycles)
    ncores = host_cores()
    torch.set_num_threads(ncores)
    sd = O.random_wavenet_state(channels, 2, layers, channels, seed=0)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    g = torch.Generator().manual_seed(1234)

    def run(L):
        q = torch.randint(0, channels, (1, L), generator=g)
        x = O.one_hot_encoding(q, channels)
        cot = torch.randn(1, channels, L, generator=g)
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        y = O.wavenet(x, sd, layers, False, impl="aten")
        (y * cot).sum().backward()
        return time.perf_counter() - t0

    probe_len = min(seq_len, 1024)
    run(probe_len)                       # warm-up (thread pool, oneDNN primitives)
    t_probe = run(probe_len)
    log("cpu baseline probe: %d steps in %.2f s on %d threads" % (probe_len, t_probe, ncores))
    L_s = int(min(seq_len, max(probe_len, budget_s / (t_probe / probe_len))))
    times = []
    for _ in range(4):
        times.append(run(L_s))
        log("cpu baseline: %d steps in %.2f s" % (L_s, times[-1]))
    best = min(times)
    return {"value": (L_s / float(seq_len)) / best, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": "oracle (ATen conv1d/einsum, fp32, torch %s) fwd+bwd, batch 1, %d ch x %d blocks, %d of %d time "
                      "steps per iteration (cost linear in L), best of 4 timed iterations after warm-up: %.2f s"
                      % (torch.__version__, channels, len(layers), L_s, seq_len, best)}


# timing classes of the library (wn_prof_*) -> the code-object symbol rocprofv3 reports them under.  The four block GEMMs
# and the conv / skips_sum launches are instantiations of ONE template; res, dx, skips_sum and the plain convs share the
# EPI_LINEAR instantiation, so rocprofv3 --stats lists them as one kernel (the largest line of the table).
SYMBOL_OF = {
    "series_gemm_kernel<res>": "linear", "series_gemm_kernel<dx>": "linear", "series_gemm_kernel<conv_fwd>": "linear",
    "series_gemm_kernel<conv_bwd_data>": "linear", "series_gemm_kernel<skips_sum>": "linear",
    "series_gemm_kernel<gate>": "gate", "series_gemm_kernel<dz,dgate>": "dgate", "wgrad_kernel": "wgrad",
}
SYMBOL_RE = {"linear": r"series_gemm_kernel<\d+, \d+, 0,", "gate": r"series_gemm_kernel<\d+, \d+, 1,",
             "dgate": r"series_gemm_kernel<\d+, \d+, 2,", "wgrad": r"wgrad_kernel<"}
SYMBOL_NAME = {"linear": "series_gemm_kernel<4, 4, 0, 3, 1>  [EPI_LINEAR: res, dx, skips_sum, conv launches]",
               "gate": "series_gemm_kernel<4, 4, 1, 3, 1>  [EPI_GATE]", "dgate": "series_gemm_kernel<4, 4, 2, 3, 1>  [EPI_DGATE: dz]",
               "wgrad": "wgrad_kernel<4>"}


def pmc_traffic(symbol):
    """HBM bytes per launch of a kernel symbol from the committed PMC summary (None if absent)."""
    import csv
    import re
    path = os.path.join(ROOT, "profiles", "r01", "pmc_hbm_traffic.csv")
    if symbol not in SYMBOL_RE or not os.path.exists(path):
        return None
    total = 0.0
    for row in csv.DictReader(open(path)):
        if re.search(SYMBOL_RE[symbol], row["kernel"]):
            total += float(row["avg_bytes_corrected"])
    return total or None


def main():
    args = parse()
    # RCCL / MIOpen print banners on stdout; the contract is ONE JSON line there.  Send fd 1 to stderr for the run
    # and restore it only to print the result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # under torch.distributed.run (RANK set) the RCCL group is always created, even for one rank, so the same
    # code path (init, barrier, flat-gradient all-reduce, MAX over ranks) runs at N=1 and N=8
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from wavenet_speech_amd import functional as HF
    from wavenet_speech_amd.modules.wavenet import WaveNet
    from wavenet_speech_amd.parallel import FlatGradAllReduce

    C, L, B = args.channels, args.seq_len, args.batch
    layers = make_layers(C, args.cycles)
    torch.manual_seed(0)  # identical replicas
    net = WaveNet(C, 2, layers, C, softmax=False).to(dev)
    nparams = sum(p.numel() for p in net.parameters())
    try:
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)   # same update rule, one multi-tensor kernel
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    sync = FlatGradAllReduce(net.parameters())

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)  # each rank its own shard of the global batch
    q = torch.randint(0, C, (B, L), generator=g)
    x = torch.zeros(B, C, L, device=dev).scatter_(1, q.to(dev).unsqueeze(1), 1.0)
    cot = torch.randn(B, C, L, generator=g).to(dev)

    def step():
        sync.zero()
        out = net(x)
        (out * cot).sum().backward()
        sync.reduce()
        opt.step()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log("model built: %d params; warm-up" % nparams)
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    fence()
    timing = not args.no_kernel_timing
    if timing:
        HF.profile_reset()
        HF.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    log("timed %d steps: %.1f ms/step" % (args.steps, elapsed / args.steps * 1e3))
    kern = {}
    if timing:
        HF.profile_enable(False)
        kern = HF.profile_read()
    # SURVEY.md 8(d) also asks for forward-only and forward+backward (no all-reduce / optimizer) figures: measured after
    # the timed region with events on the current stream, median over the same number of iterations
    def median_ms(fn):
        ts = []
        for _ in range(max(3, args.steps)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2]

    def fwd_only():
        with torch.no_grad():
            net(x)

    def fwd_bwd():
        sync.zero()
        (net(x) * cot).sum().backward()

    breakdown = {"peak_hbm_gib_allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}
    if not args.no_breakdown:
        breakdown.update({"fwd_only_ms": round(median_ms(fwd_only), 2), "fwd_bwd_ms": round(median_ms(fwd_bwd), 2)})
        breakdown["fwd_bwd_samples_per_s_per_gpu"] = round(B / (breakdown["fwd_bwd_ms"] * 1e-3), 2)
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- roofline ------------------------------------------------------------------------------------------
    nblk = len(layers)
    units = float(B) * L * nblk                        # (b, t, block) units per step per GPU
    alg_flops_step = 48.0 * C * C * units              # 48 C^2 flop per (b,t,block) fwd+bwd
    alg_bytes_step = 8.0 * C * 4 * units               # 8 C s bytes per (b,t,block)
    step_s = elapsed / args.steps
    kernels = {}
    by_symbol = {}
    for name, (ms, n, fl) in kern.items():
        if n == 0:
            continue
        kernels[name] = {"ms_total": round(ms, 3), "launches": n, "avg_ms": round(ms / n, 4),
                         "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms > 0 and fl > 0 else None}
        if name in SYMBOL_OF and fl > 0:
            acc = by_symbol.setdefault(SYMBOL_OF[name], [0.0, 0, 0.0])
            acc[0] += ms
            acc[1] += n
            acc[2] += fl
    roofline = None
    if by_symbol:
        # the dominant kernel = the symbol with the largest share of the timed region (what rocprofv3 --stats ranks first)
        dom = max(by_symbol, key=lambda k: by_symbol[k][0])
        ms, n, fl = by_symbol[dom]
        ach = fl / (ms * 1e-3) / 1e12
        roofline = {"kernel": SYMBOL_NAME[dom] if C > 64 else dom, "bound": "mfma", "achieved": round(ach, 2),
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    "traffic": pmc_traffic(dom),
                    "traffic_unit": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, "
                                    "separate passes, from profiles/r01/pmc_hbm_traffic.csv (not collected live)",
                    "avg_launch_ms": round(ms / n, 4), "launches": n, "flops_per_launch": fl / n,
                    "share_of_step": round(ms / args.steps / (step_s * 1e3), 4),
                    "other_symbols": {SYMBOL_NAME[k] if C > 64 else k:
                                      {"avg_launch_ms": round(v[0] / v[1], 4), "tflops": round(v[2] / (v[0] * 1e-3) / 1e12, 2),
                                       "frac": round(v[2] / (v[0] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                       "share_of_step": round(v[0] / args.steps / (step_s * 1e3), 4)}
                                      for k, v in by_symbol.items() if k != dom}}
    kernel_ms = sum(v[0] for v in kern.values())
    roofline_step = {
        "algorithmic_tflops": round(alg_flops_step / step_s / 1e12, 2),
        "mfma_frac": round(alg_flops_step / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
        "algorithmic_hbm_gbs": round(alg_bytes_step / step_s / 1e9, 1),
        "hbm_frac": round(alg_bytes_step / step_s / 1e9 / PEAK_HBM_GBS, 4),
        "hip_kernel_ms_per_step": round(kernel_ms / args.steps, 2) if kern else None,
        "accounting": "48*C^2 flop and 8*C*4 B per (utterance, time step, block), SURVEY.md 8(d)",
    }

    result = {
        "metric": "samples/sec fwd+bwd, 256-ch 30-block WaveNet @16k seq",
        "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: WaveNet %d ch, %d blocks (%d x dilation 1..512), k=2, "
                               "seq_len %d, batch %d per GPU, fp32, full step (fwd+bwd+grad all-reduce+Adam)"
                               % (C, nblk, args.cycles, L, B),
                   "channels": C, "blocks": nblk, "seq_len": L, "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": "dp%d" % world, "params": nparams},
        "per_gpu": round(value / world, 3),
        "breakdown": breakdown, "roofline": roofline, "roofline_step": roofline_step, "kernels": kernels,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del net, opt, sync, x, cot
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline(C, args.cycles, args.cpu_seq_len)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
