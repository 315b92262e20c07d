#!/usr/bin/env python3
"""
bench.py -- headline benchmark of BASELINE.json: samples/sec, forward+backward, of the 256-channel,
30-block (3 cycles of dilation 1..512) WaveNet at 16k-sample sequences, batch 16 per GPU
(BASELINE.json configs[2]; configs[3] = the same with N GPUs, weak scaling, global batch 16*N).

    python bench.py --gpus N --steps K --warmup W [--config cfg3|cfg2|cfg5] [--precision f32|f16x3|f16|bf16]

N > 1 without a launcher: bench.py starts the N ranks itself (a child `python -m torch.distributed.run`, one rank
per GPU over RCCL) BEFORE anything touches the GPU, relays rank 0's JSON line and the child's exit code.  Under an
external launcher (RANK / WORLD_SIZE in the environment) WORLD_SIZE must equal --gpus.  Fewer visible devices than
--gpus is an error (exit code 3) -- there is no silent fall-back to fewer GPUs.

A "step" = zero grads, forward, backward of the cotangent loss sum(out*cot), [gradient all-reduce over RCCL],
Adam update -- nothing is skipped inside the timed region.  Inputs are synthetic fixed-length waveforms resident in
HBM before timing starts; weights are random-init (reference init rules).

Prints ONE JSON line (rank 0).  Besides the contract's keys it carries
  roofline      -- dominant kernel, achieved algorithmic TFLOP/s from HIP-event timing on the launch stream
                   against the dense MFMA peak of the arithmetic used (the path is MFMA-bound)
  roofline_step -- whole-step algorithmic FLOP/s and HBM-byte fractions (both named by north_star)
  kernels       -- per-kernel-class time / launches / achieved TFLOP/s
  ranks         -- rccl_ranks (dist.get_world_size()), per-rank ms/step, gradient all-reduce ms/step (HIP events)
  cpu_baseline  -- the CPU oracle (same ATen conv1d ops the reference calls) timed on this box's host cores
                   on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense v_mfma_f32_32x32x2_f32 peak
PEAK_HALF_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA peak (the 5 PF headline figure is 2:1 sparse)
# What the chip SUSTAINS under back-to-back v_mfma_f32_32x32x16_f16 on random data: it holds ~1.75 GHz, not 2.4 (DVFS).  Measured
# here with tools/probes/hwgrad_loop.hip (profiles/r02/half_tuning.txt item 11); reported beside `peak`, never instead of it.
SUSTAINED_HALF_MFMA_TFLOPS = 1600.0
SUSTAINED_NOTE = ("1600 TFLOP/s = what back-to-back v_mfma_f32_32x32x16_f16 issue on random operands sustains on this chip (98 % of "
                  "the cycle count at a DVFS-held 1.75 GHz; tools/probes/hwgrad_loop.hip, profiles/r02/half_tuning.txt item 11); "
                  "streaming the operands from HBM beside it lowers the clock further (1.53 GHz in the same model)")
PEAK_HBM_GBS = 8000.0           # HBM3E spec peak
PROFILE_ROUND = "r03"

# BASELINE.json configs -> workloads.  cfg3 is the configuration the metric is quoted on; the others are parity-test
# cases that can be timed with the same harness (their lines are committed under profiles/, they are not the headline).
CONFIGS = {
    "cfg3": dict(model="wavenet", channels=256, cycles=3, seq_len=16000, batch=16, precision="f32",
                 label="BASELINE.json configs[2]: WaveNet 256 ch, 30 blocks (3 x dilation 1..512), k=2"),
    "cfg2": dict(model="rawctcnet", channels=128, cycles=1, seq_len=4096, batch=32, precision="bf16",
                 label="BASELINE.json configs[1]: RawCTCNet 128 ch, 10 blocks (dilation 1..512) + input block, "
                       "feature conv k=3, 5 labels, non-causal"),
    "cfg5": dict(model="wavenet", channels=512, cycles=6, seq_len=48000, batch=2, precision="f16",
                 label="BASELINE.json configs[4] shape on ONE GPU: WaveNet 512 ch, 60 blocks (6 x dilation 1..512), k=2"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS),
                    help="cfg3 = BASELINE configs[2] (the metric's config, default); cfg2 = configs[1] (RawCTCNet 128 ch); "
                         "cfg5 = configs[4]'s shape on one GPU (512 ch x 60 blocks x 48000)")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3", "f16", "bf16"],
                    help="arithmetic of the block stack; default: the config's (cfg3: f32)")
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--cycles", type=int, default=None)
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU")
    ap.add_argument("--levels-input", action="store_true",
                    help="WaveNet only: feed the quantised levels [B, L] to the entry conv as an embedding gather "
                         "(WaveNet.forward_levels) instead of a dense one-hot [B, 256, L]")
    ap.add_argument("--init", default=None, choices=["reference", "conditioned"],
                    help="weights: the reference's init rules (default), or the same with the residual path conditioned like a "
                         "trained network (proj ~ I, conv1x1_residual x0.3; default for cfg5 in f16: 60 random-init blocks "
                         "amplify the residual stream by ~2^30, beyond any 16-bit format)")
    ap.add_argument("--no-second-line", action="store_true",
                    help="cfg3/f32 only: skip the additional f16x3 measurement reported under \"split_precision\"")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a HIP graph (wavenet_speech_amd.GraphedStep: forward + backward + gradient gather in one "
                         "graph, the all-reduce eager, the optimizer in a second graph).  auto = on for cfg2, whose ~280 launches of "
                         "5-50 us take the host longer to issue than the GPU to run; off for the long-kernel configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seq-len", type=int, default=None)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the extra forward-only / forward+backward runs after the timed region (used when profiling, so "
                         "that rocprofv3's per-kernel shares are those of the timed steps)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    for k in ("channels", "cycles", "seq_len", "batch", "precision"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    if args.cpu_seq_len is None:
        args.cpu_seq_len = args.seq_len
    args.use_graph = args.graph == "on" or (args.graph == "auto" and args.config == "cfg2")
    if args.init is None:
        args.init = "conditioned" if (args.config == "cfg5" and args.precision in ("f16", "f16x3")) else "reference"
    return args


def make_layers(channels, cycles):
    return [(channels, channels, 2, 2 ** i) for _ in range(cycles) for i in range(10)]


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def die(code, msg):
    print("bench.py: ERROR: " + msg, file=sys.stderr, flush=True)
    sys.exit(code)


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


# ----------------------------------------------------------------------------------------------------------------------
# multi-GPU launch
# ----------------------------------------------------------------------------------------------------------------------
def visible_gpus():
    import torch
    return torch.cuda.device_count()   # counts devices without creating a HIP context


def share_one_gpu():
    """TEST HOOK (tests/test_gpu_driver_contract.py): WN_BENCH_SHARE_GPU=1 lets every rank use GPU 0 and replaces RCCL by gloo
    (RCCL refuses two ranks on one device), so that the launcher, the process group and every collective call site of this
    file run with world size 2 on a one-GPU box.  The JSON line says so (`ranks.backend`); never set it for a measurement."""
    return os.environ.get("WN_BENCH_SHARE_GPU") == "1"


def self_launch(args):
    """--gpus N > 1 with no launcher around us: become the launcher.  Nothing has touched the GPU yet (the ranks are
    fresh child processes; this process never initialises HIP)."""
    n = visible_gpus()
    if n < args.gpus and not (share_one_gpu() and n >= 1):
        die(3, "--gpus %d requested but only %d GPU(s) are visible; refusing to run on fewer" % (args.gpus, n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env["WN_BENCH_SELF_LAUNCHED"] = "1"
    log("self-launch: " + " ".join(cmd))
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in child.stdout.splitlines() if l.strip().startswith("{")]
    if child.returncode != 0 or not lines:
        sys.stderr.write(child.stdout)
        die(child.returncode or 4, "the %d-rank child run failed (rc %d, %d JSON lines)" % (args.gpus, child.returncode, len(lines)))
    print(lines[-1], flush=True)
    sys.exit(0)


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, budget_s=12.0):
    """Oracle (reference-equivalent ATen CPU ops) fwd+bwd at batch 1 on a bounded sample of the workload:
    the full block stack at a sequence length chosen so one iteration takes ~budget_s; cost is linear in L,
    so samples/s of the full utterance = (L_sample / seq_len) / t."""
    import torch
    from oracle import wavenet_oracle as O
    channels, cycles, seq_len = args.channels, args.cycles, args.cpu_seq_len
    layers = make_layers(channels, cycles)
    ncores = host_cores()
    torch.set_num_threads(ncores)
    g = torch.Generator().manual_seed(1234)
    if args.model == "rawctcnet":
        sd = O.random_rawctcnet_state(channels, 3, 5, layers, channels, seed=0)
    else:
        sd = O.random_wavenet_state(channels, 2, layers, channels, seed=0)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}

    def run(L):
        if args.model == "rawctcnet":
            x = torch.randn(1, 1, L, generator=g)
            cot = torch.randn(1, 5, L + 2, generator=g)
        else:
            q = torch.randint(0, channels, (1, L), generator=g)
            x = O.one_hot_encoding(q, channels)
            cot = torch.randn(1, channels, L, generator=g)
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        if args.model == "rawctcnet":
            y = O.raw_ctcnet(x, sd, layers, 3, 1, False, False, False, impl="aten")
        else:
            y = O.wavenet(x, sd, layers, False, impl="aten")
        (y * cot).sum().backward()
        return time.perf_counter() - t0

    probe_len = min(seq_len, 1024)
    run(probe_len)                       # warm-up (thread pool, oneDNN primitives)
    t_probe = run(probe_len)
    log("cpu baseline probe: %d steps in %.2f s on %d threads" % (probe_len, t_probe, ncores))
    L_s = int(min(seq_len, max(probe_len, budget_s / (t_probe / probe_len))))
    times = []
    for _ in range(4 if L_s * t_probe / probe_len > 2.0 else 6):
        times.append(run(L_s))
        log("cpu baseline: %d steps in %.2f s" % (L_s, times[-1]))
    best = min(times)
    return {"value": (L_s / float(seq_len)) / best, "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": "oracle (ATen conv1d/einsum, fp32, torch %s) fwd+bwd, batch 1, %s %d ch x %d blocks, %d of %d time "
                      "steps per iteration (cost linear in L), best of %d timed iterations after warm-up: %.2f s"
                      % (torch.__version__, args.model, channels, len(layers), L_s, seq_len, len(times), best)}


# timing classes of the library (wn_prof_*) -> the code-object symbol class rocprofv3 reports them under.  The four block
# GEMMs and the conv / skips_sum launches are instantiations of ONE template; res, dx, skips_sum and the plain convs share the
# EPI_LINEAR instantiation, so rocprofv3 --stats lists them as one kernel.
SYMBOL_OF = {
    "series_gemm_kernel<res>": "linear", "series_gemm_kernel<dx>": "linear", "series_gemm_kernel<conv_fwd>": "linear",
    "series_gemm_kernel<conv_bwd_data>": "linear", "series_gemm_kernel<skips_sum>": "linear",
    "series_gemm_kernel<gate>": "gate", "series_gemm_kernel<dz,dgate>": "dgate", "wgrad_kernel": "wgrad",
    "hgemm_kernel<gate>": "hgate", "hgemm_kernel<res>": "hstore", "hgemm_kernel<dz,dgate>": "hdgate",
    "hgemm_kernel<dx>": "hstore", "hgemm_kernel<skips_sum>": "hf32", "hgemm_kernel<conv_fwd>": "hf32",
    "hgemm_kernel<conv_bwd_data>": "hf32", "hwgrad_kernel": "hwgrad", "hfused_fwd_kernel": "hfused",
    "hcol_kernel<dz,dgate>": "hcoldz", "hcol_kernel<dx>": "hcoldx", "hcol2_kernel<dx+dz>": "hcol2", "hcol_kernel<skips_sum>": "hcolskip",
}
# class -> regex on the demangled kernel name.  Template arguments: series_gemm_kernel<MT, NT, EPI, PFB, WPS>,
# hgemm_kernel<MT, P, BF, EPI>, hgemm8_kernel<P, BF, EPI, NT, WR>, hfused_fwd_kernel<BF, NZT, MODE>, hcol_kernel<BF, NT, NKS, EPI>, hcol2_kernel<BF, NT, HASDR>
SYMBOL_RE = {"linear": r"series_gemm_kernel<\d+, \d+, 0,", "gate": r"series_gemm_kernel<\d+, \d+, 1,",
             "dgate": r"series_gemm_kernel<\d+, \d+, 2,", "wgrad": r"(?<![a-z])wgrad_kernel<",
             "hstore": r"hgemm_kernel<\d+, \d+, \w+, 0>|hgemm8_kernel<\d+, \w+, 0,", "hgate": r"hgemm_kernel<\d+, \d+, \w+, 1>|hgemm8_kernel<\d+, \w+, 1,",
             "hdgate": r"hgemm_kernel<\d+, \d+, \w+, 2>|hgemm8_kernel<\d+, \w+, 2,",
             "hf32": r"hgemm_kernel<\d+, \d+, \w+, 3>|hgemm8_kernel<\d+, \w+, 3,",
             "hwgrad": r"hwgrad_kernel<", "hfused": r"hfused_fwd_kernel<",
             "hcoldz": r"hcol_kernel<\w+, \d+, \d+, 2>", "hcoldx": r"hcol_kernel<\w+, \d+, \d+, 0>", "hcol2": r"hcol2_kernel<", "hcolskip": r"hcol_kernel<\w+, \d+, \d{2,3}, 100>"}
# half block kernels: channel vectors (of C elements) moved per (utterance, time step) by one launch when every operand is read
# once and every result written once: (always, extra when the model has a skip path -- every reference model has one).
# hfused (training): x in; z, sigmoid(g), r out.  dz: dr, sigmoid(g), z in [+ the gradient of skips_sum]; da|dg out.
# dx: da|dg, dr in; dx out.  hgate: x in; sigmoid(g), z out.
# hwgrad: x, da|dg, z, dr in (the weight gradients themselves are negligible), per block.
ALG_CHANNELS = {"hfused": (4, 0), "hdgate": (5, 1), "hstore": (4, 0), "hgate": (3, 0), "hwgrad": (5, 0),
                "hcoldz": (5, 1), "hcoldx": (4, 0), "hcol2": (8, 1)}   # hcol2: da|dg, dr in, dx out; dS, sigmoid(g), z in, da|dg out
SYMBOL_NOTE = {"linear": "EPI_LINEAR: res, dx, skips_sum, conv launches", "gate": "EPI_GATE", "dgate": "EPI_DGATE: dz", "wgrad": "",
               "hstore": "HEPI_STORE: res, dx", "hgate": "HEPI_GATE", "hdgate": "HEPI_DGATE: dz", "hf32": "HEPI_F32: skips_sum, convs",
               "hwgrad": "", "hfused": "gate -> z -> res [+ skip] in one launch",
               "hcoldz": "dz + dgate, column-owner streaming form", "hcoldx": "dx, column-owner streaming form",
               "hcol2": "dx of a block + dz of the block below it in one launch", "hcolskip": "skips_sum -> leaky series, column-owner form"}


def profile_dir(args):
    """profiles/<round>/<config>_<precision>/ : the committed rocprofv3 summaries of THIS configuration and precision (collected
    by tools/collect_profiles.sh); other configurations' counters are never quoted."""
    return os.path.join(ROOT, "profiles", PROFILE_ROUND, "%s_%s" % (args.config, args.precision))


def _read_csv(path):
    import csv
    if not os.path.exists(path):
        return None
    return list(csv.DictReader(open(path)))


def instantiation_of(args, symbol):
    """the exact instantiation (demangled name) of a kernel class that dominates this configuration: the row of the class with the
    largest total time in the committed rocprofv3 --stats table of this (config, precision); None without such a table"""
    import glob
    import re
    best = None
    for path in sorted(glob.glob(os.path.join(profile_dir(args), "kernel_stats*.csv"))):
        for row in _read_csv(path) or []:
            if symbol in SYMBOL_RE and re.search(SYMBOL_RE[symbol], row["Name"]):
                t = float(row["TotalDurationNs"])
                if best is None or t > best[0]:
                    best = (t, row["Name"], float(row["AverageNs"]) * 1e-6, os.path.relpath(path, ROOT))
    return best


def pmc_lookup(args, name, exact_kernel, column):
    """`column` of the row whose kernel name equals `exact_kernel` in profiles/<round>/<config>_<precision>/<name>; (None, None) if the
    file or the row is missing"""
    path = os.path.join(profile_dir(args), name)
    rows = _read_csv(path)
    if not rows or exact_kernel is None:
        return None, None
    total, hit = 0.0, False
    for row in rows:
        if row["kernel"].strip() == exact_kernel.strip():
            total += float(row[column])
            hit = True
    return (total if hit else None), (os.path.relpath(path, ROOT) if hit else None)


def main():
    args = parse()
    args.model = CONFIGS[args.config]["model"]
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        die(2, "--gpus must be >= 1")
    if not launched and args.gpus > 1:
        self_launch(args)                     # does not return
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if world != args.gpus:
        die(2, "--gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # RCCL / MIOpen print banners on stdout; the contract is ONE JSON line there.  Send fd 1 to stderr for the run
    # and restore it only to print the result.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    ndev = visible_gpus()
    backend = "nccl"
    if share_one_gpu():
        local_rank, backend = 0, "gloo"
    if ndev <= local_rank or ndev < 1:
        die(3, "rank %d needs GPU %d but %d GPU(s) are visible (the HIP path has no CPU fallback)" % (rank, local_rank, ndev))
    if not torch.cuda.is_available():
        die(3, "bench.py needs a GPU (the HIP path has no CPU fallback)")
    # under torch.distributed.run the RCCL group is always created, even for one rank, so the same code path
    # (init, barrier, flat-gradient all-reduce, MAX over ranks) runs at N=1 and N=8
    distributed = launched and "MASTER_PORT" in os.environ
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            die(2, "RCCL group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    import wavenet_speech_amd as W
    from wavenet_speech_amd import functional as HF
    from wavenet_speech_amd.parallel import FlatGradAllReduce

    C, L, B = args.channels, args.seq_len, args.batch
    layers = make_layers(C, args.cycles)
    torch.manual_seed(0)  # identical replicas
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)  # each rank its own shard of the global batch
    levels = None
    if args.model == "rawctcnet":
        from wavenet_speech_amd.modules.raw_ctcnet import RawCTCNet
        net = RawCTCNet(C, 3, 5, layers, C, softmax=False, causal=False).to(dev)
        x = torch.randn(B, 1, L, generator=g).to(dev)
        cot = torch.randn(B, 5, L + 2, generator=g).to(dev)
        nblk = len(layers) + 1
    else:
        from wavenet_speech_amd.modules.wavenet import WaveNet
        net = WaveNet(C, 2, layers, C, softmax=False).to(dev)
        q = torch.randint(0, C, (B, L), generator=g)
        if args.levels_input:
            levels = q.to(dev)
            x = None
        else:
            x = torch.zeros(B, C, L, device=dev).scatter_(1, q.to(dev).unsqueeze(1), 1.0)
        cot = torch.randn(B, C, L, generator=g).to(dev)
        nblk = len(layers)
    if args.init == "conditioned":
        with torch.no_grad():
            for blk in net.convolutions:
                blk.residual_proj.weight.copy_(torch.eye(blk.out_channels, blk.in_channels, device=dev)
                                               + 0.02 * torch.randn(blk.out_channels, blk.in_channels, device=dev))
                blk.conv1x1_residual.weight.mul_(0.3)
    W.set_precision(net, args.precision)
    nparams = sum(p.numel() for p in net.parameters())
    try:
        # same update rule, one multi-tensor kernel; capturable = the step count lives on the device (needed inside a HIP graph)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True, capturable=bool(args.use_graph))
    except (TypeError, RuntimeError):
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    sync = FlatGradAllReduce(net.parameters())
    ar_events = []

    def forward():
        return net.forward_levels(levels) if levels is not None else net(x)

    def step(timed=False):
        sync.zero()
        out = forward()
        (out * cot).sum().backward()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sync.reduce()
            e1.record()
            ar_events.append((e0, e1))
        else:
            sync.reduce()
        opt.step()

    gstep = None

    def graphed_step(timed=False):
        gstep.replay_forward_backward()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gstep.reduce()
            e1.record()
            ar_events.append((e0, e1))
        else:
            gstep.reduce()
        gstep.step_optimizer()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log("model built: %s, %d params, precision %s; warm-up" % (args.model, nparams, args.precision))
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    fence()
    timing = not args.no_kernel_timing
    if args.use_graph:
        # the timed steps are graph replays: the same launches as the eager step, issued by one host call.  Per-kernel HIP
        # events cannot be recorded inside a replay, so the kernel table comes from eager steps AFTER the timed region.
        gstep = W.GraphedStep(lambda: (forward() * cot).sum(), net.parameters(), optimizer=opt, sync=sync, warmup=1)
        for i in range(max(1, args.warmup)):
            graphed_step()
        fence()
        log("HIP graph captured; %d replayed warm-up step(s) done" % max(1, args.warmup))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graphed_step(timed=True)
        fence()
        elapsed = time.perf_counter() - t0
        gstep.check()
    else:
        if timing:
            HF.profile_reset()
            HF.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=True)
        fence()
        elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    log("timed %d steps: %.2f ms/step" % (args.steps, elapsed / args.steps * 1e3))
    kern = {}
    eager_ms = None
    if timing and args.use_graph:
        HF.profile_reset()
        HF.profile_enable(True)
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        eager_ms = (time.perf_counter() - t2) / args.steps * 1e3
        log("eager steps for the kernel table: %.2f ms/step" % eager_ms)
    if timing:
        HF.profile_enable(False)
        kern = HF.profile_read()
    allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(1, len(ar_events))

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
            forward()

    def fwd_bwd():
        sync.zero()
        (forward() * cot).sum().backward()

    breakdown = {"peak_hbm_gib_allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}
    if not args.no_breakdown:
        breakdown.update({"fwd_only_ms": round(median_ms(fwd_only), 2), "fwd_bwd_ms": round(median_ms(fwd_bwd), 2)})
        breakdown["fwd_bwd_samples_per_s_per_gpu"] = round(B / (breakdown["fwd_bwd_ms"] * 1e-3), 2)
    # ---- second, separately labelled measurement: the same step in the f16x3 mode (three-product fp16 split) --------
    second = None
    if args.config == "cfg3" and args.precision == "f32" and not args.no_second_line:
        with torch.no_grad():
            y32 = forward()
        W.set_precision(net, "f16x3")
        with torch.no_grad():
            y3 = forward()
        err = float((y3 - y32).abs().max() / y32.abs().max())
        del y32, y3
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        if timing:
            HF.profile_reset()
            HF.profile_enable(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el2 = time.perf_counter() - t1
        kern2 = {}
        if timing:
            HF.profile_enable(False)
            kern2 = HF.profile_read()
        if distributed:
            t = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        W.set_precision(net, "f32")
        hk = {k: v for k, v in kern2.items() if k.startswith("h") and v[1] > 0 and v[2] > 0}
        dom2 = max(hk, key=lambda k: hk[k][0]) if hk else None
        second = {"label": "same workload and step, residual stack in precision f16x3 (3-product fp16 split on "
                           "v_mfma_f32_32x32x16_f16, fp32 accumulate); NOT the headline value",
                  "dtype": "f16x3", "value": round(world * B * args.steps / el2, 3), "unit": "samples/s",
                  "ms_per_step": round(el2 / args.steps * 1e3, 2),
                  "forward_rel_err_vs_f32_path": err,
                  "error_note": "max-norm relative difference of the two modes' outputs on this run's input and weights.  With the "
                                "reference init (the default here) the 30-block map is ill-conditioned: two CPU fp32 evaluations of "
                                "it differ by 4.5e-4 and each is ~3e-4 from fp64 (DESIGN.md section 2).  On a conditioned model the "
                                "f16x3 path is 6e-6 (forward) / 5e-5 (worst gradient) from the oracle: "
                                "tests/test_gpu_half.py::test_f16x3_cfg3_one_utterance_vs_oracle",
                  "kernels": {k: {"avg_ms": round(v[0] / v[1], 4), "algorithmic_tflops": round(v[2] / (v[0] * 1e-3) / 1e12, 1)}
                              for k, v in hk.items()}}
        if dom2:
            ms2, n2, fl2 = hk[dom2]
            ach2 = 3.0 * fl2 / (ms2 * 1e-3) / 1e12
            second["roofline"] = {"kernel": dom2, "bound": "mfma", "achieved": round(ach2, 1), "peak": PEAK_HALF_MFMA_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach2 / PEAK_HALF_MFMA_TFLOPS, 4),
                                  "frac_of_sustained": round(ach2 / SUSTAINED_HALF_MFMA_TFLOPS, 4), "sustained_note": SUSTAINED_NOTE,
                                  "note": "executed fp16 MFMA flops (3 per algorithmic product) / HIP-event kernel time"}
    per_rank_ms = [round(own_elapsed / args.steps * 1e3, 2)]
    per_rank_ar = [round(allreduce_ms, 3)]
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own_elapsed / args.steps * 1e3, allreduce_ms], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(e[0]), 2) for e in every]
        per_rank_ar = [round(float(e[1]), 3) for e in every]

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- roofline ------------------------------------------------------------------------------------------
    half = args.precision != "f32"
    nprod = 3 if args.precision == "f16x3" else 1          # MFMA products executed per algorithmic product
    peak = PEAK_HALF_MFMA_TFLOPS if half else PEAK_FP32_MFMA_TFLOPS
    esz = 2 if args.precision in ("f16", "bf16") else 4     # bytes per stored activation element
    L_eff = L + 2 if args.model == "rawctcnet" else L
    units = float(B) * L_eff * nblk                    # (b, t, block) units per step per GPU
    alg_flops_step = 48.0 * C * C * units              # 48 C^2 flop per (b,t,block) fwd+bwd
    alg_bytes_step = 8.0 * C * esz * units             # 8 C s bytes per (b,t,block)
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
        ach = nprod * fl / (ms * 1e-3) / 1e12
        inst = instantiation_of(args, dom)                      # (total ns, exact demangled name, avg ms, source) or None
        exact = inst[1] if inst else None
        traffic, traffic_src = pmc_lookup(args, "pmc_hbm_traffic.csv", exact, "avg_bytes_corrected")
        busy, busy_src = pmc_lookup(args, "pmc_mfma.csv", exact, "mfma_busy_frac")

        def label(k):
            note = SYMBOL_NOTE.get(k, "")
            i = instantiation_of(args, k)
            return (i[1] if i else k) + ("  [" + note + "]" if note else "")
        roofline = {"kernel": label(dom), "bound": "mfma", "achieved": round(ach, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch of exactly this instantiation: rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + "
                                    "WRITE_SIZE, separate passes, from %s (committed summary of tools/collect_profiles.sh for THIS "
                                    "configuration and precision; null when no such capture is committed -- not collected live)" % traffic_src,
                    "mfma_busy_frac": busy, "mfma_busy_src": busy_src,
                    "rocprof_avg_launch_ms": round(inst[2], 4) if inst else None, "rocprof_src": inst[3] if inst else None,
                    "avg_launch_ms": round(ms / n, 4), "launches": n, "flops_per_launch": fl / n,
                    "mfma_products_per_algorithmic_product": nprod,
                    "share_of_step": round(ms / args.steps / (step_s * 1e3), 4),
                    "other_symbols": {label(k):
                                      {"avg_launch_ms": round(v[0] / v[1], 4),
                                       "tflops": round(nprod * v[2] / (v[0] * 1e-3) / 1e12, 2),
                                       "frac": round(nprod * v[2] / (v[0] * 1e-3) / 1e12 / peak, 4),
                                       "share_of_step": round(v[0] / args.steps / (step_s * 1e3), 4)}
                                      for k, v in by_symbol.items() if k != dom}}
        # which roof binds this kernel?  Its own algorithmic HBM bytes per launch (every operand read once, every result
        # written once: ALG_CHANNELS[symbol] channel vectors of C elements per (utterance, time step))
        # over the same HIP-event duration, against 8 TB/s -- next to the MFMA fraction above.  The larger fraction is the bound.
        if half and dom in ALG_CHANNELS:
            has_skip = 1
            alg_b = (ALG_CHANNELS[dom][0] + has_skip * ALG_CHANNELS[dom][1]) * C * esz * float(B) * L_eff
            if dom == "hwgrad":
                # its launches cover several blocks each (and the convs around the stack, whose bytes are NOT counted here: a
                # lower bound): the blocks' operand bytes of a step, spread over the launches of a step
                alg_b = alg_b * nblk / (n / float(args.steps))
            gbs = alg_b / (ms / n * 1e-3) / 1e9
            roofline["mfma"] = {"achieved": roofline["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": roofline["frac"]}
            roofline["hbm"] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "algorithmic_bytes_per_launch": alg_b,
                               "measured_traffic_gbs": round(traffic / (ms / n * 1e-3) / 1e9, 1) if traffic else None}
            if gbs / PEAK_HBM_GBS > ach / peak:
                roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(gbs / PEAK_HBM_GBS, 4)})
        if traffic is not None:
            # algorithmic bytes per (utterance, time step, block) = 8 C s for the whole fwd+bwd of a block
            roofline["traffic_vs_block_algorithmic_bytes"] = round(traffic / (8.0 * C * esz * float(B) * L_eff), 3)
        if peak == PEAK_HALF_MFMA_TFLOPS:
            roofline["frac_of_sustained"] = round(ach / SUSTAINED_HALF_MFMA_TFLOPS, 4)
            roofline["sustained_note"] = SUSTAINED_NOTE
    kernel_ms = sum(v[0] for v in kern.values())
    roofline_step = {
        "algorithmic_tflops": round(alg_flops_step / step_s / 1e12, 2),
        "mfma_frac": round(nprod * alg_flops_step / step_s / 1e12 / peak, 4),
        "algorithmic_hbm_gbs": round(alg_bytes_step / step_s / 1e9, 1),
        "hbm_frac": round(alg_bytes_step / step_s / 1e9 / PEAK_HBM_GBS, 4),
        "hip_kernel_ms_per_step": round(kernel_ms / args.steps, 2) if kern else None,
        "accounting": "48*C^2 flop and 8*C*%d B per (utterance, time step, block), SURVEY.md 8(d); mfma_frac counts %d MFMA "
                      "product(s) per algorithmic product against the %.1f TFLOP/s dense peak" % (esz, nprod, peak),
    }

    dtype = {"f32": "f32", "f16x3": "f16x3 (3-product fp16 split of 22-bit operands, fp32 accumulate; within 1e-4 of the f32 path on conditioned models)",
             "f16": "f16", "bf16": "bf16"}[args.precision]
    headline = args.config == "cfg3" and (C, args.cycles, L, B) == tuple(CONFIGS["cfg3"][k] for k in
                                                                         ("channels", "cycles", "seq_len", "batch"))
    metric = ("samples/sec fwd+bwd, 256-ch 30-block WaveNet @16k seq" if headline else
              "samples/sec fwd+bwd, %d-ch %d-block %s @%d seq" % (C, nblk, args.model, L))
    result = {
        "metric": metric,
        "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": "%s, seq_len %d, batch %d per GPU, %s, full step (fwd+bwd+grad all-reduce+Adam)"
                               % (CONFIGS[args.config]["label"] if headline or args.config != "cfg3" else
                                  "WaveNet %d ch, %d blocks" % (C, nblk), L, B, args.precision),
                   "channels": C, "blocks": nblk, "seq_len": L, "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": "dp%d" % world, "params": nparams,
                   "input": "levels (embedding gather)" if levels is not None else
                            ("raw signal" if args.model == "rawctcnet" else "dense one-hot")},
        "per_gpu": round(value / world, 3),
        "ranks": {"rccl_ranks": dist.get_world_size() if distributed else 0, "backend": backend if distributed else None,
                  "launcher": "self" if os.environ.get("WN_BENCH_SELF_LAUNCHED") else ("external" if launched else "none"),
                  "ms_per_step_by_rank": per_rank_ms, "grad_allreduce_ms_by_rank": per_rank_ar,
                  "grad_allreduce_payload_mb": round(sync.payload_bytes() / 1e6, 1)},
        "breakdown": breakdown, "roofline": roofline, "roofline_step": roofline_step, "kernels": kernels,
        "launch": ({"mode": "hipgraph", "note": "timed steps are replays of a captured HIP graph (forward + backward + gradient gather; "
                    "all-reduce eager; optimizer in a second graph): the same kernels as the eager step, one host call.  The kernel "
                    "table and the roofline object were timed with HIP events over the same number of EAGER steps after the timed region",
                    "eager_ms_per_step": round(eager_ms, 2) if eager_ms is not None else None}
                   if args.use_graph else {"mode": "eager"}),
    }
    if second is not None:
        result["split_precision"] = second
    result["config"]["init"] = args.init
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        del net, opt, sync, x, cot
        torch.cuda.empty_cache()
        result["cpu_baseline"] = cpu_baseline(args)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    # leave nothing running (one process outlived BENCH_r02): this process starts no children at N = 1; say so, and
    # end any a library may have started behind our back
    try:
        import psutil
        kids = psutil.Process().children(recursive=True)
        log("child processes at exit: %d%s" % (len(kids), (" " + str([k.name() for k in kids])) if kids else ""))
        for k in kids:
            k.terminate()
    except Exception as e:   # psutil missing: nothing to report
        log("child-process check skipped: %s" % e)


if __name__ == "__main__":
    main()
