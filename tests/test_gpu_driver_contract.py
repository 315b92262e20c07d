"""GPU: the driver-facing contract -- bench.py prints ONE JSON line with the agreed keys (tiny configuration here),
also when launched through torch.distributed.run; __graft_entry__.smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--channels", "32", "--cycles", "1", "--seq-len", "1024", "--batch", "2", "--steps", "2", "--warmup", "1"]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _check(line, n_gpus, expect_cpu):
    d = json.loads(line)
    missing = REQUIRED - set(d) - (set() if expect_cpu else {"cpu_baseline"})
    assert not missing, missing
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - n_gpus * 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if expect_cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_bench_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TINY + ["--cpu-seq-len", "512"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    _check(lines[0], 1, True)


def test_bench_under_torch_distributed_run_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1"] + TINY + ["--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    _check(lines[0], 1, False)


def test_bench_self_launches_two_ranks_and_relays_one_line():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the ranks itself.  On this one-GPU box the two ranks share
    GPU 0 over gloo (WN_BENCH_SHARE_GPU=1, a test hook: RCCL refuses two ranks on one device) -- the launcher, the process group,
    barrier, flat-gradient all-reduce, MAX over ranks, per-rank gather and the JSON relay all run with world size 2."""
    env = dict(os.environ, WN_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TINY + ["--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    r = d["ranks"]
    assert r["rccl_ranks"] == 2 and r["launcher"] == "self" and r["backend"] == "gloo"
    assert len(r["ms_per_step_by_rank"]) == 2 and len(r["grad_allreduce_ms_by_rank"]) == 2
    assert abs(d["value"] - 2 * 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "split_precision" in d and d["split_precision"]["value"] > 0      # the second line also runs under DP


def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
