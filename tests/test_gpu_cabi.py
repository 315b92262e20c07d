"""GPU: the C ABI used with NO torch on the device side -- tests/cabi/cabi_block.cpp (plain HIP host code linking
libwavenet_amd.so) runs one residual block forward + backward on hipMalloc'd buffers; this test only prepares the
problem file and checks the result file against the CPU oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def cabi_exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabi") / "cabi_block")
    libdir = os.path.join(ROOT, "wavenet_speech_amd")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_block.cpp"), "-L", libdir, "-lwavenet_amd",
                           "-Wl,-rpath," + libdir, "-Wno-unused-result", "-o", out])
    return out


@pytest.mark.parametrize("case", [(2, 300, 24, 40, 2, 8, 1), (1, 130, 64, 64, 3, 5, 0)])
def test_block_through_plain_c_abi(cabi_exe, tmp_path, case):
    B, L, Ci, Co, k, d, causal = case
    torch.manual_seed(11)
    shapes = [(Co, Ci, k), (Co,), (Co, Ci, k), (Co,), (Co, Co), (Co,), (Co, Co), (Co,), (Co, Ci), (Co,)]
    params = [torch.randn(s) * (0.2 if len(s) > 1 else 0.1) for s in shapes]
    x, cr, cs = torch.randn(B, Ci, L), torch.randn(B, Co, L), torch.randn(B, Co, L)
    prob, res = str(tmp_path / "problem.bin"), str(tmp_path / "result.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("8i", B, L, Ci, Co, k, d, causal, 0))
        for t in params + [x, cr, cs]:
            f.write(t.contiguous().numpy().astype(np.float32).tobytes())
    out = subprocess.run([cabi_exe, prob, res], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    raw = np.fromfile(res, dtype=np.float32)
    # oracle
    names = list(O.BLOCK_KEYS)
    sd = {n: p.clone().requires_grad_(True) for n, p in zip(
        ["conv_tanh.conv1d.weight", "conv_tanh.conv1d.bias", "conv_sigmoid.conv1d.weight", "conv_sigmoid.conv1d.bias",
         "conv1x1_residual.weight", "conv1x1_residual.bias", "conv1x1_skip.weight", "conv1x1_skip.bias",
         "residual_proj.weight", "residual_proj.bias"], params)}
    sd["conv1x1_residual.weight"] = params[4].clone().unsqueeze(2).requires_grad_(True)
    sd["conv1x1_skip.weight"] = params[6].clone().unsqueeze(2).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    r0, s0 = O.residual_block(xo, sd, d, bool(causal))
    ((r0 * cr).sum() + (s0 * cs).sum()).backward()
    pos = 0
    for label, ref in [("r", r0), ("s", s0), ("dx", xo.grad)] + [(n, sd[n].grad) for n in names]:
        want = ref.detach().reshape(-1)                      # 1x1 conv weights: [Co][Co][1] and [Co][Co] flatten alike
        got = torch.from_numpy(raw[pos:pos + want.numel()].copy())
        pos += want.numel()
        assert O.rel_err(got, want) < 1e-4, (label, O.rel_err(got, want))
    assert raw[pos] == 0.0, "padding of an output series is no longer zero"
    assert pos + 1 == raw.size


@pytest.fixture(scope="module")
def cabi_hexe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cabih") / "cabi_hblock")
    libdir = os.path.join(ROOT, "wavenet_speech_amd")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cabi", "cabi_hblock.cpp"), "-L", libdir, "-lwavenet_amd",
                           "-Wl,-rpath," + libdir, "-Wno-unused-result", "-o", out])
    return out


@pytest.mark.parametrize("case", [(2, 300, 24, 40, 2, 8, 1, 1, 1e-4), (1, 130, 64, 64, 3, 5, 0, 1, 1e-4), (2, 260, 32, 32, 2, 4, 1, 3, 8e-2)])
def test_half_block_through_plain_c_abi(cabi_hexe, tmp_path, case):
    """the wn_h* entry points (precision 1 = f16x3 at the fp32 tolerance, 3 = bf16 at its storage error) with no torch on the
    device side; r comes back as the raw half series and is decoded here from the documented layout"""
    B, L, Ci, Co, k, d, causal, prec, tol = case
    torch.manual_seed(12)
    shapes = [(Co, Ci, k), (Co,), (Co, Ci, k), (Co,), (Co, Co), (Co,), (Co, Co), (Co,), (Co, Ci), (Co,)]
    params = [torch.randn(s) * (0.2 if len(s) > 1 else 0.1) for s in shapes]
    x, cr, cs = torch.randn(B, Ci, L), torch.randn(B, Co, L), torch.randn(B, Co, L)
    prob, res = str(tmp_path / "hproblem.bin"), str(tmp_path / "hresult.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("8i", B, L, Ci, Co, k, d, causal, prec))
        for t in params + [x, cr, cs]:
            f.write(t.contiguous().numpy().astype(np.float32).tobytes())
    out = subprocess.run([cabi_hexe, prob, res], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    blob = open(res, "rb").read()
    ld, halo, nbytes, flag = struct.unpack("4i", blob[:16])
    assert flag == 0
    planes = 2 if prec == 1 else 1
    G = (Co + 31) // 32 * 4
    assert nbytes == B * planes * G * ld * 16
    raw16 = np.frombuffer(blob[16:16 + nbytes], dtype=np.uint16).reshape(B, planes, G, ld, 8)
    if prec == 3:
        vals = (raw16.astype(np.uint32) << 16).view(np.float32)
    else:
        vals = raw16.view(np.float16).astype(np.float32)
    assert float(np.abs(vals[:, :, :, :halo]).max(initial=0)) == 0.0 and float(np.abs(vals[:, :, :, halo + L:]).max(initial=0)) == 0.0
    r_got = torch.from_numpy(vals.sum(1)[:, :, halo:halo + L, :].transpose(0, 1, 3, 2).reshape(B, G * 8, L)[:, :Co] * 16.0)
    raw = np.frombuffer(blob[16 + nbytes:], dtype=np.float32)
    names = list(O.BLOCK_KEYS)
    sd = {n: p.clone().requires_grad_(True) for n, p in zip(names, params)}
    sd["conv1x1_residual.weight"] = params[4].clone().unsqueeze(2).requires_grad_(True)
    sd["conv1x1_skip.weight"] = params[6].clone().unsqueeze(2).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    r0, s0 = O.residual_block(xo, sd, d, bool(causal))
    ((r0 * cr).sum() + (s0 * cs).sum()).backward()
    assert O.rel_err(r_got, r0) < tol, ("r", O.rel_err(r_got, r0))
    pos = 0
    for label, ref in [("s", s0), ("dx", xo.grad)] + [(n, sd[n].grad) for n in names]:
        want = ref.detach().reshape(-1)
        got = torch.from_numpy(raw[pos:pos + want.numel()].copy())
        pos += want.numel()
        assert O.rel_err(got, want) < tol, (label, O.rel_err(got, want))
    assert pos == raw.size
