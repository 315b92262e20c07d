#!/usr/bin/env python3
"""Exploration only (nothing here ships): how accurate would the block stack be if every fp32 GEMM were emulated by
split half-precision MFMA products with fp32 accumulation?   x = hi + lo (both fp16 or bf16), W likewise,
W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi   (3 products at 16x the fp32 MFMA rate).
Runs the conditioned 30-block stack on the CPU with the GEMM operands rounded accordingly and compares against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import wavenet_oracle as O
torch.set_num_threads(8)
C, L, NB = 64, 3000, 30
layers = [(C, C, 2, 2 ** (i % 10)) for i in range(NB)]

def split(t, dt, scale):
    hi = (t * scale).to(dt).to(torch.float32)
    lo = ((t * scale) - hi).to(dt).to(torch.float32)
    return hi / scale, lo / scale

def mm(W, x, mode):
    """W [O,C] @ x [B,C,L] under the emulated arithmetic"""
    if mode == "fp32":
        return torch.einsum("oc,bcl->bol", W, x)
    dt = torch.float16 if mode.startswith("f16") else torch.bfloat16
    ws = 256.0 if dt == torch.float16 else 1.0          # power-of-two pre-scale keeps the fp16 lo parts normal
    Wh, Wl = split(W, dt, ws)
    xh, xl = split(x, dt, 1.0)
    y = torch.einsum("oc,bcl->bol", Wh, xh)
    if mode.endswith("x3"):
        y = y + torch.einsum("oc,bcl->bol", Wh, xl) + torch.einsum("oc,bcl->bol", Wl, xh)
    return y

def stack(x, sd, mode, dtype=torch.float32):
    S = torch.zeros(x.shape[0], C, x.shape[2], dtype=dtype)
    out = x
    for l, (_, _, k, d) in enumerate(layers):
        p = O.block_params(sd, "convolutions.%d." % l)
        offs = O.tap_offsets(k, d, True)
        f = (lambda W, t: torch.einsum("oc,bcl->bol", W, t)) if dtype == torch.float64 else (lambda W, t: mm(W, t, mode))
        a = sum(f(p["conv_tanh.conv1d.weight"][:, :, j], O.shifted(out, o)) for j, o in enumerate(offs)) + p["conv_tanh.conv1d.bias"].view(1, -1, 1)
        g = sum(f(p["conv_sigmoid.conv1d.weight"][:, :, j], O.shifted(out, o)) for j, o in enumerate(offs)) + p["conv_sigmoid.conv1d.bias"].view(1, -1, 1)
        z = torch.tanh(a) * torch.sigmoid(g)
        r = f(p["conv1x1_residual.weight"][:, :, 0], z) + f(p["residual_proj.weight"], out) + (p["conv1x1_residual.bias"] + p["residual_proj.bias"]).view(1, -1, 1)
        s = f(p["conv1x1_skip.weight"][:, :, 0], z) + p["conv1x1_skip.bias"].view(1, -1, 1)
        S = S + f(sd["bottlenecks.%d.weight" % l][:, :, 0], s) + sd["bottlenecks.%d.bias" % l].view(1, -1, 1)
        out = r
    return S

g = torch.Generator().manual_seed(0)
sd = O.random_wavenet_state(C, 2, layers, C, seed=0)
for l in range(NB):
    pre = "convolutions.%d." % l
    sd[pre + "residual_proj.weight"] = torch.eye(C) + 0.02 * torch.randn(C, C, generator=g)
    sd[pre + "conv1x1_residual.weight"] = sd[pre + "conv1x1_residual.weight"] * 0.3
x = torch.randn(1, C, L, generator=g)
with torch.no_grad():
    ref = stack(x.double(), {k: v.double() for k, v in sd.items()}, "fp64", torch.float64)
    for mode in ("fp32", "f16x3", "bf16x3", "f16x1", "bf16x1"):
        y = stack(x, sd, mode)
        print("%-7s skips_sum vs fp64: max-norm rel err %.2e" % (mode, O.rel_err(y.double(), ref)))
