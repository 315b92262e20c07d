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

def planes(t, dt, scale, n):
    """t ~ sum of n planes of dtype dt (each the rounded remainder of the previous ones), as fp32 tensors"""
    out, rem = [], t * scale
    for _ in range(n):
        p = rem.to(dt).to(torch.float32)
        out.append(p / scale)
        rem = rem - p
    return out

# round 3: is there a split that is NOT narrower than fp32?  name -> (weight planes, activation planes, products)
# a product (i, j) multiplies weight plane i by activation plane j; cost = number of products / 16 of the fp32 MFMA cost
SPLITS = {
    "f16x3": (2, 2, [(0, 0), (0, 1), (1, 0)]),                                 # shipped: hi*hi + hi*lo + lo*hi  (22 bits)
    "f16x4": (2, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),                         # + lo*lo
    "f16x6": (3, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),         # three planes each (33 bits), products down to 2^-22
    "f16w3x2": (3, 2, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]),               # weights in three planes, activations in two
    "f16w2x3": (2, 3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2)]),               # activations in three planes, weights in two
}

def mm(W, x, mode):
    """W [O,C] @ x [B,C,L] under the emulated arithmetic (fp32 accumulation: products are exact in fp32, the sum is torch's)"""
    if mode == "fp32":
        return torch.einsum("oc,bcl->bol", W, x)
    if mode in SPLITS:
        nw, nx, prods = SPLITS[mode]
        Wp = planes(W, torch.float16, 256.0, nw)          # power-of-two pre-scale keeps the fp16 low planes normal
        xp = planes(x, torch.float16, 1.0 / 16.0, nx)      # activations stored as x / 16 like the shipped path (|r| reaches 7e4 with the reference init)
        return sum(torch.einsum("oc,bcl->bol", Wp[i], xp[j]) for i, j in prods)
    dt = torch.float16 if mode.startswith("f16") else torch.bfloat16
    ws = 256.0 if dt == torch.float16 else 1.0
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

def model(conditioned):
    g = torch.Generator().manual_seed(0)
    sd = O.random_wavenet_state(C, 2, layers, C, seed=0)
    if conditioned:
        for l in range(NB):
            pre = "convolutions.%d." % l
            sd[pre + "residual_proj.weight"] = torch.eye(C) + 0.02 * torch.randn(C, C, generator=g)
            sd[pre + "conv1x1_residual.weight"] = sd[pre + "conv1x1_residual.weight"] * 0.3
    return sd, torch.randn(1, C, L, generator=g)


def f32_other_order(W, x):
    """the same fp32 product summed in another association (two halves of K): the spread between two legitimate fp32 results"""
    h = W.shape[1] // 2
    return torch.einsum("oc,bcl->bol", W[:, :h], x[:, :h]) + torch.einsum("oc,bcl->bol", W[:, h:], x[:, h:])


if __name__ == "__main__":
    for conditioned in (True, False):
        sd, x = model(conditioned)
        print("== 30 blocks x %d ch x L %d, %s ==" % (C, L, "conditioned residual path (proj ~ I)" if conditioned else
                                                    "the reference's random init (ill-conditioned: |r| grows ~sqrt(2) per block)"))
        with torch.no_grad():
            ref = stack(x.double(), {k: v.double() for k, v in sd.items()}, "fp64", torch.float64)
            modes = ["fp32"] + list(SPLITS) + ["bf16x3", "f16x1", "bf16x1"]
            for mode in modes:
                y = stack(x, sd, mode)
                cost = {"fp32": "16/16"}.get(mode, "%d/16" % len(SPLITS[mode][2]) if mode in SPLITS else ("3/16" if mode.endswith("x3") else "1/16"))
                print("%-8s MFMA cost %-6s skips_sum vs fp64: max-norm rel err %.2e" % (mode, cost, O.rel_err(y.double(), ref)))
            _mm = mm
            globals()["mm"] = lambda W, t, mode: f32_other_order(W, t)
            y = stack(x, sd, "fp32")
            globals()["mm"] = _mm
            print("%-8s (another summation order)             max-norm rel err %.2e" % ("fp32'", O.rel_err(y.double(), ref)))
