"""Diagnostic: one block through the wn_h* ABI, every intermediate read back and compared with an fp64 evaluation."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wavenet_speech_amd import _lib, functional as HF, functional_half as FH

torch.manual_seed(0)
dev = "cuda:0"
B, C, L, k, d = 2, int(os.environ.get("C", 64)), 300, 2, 4
prec = os.environ.get("PREC", "f16x3")
mode = FH._Mode(prec)
lib = _lib.load()
spec = HF.BlockSpec(C, C, C, k, d, True)
layout = FH.HalfLayout(L, spec.reach())
shape = FH._shape(spec, B, layout)
bound = (6.0 / (C * k)) ** 0.5
P = [((torch.rand(C, C, k) * 2 - 1) * bound), 0.1 * torch.randn(C), ((torch.rand(C, C, k) * 2 - 1) * bound), 0.1 * torch.randn(C),
     ((torch.rand(C, C) * 2 - 1) * bound), 0.1 * torch.randn(C), ((torch.rand(C, C) * 2 - 1) * bound), 0.1 * torch.randn(C),
     ((torch.rand(C, C) * 2 - 1) * bound), 0.1 * torch.randn(C)]
x = torch.randn(B, C, L)
Pd = [p.to(dev).contiguous() for p in P]

def readback(lease, scale):
    t = lease.t.float()                                   # [B][P*G][ld*8]
    G = FH._cp32(C) // 8
    t = t.view(B, mode.planes, G, layout.ld, 8).sum(1)     # hi + lo
    t = t[:, :, layout.halo:layout.halo + L, :].permute(0, 1, 3, 2).reshape(B, G * 8, L)[:, :C]
    return (t / scale).double().cpu()

flag = torch.zeros(1, dtype=torch.int32, device=dev)
rs = lib.wn_hseries_residual_scale()
xin = FH._hlease(mode, B, C, layout, dev)
FH._load(lib, mode, x.to(dev), xin, layout, rs, None, flag)
print("x round trip      ", float((readback(xin, rs) - x.double()).abs().max() / x.abs().max()))
nbytes = lib.wn_hblock_packed_bytes(ctypes.byref(shape), mode.code)
packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
ps = HF._params_struct(Pd)
_lib.check(lib.wn_hblock_pack(ctypes.byref(shape), mode.code, ctypes.byref(ps), HF._p(packed), HF._stream()), "pack")
r, sg, z = (FH._hlease(mode, B, C, layout, dev) for _ in range(3))
S = torch.empty(B, C, L, device=dev)
_lib.check(lib.wn_hblock_forward(ctypes.byref(shape), mode.code, HF._p(packed), HF._p(xin), HF._p(r), HF._p(S), 0, HF._p(sg),
                                 HF._p(z), HF._p(flag), HF._stream()), "fwd")
torch.cuda.synchronize()
xd = x.double()
Wd = [p.double() for p in P]
def tap(v, off):
    out = torch.zeros_like(v)
    if off == 0: return v
    out[:, :, -off:] = v[:, :, :L + off]
    return out
offs = _lib.tap_offsets(k, d, True)
a = sum(torch.einsum("oc,bct->bot", Wd[0][:, :, j], tap(xd, offs[j])) for j in range(k)) + Wd[1].view(1, -1, 1)
g = sum(torch.einsum("oc,bct->bot", Wd[2][:, :, j], tap(xd, offs[j])) for j in range(k)) + Wd[3].view(1, -1, 1)
ta0, sg0 = torch.tanh(a), torch.sigmoid(g)
z0 = ta0 * sg0
r0 = torch.einsum("oc,bct->bot", Wd[4], z0) + Wd[5].view(1, -1, 1) + torch.einsum("oc,bct->bot", Wd[8], xd) + Wd[9].view(1, -1, 1)
s0 = torch.einsum("oc,bct->bot", Wd[6], z0) + Wd[7].view(1, -1, 1)
def rel(got, ref): return float((got - ref).abs().max() / ref.abs().max())
print("sg", rel(readback(sg, 1), sg0), "z", rel(readback(z, 1), z0), "tanh recovered as z / sg", rel(readback(z, 1) / readback(sg, 1), ta0))
print("r ", rel(readback(r, rs), r0), "skip", rel(S.double().cpu(), s0), "flag", int(flag.item()))
# atanh-level check of the pre-activation: invert the gate where it is well conditioned
am = a.abs() < 1.0
print("a (via atanh of z / sg, |a|<1)", float(((torch.atanh((readback(z, 1) / readback(sg, 1)).clamp(-0.999999, 0.999999)) - a)[am]).abs().max()))
# ---- where is z wrong? ----
zt = z.t.float().view(B, mode.planes, FH._cp32(C) // 8, layout.ld, 8)
def plane(p):
    return zt[:, p, :, layout.halo:layout.halo + L, :].permute(0, 1, 3, 2).reshape(B, -1, L)[:, :C].double().cpu()
e_hi = (plane(0) - z0).abs()
print("z hi-only err", float(e_hi.max()), " hi+lo err", float((plane(0) + (plane(1) if mode.planes == 2 else 0) - z0).abs().max()))
if mode.planes == 2:
    lo_expected = (z0 - plane(0))
    bad = ((plane(1) - lo_expected).abs() > 1e-6)
    print("bad lo elements:", int(bad.sum()), "of", bad.numel())
    if int(bad.sum()):
        idx = bad.nonzero()
        print("  channels:", sorted(set(idx[:, 1].tolist()))[:40], "...")
        print("  times   :", sorted(set(idx[:, 2].tolist()))[:40], "...")
        i0 = idx[0]
        print("  sample: got lo", float(plane(1)[tuple(i0)]), "expected", float(lo_expected[tuple(i0)]), "z", float(z0[tuple(i0)]))
if mode.planes == 2:
    tot = plane(0) + plane(1)
    err = (tot - z0).abs()
    w = (err == err.max()).nonzero()[0]
    w = tuple(w.tolist())
    print("worst z element", w, "hi", float(plane(0)[w]), "lo", float(plane(1)[w]), "z0", float(z0[w]), "ta0", float(ta0[w]), "sg0", float(sg0[w]),
          "ta got", float(readback(ta, 1)[w]), "sg got", float(readback(sg, 1)[w]))
    big = (err > 1e-5).nonzero()
    print("elements with |err| > 1e-5:", big.shape[0])
    for i in big[:12]:
        i = tuple(i.tolist())
        print("   ", i, "hi %.8f lo %.3e z0 %.8f" % (float(plane(0)[i]), float(plane(1)[i]), float(z0[i])))
