import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from wavenet_speech_amd import _lib, functional as HF, functional_half as FH
lib=_lib.load(); dev=torch.device("cuda:0")
prec=sys.argv[1]; C=int(sys.argv[2]); L=256; B=2
mode=FH._Mode(prec); layout=FH.HalfLayout(L,0)
def dense(lease, C):
    t=lease.t
    G=t.shape[1]//mode.planes
    v=t.view(B, mode.planes, G, layout.ld, 8).float().sum(1)
    return v[:,:,layout.halo:layout.halo+L,:].permute(0,1,3,2).reshape(B,G*8,L)[:,:C]
torch.manual_seed(0)
w=torch.randn(C,C,1,device=dev)*0.09; b=torch.zeros(C,device=dev)
sh=_lib.ConvShape(B,L,C,C,1,1,1,layout.ld,layout.halo)
pk=torch.empty(lib.wn_hconv_packed_bytes(ctypes.byref(sh),mode.code),dtype=torch.uint8,device=dev)
rs=float(lib.wn_hseries_residual_scale())
_lib.check(lib.wn_hconv_pack(ctypes.byref(sh),mode.code,HF._p(w),HF._p(b),ctypes.c_float(rs),HF._p(pk),HF._stream()),"pack")
dy=torch.randn(B,C,L,device=dev)*0.1; act=torch.randn(B,C,L,device=dev)
act=torch.where(act>0,act,act*0.01)
DY=FH._hlease(mode,B,C,layout,dev); FH._load(lib,mode,dy,DY,layout,1.0,None,None)
ACT=FH._hlease(mode,B,C,layout,dev); FH._load(lib,mode,act,ACT,layout,rs,None,None)
DX=FH._hlease(mode,B,C,layout,dev)
_lib.check(lib.wn_hconv_backward_data_series(ctypes.byref(sh),mode.code,HF._p(pk),HF._p(DY),HF._p(ACT),ctypes.c_float(0.01),HF._p(DX),None,HF._stream()),"bwd")
got=dense(DX,C)
dyq=dense(DY,C); wq=w[:,:,0]
ref=torch.einsum('oc,bot->bct', wq, dyq)*torch.where(act>0,torch.ones_like(act),torch.full_like(act,0.01))
print(prec,C,"rel err", float((got-ref).abs().max()/ref.abs().max()), "max",float(ref.abs().max()))
bad=((got-ref).abs()>0.02*ref.abs().max()).nonzero()
print("bad elements",bad.shape[0], bad[:10].tolist())
