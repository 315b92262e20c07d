import os, sys, torch
sys.path.insert(0, os.getcwd())
import wavenet_speech_amd as W
from wavenet_speech_amd import functional_half as FH
from tests.test_gpu_half import _cond_wavenet
DEV="cuda:0"
c, dims, L, B = 256, (1,4), 256, 2
layers = [(c, c, 2, d) for d in dims]
net = _cond_wavenet(c, layers, seed=c + L).to(DEV)
g = torch.Generator().manual_seed(L + B)
x, cot = torch.randn(B, c, L, generator=g).to(DEV), torch.randn(B, c, L, generator=g).to(DEV)
prec=sys.argv[1]
W.set_precision(net, prec)
mode=FH._Mode(prec)
made=[]
orig=FH._hlease
def rec(*a,**k):
    l=orig(*a,**k); made.append(l); return l
FH._hlease=rec
def dense(lease, C, layout):
    t=lease.t
    G=t.shape[1]//mode.planes
    v=t.view(B, mode.planes, G, layout.ld, 8).float().sum(1)
    return v[:,:,layout.halo:layout.halo+L,:].permute(0,1,3,2).reshape(B,G*8,L)[:,:C]
xg=x.clone().requires_grad_(True)
y=net(xg)
nf=len(made)
h0,h1=made[nf-2],made[nf-1]
lay=h0.layout
H0=dense(h0,c,lay).clone(); H1=dense(h1,c,lay).clone()
(y*cot).sum().backward()
dY,dh1,dS=made[nf],made[nf+1],made[nf+2]
DY=dense(dY,c,lay); DH1=dense(dh1,c,lay); DS=dense(dS,c,lay)
w1=net.output_stack[1].weight[:,:,0]; w2=net.output_stack[3].weight[:,:,0]
# reference chain in fp32 from the stored tensors
ref_dh1=torch.einsum('oc,bot->bct', w2, DY)*torch.where(H1>0,torch.ones_like(H1),torch.full_like(H1,0.01))
print("dh1 rel err vs recomputation", float((DH1-ref_dh1).abs().max()/ref_dh1.abs().max()))
ref_ds=torch.einsum('oc,bot->bct', w1, DH1)*torch.where(H0>0,torch.ones_like(H0),torch.full_like(H0,0.01))
print("dS rel err", float((DS-ref_ds).abs().max()/ref_ds.abs().max()))
# check h0 against leaky(S)/16 computed by the unfused path
os.environ["WN_SERIES_HEAD"]="0"
from oracle import wavenet_oracle as O
sl,rm=O.capture_leaky_slopes(net)
FH._hlease=orig
with torch.no_grad(): net(x)
rm()
m0=sl["output_stack.0"].to(DEV); m1=sl["output_stack.2"].to(DEV)
print("mask0 mismatches", int(((H0>0)!=(m0>0.5)).sum()), "mask1 mismatches", int(((H1>0)!=(m1>0.5)).sum()), "of", H0.numel())
