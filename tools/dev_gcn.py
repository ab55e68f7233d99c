import sys, ctypes, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pose2room_amd import _lib
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_tables
dev=torch.device('cuda:0')
A=Graph().A
K,V=A.shape[0],A.shape[1]
nbr,gidx,Lk=gcn_tables.build(A)
print('Lk',Lk)
torch.manual_seed(0)
def run(N,T,check=True,reps=0):
    x=torch.randn(N,64,T,V,device=dev)
    W=torch.randn(K*64,64,device=dev)/8
    b=torch.randn(K*64,device=dev)*0.1
    imp=1+0.1*torch.randn(K,V,V,device=dev)
    Aeff=torch.tensor(A,dtype=torch.float32,device=dev)*imp
    coef=gcn_tables.coefficients(Aeff,gidx.to(dev)).contiguous()
    colsum=Aeff.sum(1)  # (K,V): sum over v
    bias_cv=(b.view(K,64).t() @ colsum).contiguous()  # (64,V)
    z=torch.empty_like(x)
    LkA=(ctypes.c_int*K)(*Lk)
    nb=nbr.to(dev)
    def call():
        st=_lib.lib().p2r_stgcn_gcn_forward(N,T,V,K,LkA,_lib.ptr(x),_lib.ptr(W),_lib.ptr(nb),_lib.ptr(coef),_lib.ptr(bias_cv),_lib.ptr(z),_lib.current_stream(dev))
        assert st==0, st
    call(); torch.cuda.synchronize()
    if check:
        y=torch.nn.functional.conv2d(x.double(),W.double().view(K*64,64,1,1),b.double())
        ref=torch.einsum('nkctv,kvw->nctw',y.view(N,K,64,T,V),Aeff.double())
        err=(z.double()-ref).abs().max().item(); print(N,T,'maxerr',err,'refmax',ref.abs().max().item())
    if reps:
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); e1.synchronize()
        ms=e0.elapsed_time(e1)/reps
        fl=2*64*K*64*N*T*V
        print(N,T,'ms',ms,'dense TFLOP/s',fl/ms/1e9, 'ref-algorithmic TFLOP/s', (fl+2*64*V*V*K*N*T)/ms/1e9)
run(1,7); run(2,20); run(1,1); run(3,33)
run(32,1024,check=False,reps=5)
