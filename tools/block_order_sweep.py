import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn, from_lists
import run_configs as rc
N,R,B=10000,50000,4096
net,Ea,A=synthetic_crn(N,R)
F=R//2
order=np.concatenate([np.arange(0,R,2),np.arange(1,R,2)])   # forwards block, then reverses (duplicate_reverse order)
net2=net.subset(order)
h=capi.HipNetwork.from_flat(net2)
h.set_rates(np.ones(R))
dev=torch.device("cuda"); g=torch.Generator(device=dev); g.manual_seed(1)
u=torch.pow(10.0, torch.rand((B,N),dtype=torch.float64,device=dev,generator=g)*12-12)
k=torch.rand((B,R),dtype=torch.float64,device=dev,generator=g)+0.5
du=torch.empty_like(u)
st=torch.cuda.Stream(); torch.cuda.set_stream(st)
dt=rc.timed(lambda: h.rhs_batched_dev(B,u.data_ptr(),k.data_ptr(),du.data_ptr(),st.cuda_stream))
alg=20*R+B*(8*R+16*N)
print("block order: %.3f ms  %.0f GB/s  %.1f %%"%(dt*1e3, alg/dt/1e9, alg/dt/8e10))
