import sys, time, os, numpy as np
sys.path.insert(0,'/root/repo')
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
N,R=50000,250000
net,Ea,A=synthetic_crn(N,R)
def kp(t1):
    return capi.KinParams(tspan0=0.0,tspan1=t1,abstol=1e-10,reltol=1e-8,adaptive_tols=1,update_tols=0,solve_chunks=1,ban_negatives=0,solve_chunkstep=1e-3,maxiters=100000,save_interval=-1.0)
h=capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea,A,k_max=1e12); h.rates_at(1000.0)
u0=np.zeros(N); u0[0]=1.0
h.solve(kp(1e-3),u0)
t0=time.perf_counter(); t,u,rc,st,_=h.solve(kp(3e-3),u0); dt=time.perf_counter()-t0
print(os.environ.get("TAG"), "wall %.3f"%dt, "m",st["lu_dense_dim"],"rounds",st["lu_rounds"],"steps",st["n_steps"],"factors",st["n_factor"], "rc",rc, flush=True)
