import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
N, R = int(sys.argv[1]), int(sys.argv[2]); t_end = float(sys.argv[3]); T0 = float(sys.argv[4])
net, Ea, A = synthetic_crn(N, R)
h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
tst = np.arange(int(round(t_end / 1e-3)) + 1) * 1e-3
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=t_end, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0, solve_chunkstep=1e-2, maxiters=int(sys.argv[5]), save_interval=5e-3)
t0 = time.time()
t, u, rc, st, status = h.solve(p, u0, tstops=tst, T_stops=T0 + 50.0 * tst)
print("rc", rc, "status", status, "wall", time.time() - t0, st, flush=True)
