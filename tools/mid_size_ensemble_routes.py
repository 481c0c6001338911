import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
for N in (600, 1000):
    net, Ea, A = synthetic_crn(N, 5 * N)
    u0 = np.zeros(N); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    for K in (2, 4, 8, 12, 24):
        U0 = np.tile(u0, (K, 1)); T = np.linspace(950.0, 1150.0, K)
        h.solve_ensemble(p, U0, T=T)
        walls = []
        for _ in range(3):
            t0 = time.perf_counter(); h.solve_ensemble(p, U0, T=T); walls.append(time.perf_counter() - t0)
        print(json.dumps({"N": N, "route": os.environ.get("ROUTE"), "K": K, "wall": round(min(walls), 4), "solves_per_s": round(K / min(walls), 1)}), flush=True)
    h.close()
