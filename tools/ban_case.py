"""One ramp solve with ban_negatives (the case of tools/robustness_sweep.py that needs retries): python tools/ban_case.py N R SEED"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
n, r, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net, Ea, A = synthetic_crn(n, r, seed=seed)
h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
u0 = np.zeros(n); u0[0] = 1.0
tst = np.arange(21) * 5e-4
p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                   ban_negatives=int(os.environ.get("BAN_NEG", "1")), solve_chunkstep=2.5e-3, maxiters=100000, save_interval=2.5e-3, dtmin=1e-30)
t0 = time.perf_counter()
t, u, rc, st, status = h.solve(p, u0, tstops=tst, T_stops=600.0 + 5e4 * tst)
print(json.dumps({"n": n, "seed": seed, "rc": rc, "retries": st["n_retries"], "steps": st["n_steps"], "factor": st["n_factor"],
                  "fail": st["n_newton_fail"], "rejected": st["n_rejected"], "wall": round(time.perf_counter() - t0, 2), "umin": float(u.min())}))
