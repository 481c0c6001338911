import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
for n in (300, 1000):
    net, Ea, A = synthetic_crn(n, 5 * n)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    p = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
    for K in (256, 1024, 2048):
        U0 = np.zeros((K, n)); U0[:, 0] = 1.0
        for Tm, name in ((np.linspace(900.0, 1300.0, K), "900-1300K"), (np.full(K, 1000.0), "1000K")):
            h.solve_ensemble(p, U0, T=Tm)
            t0 = time.perf_counter(); _, u, ns, rcs, sts = h.solve_ensemble(p, U0, T=Tm); w = time.perf_counter() - t0
            print(json.dumps({"species": n, "K": K, "T": name, "wall_s": w, "solves_per_s": K / w, "ok": int((rcs == 0).sum()),
                              "steps_mean": float(np.mean([q["n_steps"] for q in sts])), "steps_max": int(max(q["n_steps"] for q in sts)), "slots": sts[0]["lu_slots"]}), flush=True)
    h.close()
