import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
for n, seed, T in ((200, 3, 1000.0), (300, 2, 1300.0), (100, 12345, 1000.0)):
    net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=10**6, save_interval=1e-3, dtmin=1e-30)
    t, u, ns, rcs, sts = h.solve_ensemble(p, u0[None, :], T=np.array([T]))
    st = sts[0]
    print(json.dumps({"n": n, "path": os.environ.get("KIN_ENSEMBLE_BATCHED", "0") == "1" and "lockstep (resident_core controller, multi-kernel linear algebra)" or "resident ensemble",
                      "rc": int(rcs[0]), "steps": st["n_steps"], "rejected": st["n_rejected"], "factor": st["n_factor"], "newton_fail": st["n_newton_fail"]}), flush=True)
