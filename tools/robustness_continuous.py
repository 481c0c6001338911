"""Robustness sweep of the continuous-rate path (kin_solve_continuous: k(T(t)) evaluated on the device at every step attempt)
over network seeds and ramp rates; every run must end with retcode Success. Usage: python tools/robustness_continuous.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

bad = 0
for (n, r) in ((1000, 5000), (3000, 15000)):
    for seed in (1, 2, 3, 4, 5, 6):
        net, Ea, A = synthetic_crn(n, r, seed=seed)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        u0 = np.zeros(n); u0[0] = 1.0
        for (T0, rate) in ((600.0, 2e4), (900.0, 5e4), (1200.0, -3e4)):
            for chunks in (1, 0):
                tn = np.linspace(0.0, 1e-2, 21)
                Tn = T0 + rate * tn
                p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=chunks,
                                   ban_negatives=0, solve_chunkstep=2.5e-3, maxiters=200000, save_interval=2.5e-3, dtmin=1e-30)
                t0 = time.perf_counter()
                t, u, rc, st, status = h.solve_continuous(p, u0, tn, Tn)
                m = u @ net.mass.astype(float)
                rec = {"n": n, "seed": seed, "T0": T0, "rate": rate, "chunks": chunks, "rc": rc, "retries": st["n_retries"], "steps": st["n_steps"],
                       "factor": st["n_factor"], "fail": st["n_newton_fail"], "wall": round(time.perf_counter() - t0, 3),
                       "mass_drift": float(np.abs(m / m[0] - 1).max())}
                bad += (rc != 0)
                print(json.dumps(rec), flush=True)
        h.close()
print("runs that did not end with Success:", bad)
