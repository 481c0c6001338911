"""Robustness sweep of kin_solve over network seeds, sizes, temperatures and tolerance settings (static chunkwise solves and
short ramps; BAN_NEG=1 in the environment runs them with ban_negatives): every run must end with retcode Success and without a tolerance retry; prints steps / factorisations / wall
and the mass-invariant drift. Usage: python tools/robustness_sweep.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

bad = 0
SOLVE_CHUNKS = int(os.environ.get("SOLVE_CHUNKS", "1"))   # 2: the warm-chunk-start extension (kin_params.solve_chunks = 2)
BAN = int(os.environ.get("BAN_NEG", "0"))      # ODESimulationParams.ban_negatives (isoutofdomain, methods.jl:169-171)
WIDE = len(sys.argv) > 1 and sys.argv[1] in ("wide", "wide2")      # more seeds and temperatures on the two smaller sizes
WIDE2 = len(sys.argv) > 1 and sys.argv[1] == "wide2"               # other seeds, in-between temperatures, looser / tighter tolerances
TOLS = ((1e-10, 1e-8), (1e-8, 1e-6), (1e-12, 1e-10)) if WIDE2 else ((1e-10, 1e-8),)
BIG = len(sys.argv) > 1 and sys.argv[1] == "big"                   # 10k species only, eight other seeds, five temperatures
for (n, r) in (((10000, 50000),) if BIG else ((1000, 5000), (3000, 15000)) if WIDE else ((1000, 5000), (3000, 15000), (10000, 50000))):
    for seed in ((21, 22, 23, 24, 25, 26, 27, 28) if BIG else (11, 12, 13, 14, 15, 16, 17, 18) if WIDE2 else (12345, 1, 2, 3, 4, 5, 6, 7, 8, 9) if WIDE else (12345, 1, 2, 3)):
        net, Ea, A = synthetic_crn(n, r, seed=seed)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        u0 = np.zeros(n); u0[0] = 1.0
        for T, (ATOL, RTOL) in [(T, tl) for T in ((700.0, 900.0, 1100.0, 1300.0, 1500.0) if BIG else (900.0, 1100.0, 1300.0, 1500.0) if WIDE2 else (600.0, 800.0, 1000.0, 1200.0, 1400.0, 1800.0) if WIDE else (800.0, 1000.0, 1400.0)) for tl in TOLS]:
            h.rates_at(T)
            p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=ATOL, reltol=RTOL, adaptive_tols=1, update_tols=0, solve_chunks=SOLVE_CHUNKS,
                               ban_negatives=BAN, solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3, dtmin=1e-30)
            t0 = time.perf_counter()
            t, u, rc, st, status = h.solve(p, u0)
            m = u @ net.mass.astype(float)
            rec = {"kind": "static", "n": n, "seed": seed, "T": T, "rtol": RTOL, "rc": rc, "retries": st["n_retries"], "steps": st["n_steps"],
                   "factor": st["n_factor"], "fail": st["n_newton_fail"], "wall": round(time.perf_counter() - t0, 3),
                   "mass_drift": float(np.abs(m / m[0] - 1).max()), "umin": float(u.min())}
            bad += (rc != 0) or st["n_retries"] > 0
            print(json.dumps(rec), flush=True)
        # a short ramp: 600 -> 1100 K over 10 ms, rate update every 0.5 ms, 2.5 ms chunks
        tst = np.arange(21) * 5e-4
        p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=SOLVE_CHUNKS,
                           ban_negatives=BAN, solve_chunkstep=2.5e-3, maxiters=100000, save_interval=2.5e-3, dtmin=1e-30)
        t0 = time.perf_counter()
        t, u, rc, st, status = h.solve(p, u0, tstops=tst, T_stops=600.0 + 5e4 * tst)
        m = u @ net.mass.astype(float)
        rec = {"kind": "ramp", "n": n, "seed": seed, "rc": rc, "retries": st["n_retries"], "steps": st["n_steps"], "factor": st["n_factor"],
               "fail": st["n_newton_fail"], "wall": round(time.perf_counter() - t0, 3), "mass_drift": float(np.abs(m / m[0] - 1).max())}
        bad += (rc != 0) or st["n_retries"] > 0
        print(json.dumps(rec), flush=True)
        h.close()
print("runs with a failure or a retry:", bad)
