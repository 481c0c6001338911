"""BASELINE configuration C3 as SURVEY 8(d) specifies it: 10k species / 50k reactions, static 1000 K, tspan (0, 1) s,
solve_chunks = true with the default 1 ms chunks (1 000 chunks) AND solve_chunks = false (one integration over the whole
span, methods.jl:132-183); abstol 1e-10, reltol 1e-8. One JSON record per variant: wall-clock, retcode, step counts, the
mass invariant over the whole run, and how far the two variants' final states are apart in tolerance units.
Usage: python tools/c3_full.py [t_end=1.0]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

t_end = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
N, R = 10000, 50000
net, Ea, A = synthetic_crn(N, R)
u0 = np.zeros(N); u0[0] = 1.0
finals = {}
for name, chunks, dtmin in (("chunkwise_1ms", 1, 0.0), ("complete_timespan", 0, 1e-30)):
    t0 = time.perf_counter()
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(1000.0)
    p = capi.KinParams(tspan0=0.0, tspan1=t_end, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=chunks,
                       ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3 if not chunks else -1.0, dtmin=dtmin)
    t, u, rc, st, status = h.solve(p, u0)
    cold = time.perf_counter() - t0
    t1 = time.perf_counter()
    t, u, rc, st, status = h.solve(p, u0)
    warm = time.perf_counter() - t1
    m = h.solution_dot(net.mass.astype(float))
    finals[name] = u[-1].copy()
    print(json.dumps({"config": "C3", "variant": name, "tspan": [0.0, t_end], "dtmin": dtmin or "reference (eps)",
                      "cold_wall_s_incl_create": cold, "warm_wall_s": warm, "retcode": capi.RETCODE_NAMES[rc], "status": status,
                      "n_saved": len(t), "mass_invariant_max_rel_drift": float(np.max(np.abs(m / m[0] - 1.0))),
                      "min_concentration": float(u.min()), "stats": st}), flush=True)
    h.close()
a, b = finals["chunkwise_1ms"], finals["complete_timespan"]
e = np.abs(a - b) / (1e-10 + 1e-8 * np.abs(b))
print(json.dumps({"config": "C3", "final_state_chunkwise_vs_complete_in_tolerance_units": {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean()))}}))
