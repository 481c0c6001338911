"""Resident integrator (one workgroup owns the trajectory, resident.hip) against the host-driven multi-kernel integrator
(solver.cpp) on the same static 20-chunk solves, by network size - the measurement behind KIN_RESIDENT_MAX_N's default.
Usage: python tools/resident_crossover.py [sizes...] > profiles/r04_resident_vs_host.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

sizes = [int(a) for a in sys.argv[1:]] or [100, 200, 300, 400, 500, 700, 1000]
os.environ["KIN_RESIDENT_MAX_N"] = "100000"; os.environ["KIN_RESIDENT_MAX_DENSE"] = "100000"   # eligibility by size off: the kernel's own limits decide
p = capi.KinParams(tspan0=0.0, tspan1=2e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                   solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
for N in sizes:
    net, Ea, A = synthetic_crn(N, 5 * N)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(1000.0)
    u0 = np.zeros(N); u0[0] = 1.0
    rec = {"species": N, "reactions": 5 * N}
    for name, env in (("resident", "1"), ("host_driven", "0")):
        os.environ["KIN_RESIDENT"] = env
        h.solve(p, u0)
        walls = []
        for _ in range(3):
            t0 = time.perf_counter()
            t, u, rc, st, status = h.solve(p, u0)
            walls.append(time.perf_counter() - t0)
        rec[name + "_s"] = min(walls)
        rec[name + "_steps"] = st["n_steps"]
        rec[name + "_factor"] = st["n_factor"]
        rec[name + "_rc"] = rc
        rec["dense_block"] = st["lu_dense_dim"]
        if name == "resident":
            ur = u
        else:
            rec["units_apart_max"] = float((np.abs(ur - u) / (1e-10 + 1e-8 * np.abs(u))).max())
    rec["host_over_resident"] = rec["host_driven_s"] / rec["resident_s"]
    print(json.dumps(rec), flush=True)
    h.close()
