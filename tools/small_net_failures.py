"""Corrector failures of the host-driven integrator on SMALL networks against the resident kernel and the CPU port: which switch
of the host-driven path (fused corrector update, speculation, blind iterations) is behind the excess seen in
tools/robustness_resident.py. Usage: python tools/small_net_failures.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

VARIANTS = (("resident", {"KIN_RESIDENT": "1"}), ("host", {"KIN_RESIDENT": "0"}), ("host, update not fused", {"KIN_RESIDENT": "0", "KIN_FUSE_NEWTON": "0"}),
            ("host, no speculation", {"KIN_RESIDENT": "0", "KIN_SPECULATE": "0"}), ("host, neither", {"KIN_RESIDENT": "0", "KIN_SPECULATE": "0", "KIN_FUSE_NEWTON": "0"}),
            ("host, no fast sync", {"KIN_RESIDENT": "0", "KIN_NO_FAST_SYNC": "1"}), ("host, no LU cache", {"KIN_RESIDENT": "0", "KIN_LU_BAND": "0"}),
            ("host, plain substitution", {"KIN_RESIDENT": "0", "KIN_LU_EXPLICIT": "0"}))
if len(sys.argv) == 1:          # the switches are read once per process: one child per variant
    for name, env in VARIANTS:
        subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, **env), check=False)
    sys.exit(0)
name = sys.argv[1]
for n, seed, T in ((200, 3, 1000.0), (300, 2, 1300.0), (100, 12345, 1000.0), (1000, 12345, 1000.0)):
    net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(T)
    u0 = np.zeros(n); u0[0] = 1.0
    for tol in (1.0, 1e-2):
        p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10 * tol, reltol=1e-8 * tol, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                           solve_chunkstep=1e-3, maxiters=10**6, save_interval=1e-3, dtmin=1e-30)
        t, u, rc, st, _ = h.solve(p, u0)
        print(json.dumps({"n": n, "seed": seed, "T": T, "tol": tol, "path": name, "rc": rc, "steps": st["n_steps"], "rejected": st["n_rejected"], "factor": st["n_factor"],
                          "newton_fail": st["n_newton_fail"], "linsolve": st["n_linsolve"], "jac": st["n_jac"]}), flush=True)
    h.close()
