"""One static chunkwise solve of a synthetic network: python tools/solve_case.py N R SEED T [cpu]  (10 chunks of 1 ms, tolerances
from the environment (ATOL, RTOL; defaults 1e-10 / 1e-8), dtmin 1e-30) - prints retcode, retries, steps, factorisations, corrector failures, rejections. With `cpu` the
compiled CPU baseline runs instead of the device."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

n, r, seed, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
ATOL, RTOL = float(os.environ.get("ATOL", "1e-10")), float(os.environ.get("RTOL", "1e-8"))
net, Ea, A = synthetic_crn(n, r, seed=seed)
u0 = np.zeros(n); u0[0] = 1.0
if len(sys.argv) > 5 and sys.argv[5] == "cpu":
    from oracle import cpu_bdf, oracle as orc
    k = orc.arrhenius(Ea, A, T, k_max=1e12)
    t0 = time.time()
    t, u, rc, st = cpu_bdf.CpuSolver(net).solve(dict(tspan=(0.0, 1e-2), solve_chunks=True, solve_chunkstep=1e-3, save_interval=1e-3, dtmin=1e-30, abstol=ATOL, reltol=RTOL), u0, k0=k)
    print("cpu", rc, {q: st[q] for q in ("n_steps", "n_factor", "n_newton_fail", "n_retries") if q in st}, round(time.time() - t0, 1))
else:
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(T)
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=ATOL, reltol=RTOL, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=int(os.environ.get("BAN_NEG", "0")),
                       solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3, dtmin=1e-30)
    t, u, rc, st, status = h.solve(p, u0)
    print("gpu rc", rc, "retries", st["n_retries"], "steps", st["n_steps"], "factor", st["n_factor"], "fail", st["n_newton_fail"], "rejected", st["n_rejected"])
