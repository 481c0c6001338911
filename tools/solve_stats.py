"""Prints kin_solve statistics of the C3-style static solve (N species, R reactions, n chunks of 1 ms at 1000 K): steps,
factorisations, cache hits, wall. Env knobs of the solver (KIN_LU_CACHE_SLOTS, KIN_LU_BAND) apply."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

N, R, nch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net, Ea, A = synthetic_crn(N, R)
h = capi.HipNetwork.from_flat(net)
h.set_arrhenius(Ea, A, k_max=1e12)
h.rates_at(1000.0)
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                   ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
h.solve(p, u0)
walls = []
for _ in range(int(__import__("os").environ.get("SOLVE_REPEATS", "3"))):
    t0 = time.perf_counter()
    t, u, rc, st, status = h.solve(p, u0)
    walls.append(time.perf_counter() - t0)
wall = min(walls)
print(json.dumps({"N": N, "R": R, "chunks": nch, "rc": rc, "wall_s": wall, "walls": [round(w, 4) for w in walls], **{q: st[q] for q in
      ("n_steps", "n_rejected", "n_factor", "n_linsolve", "n_newton_fail", "n_jac", "n_restarts", "n_lu_reused", "n_lu_dropped", "lu_slots", "lu_dense_dim")}}))
if len(sys.argv) > 4:
    np.save(sys.argv[4], u)
