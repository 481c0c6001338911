import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
N, R, nch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net, Ea, A = synthetic_crn(N, R)
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                   ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
res = {}
for mode in ("0", "1"):
    os.environ["KIN_SPECULATE"] = mode
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(1000.0)
    h.solve(p, u0)
    ws = []
    for _ in range(3):
        t0 = time.perf_counter(); t, u, rc, st, status = h.solve(p, u0); ws.append(time.perf_counter() - t0)
    res[mode] = (t, u, rc, st, min(ws))
    h.close()
a, b = res["0"], res["1"]
print(json.dumps({"N": N, "chunks": nch, "wall_off": a[4], "wall_on": b[4], "rc": [a[2], b[2]], "identical_t": bool(np.array_equal(a[0], b[0])),
                  "identical_u": bool(np.array_equal(a[1], b[1])), "steps": [a[3]["n_steps"], b[3]["n_steps"]],
                  "linsolve": [a[3]["n_linsolve"], b[3]["n_linsolve"]], "factor": [a[3]["n_factor"], b[3]["n_factor"]]}))
