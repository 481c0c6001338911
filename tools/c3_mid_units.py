import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
WHICH = sys.argv[1] if len(sys.argv) > 1 else "mid"
z = np.load(f"tests/golden/truth_c3_{WHICH}.npz")
TEND = float(z["t"][-1])
net, Ea, A = synthetic_crn(10000, 50000)
h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(1000.0)
u0 = np.zeros(10000); u0[0] = 1.0
def units(a, b): return np.abs(a - b) / (1e-10 + 1e-8 * np.abs(b))
for name, kw in (("chunkwise", dict(solve_chunks=1, save_interval=-1.0, dtmin=0.0)), ("complete", dict(solve_chunks=0, save_interval=float(z["t"][1]), dtmin=1e-30)),
                 ("chunkwise_warm", dict(solve_chunks=2, save_interval=-1.0, dtmin=0.0)),
                 ("chunkwise_x0.1", dict(solve_chunks=1, save_interval=-1.0, dtmin=1e-30, tol=0.1))):
    tol = kw.pop("tol", 1.0)
    p = capi.KinParams(tspan0=0.0, tspan1=TEND, abstol=1e-10 * tol, reltol=1e-8 * tol, adaptive_tols=1, update_tols=0, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=1000000, **kw)
    t, u, rc, st, status = h.solve(p, u0)
    sel = [int(np.argmin(np.abs(t - tt))) for tt in z["t"]]
    e = units(u[sel], z["u"])
    print(json.dumps({"case": name, "rc": rc, "n_saved": len(t), "t_err": float(np.abs(t[sel] - z["t"]).max()), "max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean(axis=1)).max()),
                      "p99.9": float(np.percentile(e, 99.9)), "per_save_max": [round(float(x), 1) for x in e.max(axis=1)], "steps": st["n_steps"], "nf": st["n_newton_fail"],
                      "worst_species": [int(i) for i in np.argsort(e.max(axis=0))[-3:]]}), flush=True)
