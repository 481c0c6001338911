"""Deviation of the C4 / C5 ramp prefixes from their committed truths, in tolerance units (what tests/test_gpu_configs.py bounds).
Usage: python tools/ramp_units.py [c4] [c5]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn


def units(u, ref):
    return np.abs(u - ref) / (1e-10 + 1e-8 * np.abs(ref))


for name in (sys.argv[1:] or ["c4", "c5"]):
    n, r, nch = (10000, 50000, 3) if name == "c4" else (50000, 250000, 2)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"truth_{name}.npz"))
    net, Ea, A = synthetic_crn(n, r)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                       ban_negatives=0, solve_chunkstep=1e-2, maxiters=100000, save_interval=5e-3, dtmin=1e-30)
    t, u, rc, st, _ = h.solve(p, u0, tstops=z["tstops"], T_stops=z["T_stops"])
    sel = np.searchsorted(t, z["t"])
    e = units(u[sel], z["u"])
    print(json.dumps({"config": name, "rc": rc, "max_units": float(e.max()),
                      "rms_units": float(np.sqrt((e ** 2).mean(axis=1)).max()), "steps": st["n_steps"], "factor": st["n_factor"],
                      "retries": st["n_retries"], "wall_s": st["wall_seconds"]}), flush=True)
    h.close()
