"""The statistics tests/test_gpu_configs.py gates on (oracle.deviation_stats: rms, p99.9, the 50 major species; max reported),
for every configuration case of that file, one JSON line each. Usage: python tools/config_stats.py [c3 mid long c4 c4long c5]
LU_BAND=0.32,0.35,0.38 repeats every case under those reuse bands (a perturbation that leaves the algorithm alone)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def kp(t1, chunk, save=None, tol=1.0, dtmin=0.0, chunks=1):
    return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10 * tol, reltol=1e-8 * tol, adaptive_tols=1, update_tols=0, solve_chunks=chunks,
                          ban_negatives=0, solve_chunkstep=chunk, maxiters=1000000, save_interval=-1.0 if save is None else save, dtmin=dtmin)


def line(case, t, u, rc, st, z, wall):
    sel = [int(np.argmin(np.abs(t - tt))) for tt in z["t"]]
    assert np.abs(t[sel] - z["t"]).max() < 1e-12
    s = orc.deviation_stats(u[sel], z["u"])
    print(json.dumps({"case": case, "band": os.environ.get("KIN_LU_BAND", "default"), "rc": rc, "wall_s": round(wall, 4), "steps": st["n_steps"],
                      "factor": st["n_factor"], "nf": st["n_newton_fail"], **{k: round(v, 2) for k, v in s.items()}}), flush=True)


def static_cases(name, t_end, save_complete):
    z = np.load(os.path.join(G, f"truth_c3{name}.npz"))
    net, Ea, A = synthetic_crn(10000, 50000)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(1000.0)
    u0 = np.zeros(10000); u0[0] = 1.0
    cases = [("chunkwise", kp(t_end, 1e-3)), ("complete", kp(t_end, 1e-3, save=save_complete, dtmin=1e-30, chunks=0)),
             ("warm", kp(t_end, 1e-3, chunks=2)), ("x0.1", kp(t_end, 1e-3, tol=0.1, dtmin=1e-30))]
    h.solve(kp(2e-3, 1e-3), u0)
    for cname, p in cases:
        t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(p, u0); w = time.perf_counter() - t0
        line(f"c3{name}_{cname}", t, u, rc, st, z, w)
    h.close()


def ramp_case(name, n, r, n_chunks):
    z = np.load(os.path.join(G, f"truth_{name}.npz"))
    net, Ea, A = synthetic_crn(n, r)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    for cname, tol in (("default", 1.0), ("x0.1", 0.1)):
        t0 = time.perf_counter()
        t, u, rc, st, _ = h.solve(kp(1e-2 * n_chunks, 1e-2, 5e-3, tol=tol, dtmin=1e-30), u0, tstops=z["tstops"], T_stops=z["T_stops"])
        line(f"{name}_{cname}", t, u, rc, st, z, time.perf_counter() - t0)
    h.close()


def main():
    which = sys.argv[1:] or ["c3", "mid", "long", "c4", "c4long", "c5"]
    bands = os.environ.get("LU_BAND", "")
    for band in (bands.split(",") if bands else [None]):
        if band:
            os.environ["KIN_LU_BAND"] = band
        if "c3" in which: static_cases("", 2e-3, 1e-3)
        if "mid" in which: static_cases("_mid", 0.03, 5e-3)
        if "long" in which: static_cases("_long", 0.1, 1e-2)
        if "c4" in which: ramp_case("c4", 10000, 50000, 3)
        if "c4long" in which: ramp_case("c4_long", 10000, 50000, 20)
        if "c5" in which: ramp_case("c5", 50000, 250000, 2)


if __name__ == "__main__":
    main()
