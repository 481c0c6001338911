"""Deviation of the device solves of the BASELINE configurations from the committed tight-tolerance truths
(tests/golden/truth_c3.npz, truth_c4.npz, truth_c5.npz) and from the compiled CPU baseline at default tolerances, in
tolerance units |u - ref| / (abstol + reltol |ref|). Usage: python tools/config_parity.py [c3 c4 c5]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402
from oracle import cpu_bdf  # noqa: E402
from oracle import oracle as orc  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def units(u, ref):
    return np.abs(u - ref) / (1e-10 + 1e-8 * np.abs(ref))


def report(name, u, ref):
    d = units(u, ref)
    return {f"{name}_max": float(d.max()), f"{name}_rms": float(np.sqrt((d ** 2).mean(axis=1)).max()),
            f"{name}_p999": float(np.percentile(d, 99.9))}


def kp(t1, chunk, save=None, dtmin=0.0):
    return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                          ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=-1.0 if save is None else save,
                          dtmin=dtmin)


def main():
    which = sys.argv[1:] or ["c3", "c4", "c5"]
    if "c3" in which:
        z = np.load(os.path.join(G, "truth_c3.npz"))
        net, Ea, A = synthetic_crn(10000, 50000)
        k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
        u0 = np.zeros(10000); u0[0] = 1.0
        h = capi.HipNetwork.from_flat(net)
        h.set_rates(k)
        t, u, rc, st, _ = h.solve(kp(2e-3, 1e-3), u0)
        rec = {"config": "c3", "rc": rc, "steps": st["n_steps"], "factor": st["n_factor"], **report("vs_truth", u, z["u"])}
        t0 = time.time()
        tc, uc, rcc, stc = cpu_bdf.CpuSolver(net).solve(dict(tspan=(0.0, 2e-3)), u0, k0=k)
        rec.update(cpu_s=time.time() - t0, cpu_steps=stc["n_steps"], **report("vs_cpu", u, uc), **report("cpu_vs_truth", uc, z["u"]))
        print(json.dumps(rec), flush=True)
        h.close()
    for name, (n, r, nch) in (("c4", (10000, 50000, 3)), ("c5", (50000, 250000, 2))):
        if name not in which:
            continue
        z = np.load(os.path.join(G, f"truth_{name}.npz"))
        net, Ea, A = synthetic_crn(n, r)
        u0 = np.zeros(n); u0[0] = 1.0
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        tst, T = z["tstops"], z["T_stops"]
        t0 = time.time()
        t, u, rc, st, _ = h.solve(kp(1e-2 * nch, 1e-2, 5e-3, dtmin=1e-30), u0, tstops=tst, T_stops=T)
        wall = time.time() - t0
        sel = np.searchsorted(t, z["t"])
        rec = {"config": name, "rc": rc, "steps": st["n_steps"], "factor": st["n_factor"], "wall_s": wall, "dense": st["lu_dense_dim"],
               **report("vs_truth", u[sel], z["u"])}
        print(json.dumps(rec), flush=True)
        h.close()


if __name__ == "__main__":
    main()
