"""at rtol = 1e-10 the resident and the host-driven integrator end up to 6 800 tolerance units apart
(profiles/r04_robustness_resident.jsonl) and the resident one takes half the steps. Which of them is off? Both - and the CPU port -
against tests/golden/truth_tight_200.npz (SciPy Radau at 10x tighter tolerances, tests/golden/make_truth_tight.py), in units of the
TIGHT tolerances (1e-12 + 1e-10 |u|) and, for scale, at the default tolerances in default units.
Usage: python tools/tight_tol_truth.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
from oracle import cpu_bdf, oracle as orc

z = np.load(os.path.join(ROOT, "tests", "golden", "truth_tight_200.npz"))
n, seed, T = int(z["n"]), int(z["seed"]), float(z["T"])
truth = z["u"]
os.environ["KIN_RESIDENT_MAX_N"] = "600"; os.environ["KIN_RESIDENT_MAX_DENSE"] = "512"
net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
k = orc.arrhenius(Ea, A, T, k_max=1e12)
u0 = np.zeros(n); u0[0] = 1.0
rec = {"n": n, "seed": seed, "T": T, "truth_rtol": float(z["rtol"]), "truth_self_check": float(z["self_check"])}
h = capi.HipNetwork.from_flat(net); h.set_rates(k)
for tag, (ATOL, RTOL) in (("tight", (1e-12, 1e-10)), ("default", (1e-10, 1e-8))):
    units = lambda u: np.abs(u - truth) / (ATOL + RTOL * np.abs(truth))
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=ATOL, reltol=RTOL, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=400000, save_interval=1e-3, dtmin=1e-30)
    for name, env in (("resident", "1"), ("host_driven", "0")):
        os.environ["KIN_RESIDENT"] = env
        t, u, rc, st, _ = h.solve(p, u0)
        e = units(u[1:])
        rec[f"{name}_{tag}"] = {"rc": rc, "steps": st["n_steps"], "fail": st["n_newton_fail"], "factor": st["n_factor"], "max_units": round(float(e.max()), 1),
                               "rms_units": round(float(np.sqrt((e ** 2).mean(axis=1)).max()), 2), "p999": round(float(np.percentile(e, 99.9)), 1)}
        print(json.dumps({f"{name}_{tag}": rec[f"{name}_{tag}"]}), flush=True)
    tc, uc, rcc, stc = cpu_bdf.CpuSolver(net).solve(dict(tspan=(0.0, 1e-2), solve_chunks=True, solve_chunkstep=1e-3, save_interval=1e-3, abstol=ATOL, reltol=RTOL,
                                                         dtmin=1e-30, maxiters=400000), u0, k0=k)
    e = units(uc[1:])
    rec[f"cpu_port_{tag}"] = {"rc": rcc, "steps": stc["n_steps"], "fail": stc["n_newton_fail"], "factor": stc["n_factor"], "max_units": round(float(e.max()), 1),
                              "rms_units": round(float(np.sqrt((e ** 2).mean(axis=1)).max()), 2)}
    print(json.dumps({f"cpu_port_{tag}": rec[f"cpu_port_{tag}"]}), flush=True)
h.close()
print(json.dumps(rec), flush=True)
