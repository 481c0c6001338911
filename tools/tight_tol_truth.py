"""at rtol = 1e-10 the resident and the host-driven integrator end up to 6 800 tolerance units apart
(profiles/r04_robustness_resident.jsonl) and the resident one takes half the steps. Which of them is off? Both against a truth from
an integrator outside the BDF family (SciPy Radau on the oracle's RHS and Jacobian, rtol 1e-12 / atol 1e-14), in units of the
TIGHT tolerances (1e-12 + 1e-10 |u|), at every saved time; and the CPU port at the same tolerances next to them.
Usage: python tools/tight_tol_truth.py [n seed T]..."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.integrate import solve_ivp
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
from oracle import cpu_bdf, oracle as orc

cases = [(200, 3, 1000.0), (300, 2, 1300.0), (200, 1, 1000.0)]
if len(sys.argv) > 3:
    cases = [(int(sys.argv[i]), int(sys.argv[i + 1]), float(sys.argv[i + 2])) for i in range(1, len(sys.argv) - 2, 3)]
os.environ["KIN_RESIDENT_MAX_N"] = "600"; os.environ["KIN_RESIDENT_MAX_DENSE"] = "512"
ATOL, RTOL = 1e-12, 1e-10
for n, seed, T in cases:
    net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, T, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    t_eval = np.arange(1, 11) * 1e-3
    t0 = time.time()
    sol = solve_ivp(lambda t, u: on.rhs(k, u), (0.0, 1e-2), u0, method="Radau", jac=lambda t, u: on.jac(k, u).toarray(), rtol=1e-12, atol=1e-14,
                    t_eval=t_eval, first_step=1e-22)
    assert sol.success
    truth = sol.y.T
    rec = {"n": n, "seed": seed, "T": T, "radau_s": round(time.time() - t0, 1), "radau_nfev": int(sol.nfev)}
    units = lambda u: float((np.abs(u - truth) / (ATOL + RTOL * np.abs(truth))).max())
    rmsu = lambda u: float(np.sqrt(((np.abs(u - truth) / (ATOL + RTOL * np.abs(truth))) ** 2).mean(axis=1)).max())
    h = capi.HipNetwork.from_flat(net); h.set_rates(k)
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=ATOL, reltol=RTOL, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=200000, save_interval=1e-3, dtmin=1e-30)
    for name, env in (("resident", "1"), ("host_driven", "0")):
        os.environ["KIN_RESIDENT"] = env
        t, u, rc, st, _ = h.solve(p, u0)
        rec[name] = {"rc": rc, "steps": st["n_steps"], "fail": st["n_newton_fail"], "factor": st["n_factor"], "max_units": round(units(u[1:]), 1), "rms_units": round(rmsu(u[1:]), 2)}
    tc, uc, rcc, stc = cpu_bdf.CpuSolver(net).solve(dict(tspan=(0.0, 1e-2), solve_chunks=True, solve_chunkstep=1e-3, save_interval=1e-3, abstol=ATOL, reltol=RTOL,
                                                         dtmin=1e-30, maxiters=200000), u0, k0=k)
    rec["cpu_port"] = {"rc": rcc, "steps": stc["n_steps"], "fail": stc["n_newton_fail"], "factor": stc["n_factor"], "max_units": round(units(uc[1:]), 1), "rms_units": round(rmsu(uc[1:]), 2)}
    # and at the default tolerances, in default units, for scale
    pd = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                        solve_chunkstep=1e-3, maxiters=200000, save_interval=1e-3, dtmin=1e-30)
    for name, env in (("resident_default_tol", "1"), ("host_driven_default_tol", "0")):
        os.environ["KIN_RESIDENT"] = env
        t, u, rc, st, _ = h.solve(pd, u0)
        rec[name] = {"rc": rc, "steps": st["n_steps"], "max_default_units": round(float((np.abs(u[1:] - truth) / (1e-10 + 1e-8 * np.abs(truth))).max()), 1)}
    h.close()
    print(json.dumps(rec), flush=True)
