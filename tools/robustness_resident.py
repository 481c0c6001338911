"""Robustness sweep of the RESIDENT integrator (resident.hip: one workgroup owns the trajectory) over network seeds, sizes,
temperatures and tolerances - static chunkwise solves and short ramps, each also through the host-driven integrator
(KIN_RESIDENT=0) and, as members of ONE kin_solve_ensemble launch, through the shared-CU build of the kernel. Every run must
end with Success and without a tolerance retry; the two integrators must agree within the step-sequence tolerance.
Usage: python tools/robustness_resident.py > profiles/r04_robustness_resident.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

bad = 0
worst = 0.0
n_runs = 0
os.environ["KIN_RESIDENT_MAX_N"] = "600"; os.environ["KIN_RESIDENT_MAX_DENSE"] = "512"
for n in (60, 100, 200, 300, 400, 550):
    for seed in (12345, 1, 2, 3):
        net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        u0 = np.zeros(n); u0[0] = 1.0
        Ts = (700.0, 1000.0, 1300.0, 1600.0)
        for (ATOL, RTOL) in ((1e-10, 1e-8), (1e-8, 1e-6), (1e-12, 1e-10)):
            p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=ATOL, reltol=RTOL, adaptive_tols=1, update_tols=0, solve_chunks=1,
                               ban_negatives=0, solve_chunkstep=1e-3, maxiters=200000, save_interval=1e-3, dtmin=1e-30)
            res = {}
            for T in Ts:
                h.rates_at(T)
                for name, env in (("resident", "1"), ("host", "0")):
                    os.environ["KIN_RESIDENT"] = env
                    t0 = time.perf_counter()
                    t, u, rc, st, status = h.solve(p, u0)
                    res[(T, name)] = (u, rc, st, time.perf_counter() - t0)
                ur, rcr, sr, wr = res[(T, "resident")]
                uh, rch, sh, wh = res[(T, "host")]
                e = float((np.abs(ur - uh) / (ATOL + RTOL * np.abs(uh))).max()) if rcr == 0 and rch == 0 else float("nan")
                m = ur @ net.mass.astype(float)
                rec = {"kind": "static", "n": n, "seed": seed, "T": T, "rtol": RTOL, "rc": [rcr, rch], "retries": [sr["n_retries"], sh["n_retries"]],
                       "steps": [sr["n_steps"], sh["n_steps"]], "factor": [sr["n_factor"], sh["n_factor"]], "fail": [sr["n_newton_fail"], sh["n_newton_fail"]],
                       "wall": [round(wr, 4), round(wh, 4)], "units_apart": e, "mass_drift": float(np.abs(m / m[0] - 1).max())}
                bad += (rcr != 0) or sr["n_retries"] > 0
                worst = max(worst, e if e == e else 0.0)
                n_runs += 1
                print(json.dumps(rec), flush=True)
            # the same four solves as members of one ensemble launch (forced into the two-workgroups-per-CU build)
            os.environ["KIN_RESIDENT_SHARED_CU"] = "1"
            te, ue, ns, rcs, sts = h.solve_ensemble(p, np.tile(u0, (4, 1)), T=np.array(Ts))
            os.environ.pop("KIN_RESIDENT_SHARED_CU")
            same = all(rcs[i] == res[(T, "resident")][1] and (rcs[i] != 0 or np.array_equal(ue[i], res[(T, "resident")][0])) for i, T in enumerate(Ts))
            print(json.dumps({"kind": "ensemble_of_the_four", "n": n, "seed": seed, "rtol": RTOL, "rcs": [int(x) for x in rcs], "bit_identical_to_solo": bool(same)}), flush=True)
            bad += not same
        # a short ramp: 600 -> 1100 K over 10 ms, rate update every 0.5 ms, 2.5 ms chunks
        tst = np.arange(21) * 5e-4
        p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                           ban_negatives=0, solve_chunkstep=2.5e-3, maxiters=200000, save_interval=2.5e-3, dtmin=1e-30)
        out = {}
        for name, env in (("resident", "1"), ("host", "0")):
            os.environ["KIN_RESIDENT"] = env
            out[name] = h.solve(p, u0, tstops=tst, T_stops=600.0 + 5e4 * tst)
        (tr, ur, rcr, sr, _), (th, uh, rch, sh, _) = out["resident"], out["host"]
        e = float((np.abs(ur - uh) / (1e-10 + 1e-8 * np.abs(uh))).max()) if rcr == 0 and rch == 0 else float("nan")
        print(json.dumps({"kind": "ramp", "n": n, "seed": seed, "rc": [rcr, rch], "retries": [sr["n_retries"], sh["n_retries"]], "steps": [sr["n_steps"], sh["n_steps"]],
                          "restarts": [sr["n_restarts"], sh["n_restarts"]], "units_apart": e}), flush=True)
        bad += (rcr != 0) or sr["n_retries"] > 0
        worst = max(worst, e if e == e else 0.0)
        n_runs += 1
        os.environ["KIN_RESIDENT"] = "1"
        h.close()
print(json.dumps({"summary": True, "resident_runs": n_runs, "runs_with_a_failure_retry_or_mismatch": int(bad), "largest_distance_resident_vs_host_driven_units": worst}))
