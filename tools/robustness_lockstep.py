"""Robustness sweep of the LOCKSTEP ensemble (ensemble.cpp: kin_solve_ensemble beyond the resident kernel's size): networks of
1 500-10 000 species, several seeds, members at 700-1600 K, static 10-chunk solves and a shared ramp - every member against its solo
kin_solve (the host-driven integrator). Usage: python tools/robustness_lockstep.py > profiles/r04_robustness_lockstep.jsonl"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

os.environ["KIN_ENSEMBLE_BATCHED"] = "1"
bad = 0; worst = 0.0; n_members = 0
for n, seeds in ((1500, (1, 2, 3)), (3000, (1, 2, 3)), (10000, (12345, 21))):
    for seed in seeds:
        net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        u0 = np.zeros(n); u0[0] = 1.0
        T = np.array([700.0, 900.0, 1000.0, 1200.0, 1400.0, 1600.0])
        p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                           solve_chunkstep=1e-3, maxiters=200000, save_interval=1e-3, dtmin=1e-30)
        t0 = time.perf_counter()
        te, ue, ns, rcs, sts = h.solve_ensemble(p, np.tile(u0, (len(T), 1)), T=T)
        we = time.perf_counter() - t0
        for i, Ti in enumerate(T):
            h.rates_at(float(Ti))
            ts, us, rc, st, _ = h.solve(p, u0)
            e = float((np.abs(ue[i] - us) / (1e-10 + 1e-8 * np.abs(us))).max()) if rc == 0 and rcs[i] == 0 else float("nan")
            rec = {"kind": "static", "n": n, "seed": seed, "T": float(Ti), "rc": [int(rcs[i]), rc], "retries": [sts[i]["n_retries"], st["n_retries"]],
                   "steps": [sts[i]["n_steps"], st["n_steps"]], "factor": [sts[i]["n_factor"], st["n_factor"]], "units_apart": e}
            bad += (rcs[i] != 0) or (rc != 0) or not (e < 300)
            worst = max(worst, e if e == e else 0.0); n_members += 1
            print(json.dumps(rec), flush=True)
        print(json.dumps({"kind": "ensemble_wall", "n": n, "seed": seed, "members": len(T), "wall_s": round(we, 3)}), flush=True)
        # shared ramp 600 -> 1100 K: two members with different initial states
        tst = np.arange(21) * 5e-4
        pr = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                            solve_chunkstep=2.5e-3, maxiters=200000, save_interval=2.5e-3, dtmin=1e-30)
        U0 = np.tile(u0, (2, 1)); U0[1, 1] = 0.2; U0[1, 0] = 0.8
        te, ue, ns, rcs, sts = h.solve_ensemble(pr, U0, tstops=tst, T_stops=600.0 + 5e4 * tst)
        for i in range(2):
            ts, us, rc, st, _ = h.solve(pr, U0[i], tstops=tst, T_stops=600.0 + 5e4 * tst)
            e = float((np.abs(ue[i] - us) / (1e-10 + 1e-8 * np.abs(us))).max()) if rc == 0 and rcs[i] == 0 else float("nan")
            print(json.dumps({"kind": "ramp", "n": n, "seed": seed, "member": i, "rc": [int(rcs[i]), rc], "steps": [sts[i]["n_steps"], st["n_steps"]], "units_apart": e}), flush=True)
            bad += (rcs[i] != 0) or (rc != 0) or not (e < 300)
            worst = max(worst, e if e == e else 0.0); n_members += 1
        h.close()
print(json.dumps({"summary": True, "members_checked": n_members, "failures_or_mismatches": int(bad), "largest_distance_member_vs_solo_units": worst}))
