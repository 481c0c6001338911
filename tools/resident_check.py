"""Resident integrator (one workgroup per trajectory) against the host-driven path, the CPU port and the truths; wall-clocks.
Usage: python tools/resident_check.py [sizes...]   (default 300 1000)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import oracle as orc


def kp(t1, chunk=1e-3, save=None, chunks=True, **kw):
    d = dict(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1 if chunks else 0,
             ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=-1.0 if save is None else save, dtmin=0.0)
    d.update(kw)
    return capi.KinParams(**d)


def units(u, ref):
    return float((np.abs(u - ref) / (1e-10 + 1e-8 * np.abs(ref))).max())


def timed(h, pars, u0, reps=3, **kw):
    h.solve(pars, u0, **kw)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = h.solve(pars, u0, **kw)
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [300, 1000]
    # Robertson
    rob = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
    z = np.load(os.path.join(ROOT, "tests", "golden", "truth_small.npz"))
    h = capi.HipNetwork.from_flat(rob)
    h.set_rates(np.array([0.04, 3e7, 1e4]))
    t, u, rc, st, _ = h.solve(kp(40.0, chunks=False, save=4.0), [1.0, 0.0, 0.0])
    print(json.dumps({"case": "robertson", "rc": rc, "steps": st["n_steps"], "units_vs_truth": units(u, z["rober_u"])}), flush=True)
    h.close()
    for n in sizes:
        net, Ea, A = synthetic_crn(n, 5 * n)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        h.rates_at(1000.0)
        u0 = np.zeros(n); u0[0] = 1.0
        pars = kp(20e-3)
        os.environ["KIN_RESIDENT"] = "0"
        t_host, (th, uh, rch, sth, _) = timed(h, pars, u0)
        os.environ.pop("KIN_RESIDENT")
        t_res, (tr, ur, rcr, str_, _) = timed(h, pars, u0)
        rec = {"case": f"static_{n}", "host_s": t_host, "resident_s": t_res, "rc": [rch, rcr],
               "steps": [sth["n_steps"], str_["n_steps"]], "factor": [sth["n_factor"], str_["n_factor"]],
               "newton_fail": [sth["n_newton_fail"], str_["n_newton_fail"]], "dense": [sth["lu_dense_dim"], str_["lu_dense_dim"]],
               "slots": [sth["lu_slots"], str_["lu_slots"]], "units_res_vs_host": units(ur, uh), "times_equal": bool(np.array_equal(tr, th))}
        print(json.dumps(rec), flush=True)
        # ensemble: K members, bit-identical to solo runs
        K = 8
        Ts = 1000.0 + 10.0 * np.arange(K)
        t0 = time.perf_counter()
        te, ue, ns, rcs, sts = h.solve_ensemble(kp(2e-3), np.tile(u0, (K, 1)), T=Ts)
        t_ens = time.perf_counter() - t0
        t0 = time.perf_counter()
        te, ue, ns, rcs, sts = h.solve_ensemble(kp(2e-3), np.tile(u0, (K, 1)), T=Ts)
        t_ens = time.perf_counter() - t0
        same = True
        for i in (0, K - 1):
            h.rates_at(float(Ts[i]))
            ts_, us_, rcs_, _, _ = h.solve(kp(2e-3), u0)
            same = same and np.array_equal(us_, ue[i]) and np.array_equal(ts_, te)
        print(json.dumps({"case": f"ensemble_{n}", "K": K, "wall_s": t_ens, "solves_per_s": K / t_ens, "rcs": rcs.tolist(),
                          "bit_identical_to_solo": bool(same), "steps": [s["n_steps"] for s in sts]}), flush=True)
        for K in (64, 256):
            Ts = np.linspace(900.0, 1300.0, K)
            h.solve_ensemble(kp(2e-3), np.tile(u0, (K, 1)), T=Ts)
            t0 = time.perf_counter()
            te, ue, ns, rcs, sts = h.solve_ensemble(kp(2e-3), np.tile(u0, (K, 1)), T=Ts)
            t_ens = time.perf_counter() - t0
            print(json.dumps({"case": f"ensemble_{n}", "K": K, "wall_s": t_ens, "solves_per_s": K / t_ens, "ok": int((rcs == 0).sum()),
                              "slots": sts[0]["lu_slots"], "steps_mean": float(np.mean([s["n_steps"] for s in sts]))}), flush=True)
        h.close()


if __name__ == "__main__":
    main()
