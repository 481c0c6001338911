"""Where the C4 ramp prefix deviates from its truth: worst species / save points, and an A/B over the
integrator's switches. Every variant runs in a child process (the switches are read when the solver is created).
Usage: python tools/ramp_diag.py            (all variants)       python tools/ramp_diag.py --one   (this process, current env)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

VARIANTS = [
    ("default", {}),
    ("no_lu_cache", {"KIN_LU_BAND": "0"}),
    ("single_slot_band", {"KIN_LU_CACHE_SLOTS": "1", "KIN_LU_BAND": "0.3"}),
    ("no_speculation", {"KIN_SPECULATE": "0"}),
    ("no_fused_newton", {"KIN_FUSE_NEWTON": "0"}),
    ("no_carried_rate", {"KIN_CARRY_RATE": "0"}),
    ("newton_tol_0.01", {"KIN_NEWTON_TOL": "0.01"}),
    ("rate_max_0.1", {"KIN_LU_RATE_MAX": "0.1"}),
    ("band_0.2", {"KIN_LU_BAND": "0.2"}),
    ("band_0.1", {"KIN_LU_BAND": "0.1"}),
    ("drift_0.1", {"KIN_LU_DRIFT": "0.1"}),
    ("lu_plain_substitution", {"KIN_LU_EXPLICIT": "0"}),
]


def units(u, ref):
    return np.abs(u - ref) / (1e-10 + 1e-8 * np.abs(ref))


def one():
    from kinetica_jl_amd import capi
    from kinetica_jl_amd.synth import synthetic_crn
    z = np.load(os.path.join(ROOT, "tests", "golden", "truth_c4.npz"))
    net, Ea, A = synthetic_crn(10000, 50000)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=3e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                       ban_negatives=0, solve_chunkstep=1e-2, maxiters=100000, save_interval=5e-3, dtmin=1e-30)
    t, u, rc, st, _ = h.solve(p, u0, tstops=z["tstops"], T_stops=z["T_stops"])
    sel = np.searchsorted(t, z["t"])
    e = units(u[sel], z["u"])
    flat = np.argsort(e, axis=None)[::-1][:12]
    worst = []
    for f in flat:
        si, sp = np.unravel_index(f, e.shape)
        worst.append({"save": int(si), "t": float(z["t"][si]), "species": int(sp), "units": float(e[si, sp]), "truth": float(z["u"][si, sp]),
                      "dev": float(u[sel][si, sp])})
    rec = {"rc": rc, "max_units": float(e.max()), "rms_units": float(np.sqrt((e ** 2).mean(axis=1)).max()),
           "p999": float(np.percentile(e, 99.9)), "p9999": float(np.percentile(e, 99.99)), "n_over_100": int((e > 100).sum()),
           "max_per_save": [float(x) for x in e.max(axis=1)], "steps": st["n_steps"], "factor": st["n_factor"],
           "rejected": st["n_rejected"], "newton_fail": st["n_newton_fail"], "wall_s": st["wall_seconds"], "worst": worst}
    print("RESULT " + json.dumps(rec), flush=True)
    h.close()


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        names = [a for a in sys.argv[1:] if not a.startswith("-")]
        for name, env in VARIANTS:
            if names and name not in names:
                continue
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, **env), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            rec = json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]}
            print(json.dumps({"variant": name, "env": env, **rec}), flush=True)
