"""KIN_WARM_RESTART=1 (difference history, order and step size carried across rate updates and chunk starts instead of the
reference's re-initialisation) against the default on the C4 ramp's first 20 chunks (200 rate updates) and on C3's first 30
chunks: wall, steps, factorisations, distance from the committed truths. One child process per setting (the switch is read once).
Usage: python tools/warm_restart_ab.py"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for w in ("0", "1"):
        subprocess.run([sys.executable, os.path.abspath(__file__), w], env=dict(os.environ, KIN_WARM_RESTART=w, KIN_RESIDENT="0"), check=False)
    sys.exit(0)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
net, Ea, A = synthetic_crn(10000, 50000)
h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
u0 = np.zeros(10000); u0[0] = 1.0
def kp(t1, chunk, save=-1.0, dtmin=0.0):
    return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                          solve_chunkstep=chunk, maxiters=1000000, save_interval=save, dtmin=dtmin)
def units(a, b): return np.abs(a - b) / (1e-10 + 1e-8 * np.abs(b))
# C4 prefix
tst = np.arange(201) * 1e-3; Tst = 500.0 + 50.0 * tst
z = np.load(os.path.join(ROOT, "tests", "golden", "truth_c4_long.npz"))
h.solve(kp(0.02, 1e-2, 5e-3, 1e-30), u0, tstops=tst[:21], T_stops=Tst[:21])
t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(0.2, 1e-2, 5e-3, 1e-30), u0, tstops=tst, T_stops=Tst); w = time.perf_counter() - t0
e = units(u[np.searchsorted(t, z["t"])], z["u"])
print(json.dumps({"warm": sys.argv[1], "case": "C4 first 20 chunks", "rc": rc, "wall_s": w, "steps": st["n_steps"], "factor": st["n_factor"], "rejected": st["n_rejected"],
                  "newton_fail": st["n_newton_fail"], "retries": st["n_retries"], "max_units": float(e.max()), "rms_units": float(np.sqrt((e ** 2).mean(axis=1)).max()), "umin": float(u.min())}), flush=True)
# C3 30 chunks
z = np.load(os.path.join(ROOT, "tests", "golden", "truth_c3_mid.npz"))
h.rates_at(1000.0)
h.solve(kp(2e-3, 1e-3), u0)
t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(0.03, 1e-3), u0); w = time.perf_counter() - t0
e = units(u[[int(np.argmin(np.abs(t - x))) for x in z["t"]]], z["u"])
print(json.dumps({"warm": sys.argv[1], "case": "C3 30 chunks", "rc": rc, "wall_s": w, "steps": st["n_steps"], "factor": st["n_factor"], "rejected": st["n_rejected"],
                  "newton_fail": st["n_newton_fail"], "retries": st["n_retries"], "max_units": float(e.max()), "rms_units": float(np.sqrt((e ** 2).mean(axis=1)).max()), "umin": float(u.min())}), flush=True)
