"""A/B of integrator rules on the CPU port (oracle/cpu_bdf.cpp): C3 static chunkwise against the committed truths, C4 ramp prefix.
Environment switches are read by the port at solve time; run once per setting:
    KIN_CVHIN=1 KIN_ETAMX1=1e4 python tools/cpu_restart_ab.py c3 30
"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd.synth import synthetic_crn
from oracle import cpu_bdf, oracle as orc

def units(a, b, tol=1.0): return np.abs(a - b) / (1e-10 * tol + 1e-8 * tol * np.abs(b))

def report(name, t, u, rc, st, z, tol=1.0, wall=0.0):
    sel = [int(np.argmin(np.abs(t - tt))) for tt in z["t"]]
    e = units(u[sel], z["u"])
    top = np.argsort(z["u"].max(axis=0))[-50:]
    print(json.dumps({"case": name, "rc": rc, "wall": round(wall, 2), "steps": st["n_steps"], "factor": st["n_factor"], "nf": st["n_newton_fail"], "rej": st["n_rejected"],
                      "max": round(float(e.max()), 1), "rms": round(float(np.sqrt((e ** 2).mean(axis=1)).max()), 2), "p99.9": round(float(np.percentile(e, 99.9)), 1),
                      "top50_max": round(float(e[:, top].max()), 1)}), flush=True)

which = sys.argv[1]
if which == "c3":
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    truth = {2: "truth_c3", 30: "truth_c3_mid", 100: "truth_c3_long"}.get(n_chunks)
    net, Ea, A = synthetic_crn(10000, 50000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    p = dict(tspan=(0.0, 1e-3 * n_chunks), solve_chunks=True, solve_chunkstep=1e-3, abstol=1e-10 * tol, reltol=1e-8 * tol, maxiters=1000000, dtmin=0.0 if tol == 1.0 else 1e-30)
    t0 = time.time(); t, u, rc, st = cs.solve(p, u0, k0=k); w = time.time() - t0
    if truth:
        report(f"c3_{n_chunks}_x{tol:g}", t, u, rc, st, np.load(f"tests/golden/{truth}.npz"), wall=w)
    else:
        print(json.dumps({"case": f"c3_{n_chunks}", "rc": rc, "wall": round(w, 2), "steps": st["n_steps"], "factor": st["n_factor"], "nf": st["n_newton_fail"]}))
elif which == "c4":
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    z = np.load("tests/golden/truth_c4.npz" if n_chunks == 3 else "tests/golden/truth_c4_long.npz")
    net, Ea, A = synthetic_crn(10000, 50000)
    tst = np.arange(0, 10 * n_chunks + 1) * 1e-3
    ks = orc.rate_table(Ea, A, 500.0 + 50.0 * tst, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    p = dict(tspan=(0.0, 1e-2 * n_chunks), solve_chunks=True, solve_chunkstep=1e-2, save_interval=5e-3, dtmin=1e-30, maxiters=1000000)
    t0 = time.time(); t, u, rc, st = cs.solve(p, u0, tstops=tst, k_table=ks); w = time.time() - t0
    report(f"c4_{n_chunks}", t, u, rc, st, z, wall=w)
elif which == "small":
    n = int(sys.argv[2]); n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    net, Ea, A = synthetic_crn(n, 5 * n)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    p = dict(tspan=(0.0, 1e-3 * n_chunks), solve_chunks=True, solve_chunkstep=1e-3, maxiters=1000000)
    pt = dict(p, abstol=1e-13, reltol=1e-11, dtmin=1e-300, adaptive_tols=False)
    t0 = time.time(); t, u, rc, st = cs.solve(p, u0, k0=k); w = time.time() - t0
    env = {k_: os.environ.pop(k_) for k_ in ("KIN_CVHIN", "KIN_ETAMX1", "KIN_H0_DECADE") if k_ in os.environ}
    tt, ut, rct, stt = cs.solve(pt, u0, k0=k)
    report(f"small_{n}_{n_chunks}", t, u, rc, st, {"t": tt, "u": ut}, wall=w)
