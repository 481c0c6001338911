"""Lockstep ensemble of LARGE networks (kin_solve_ensemble beyond the resident kernel's size, ensemble.cpp): members against solo
kin_solve runs of the same inputs, and throughput. Usage: python tools/ensemble_batched_check.py [N=10000] [K list...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

if not os.environ.get("ENS_DEFAULT_ROUTE"):        # this tool measures the lockstep rounds; K <= 12 would otherwise be K kin_solve calls on threads
    os.environ.setdefault("KIN_ENSEMBLE_BATCHED", "1")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
Ks = [int(a) for a in sys.argv[2:]] or [4, 16]
net, Ea, A = synthetic_crn(N, 5 * N)
h = capi.HipNetwork.from_flat(net)
h.set_arrhenius(Ea, A, k_max=1e12)
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                   solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
for K in Ks:
    T = 1000.0 + 10.0 * np.arange(K)
    U0 = np.tile(u0, (K, 1))
    t0 = time.perf_counter()
    t, u, ns, rcs, sts = h.solve_ensemble(p, U0, T=T)
    first = time.perf_counter() - t0
    walls = []
    for _ in range(int(os.environ.get("ENS_REPEATS", "1"))):
        t0 = time.perf_counter()
        t, u, ns, rcs, sts = h.solve_ensemble(p, U0, T=T)
        walls.append(time.perf_counter() - t0)
    wall = min(walls)
    worst = 0.0
    same_t = None
    for i in ([] if os.environ.get("ENS_NO_SOLO") else sorted(set([0, K // 2, K - 1]))):
        h.rates_at(float(T[i]))
        ts, us, rc, st, _ = h.solve(p, u0)
        e = np.abs(u[i] - us) / (1e-10 + 1e-8 * np.abs(us))
        worst = max(worst, float(e.max()))
        same_t = bool(np.array_equal(ts, t))
    print(json.dumps({"N": N, "K": K, "first_s": first, "wall_s": wall, "walls": [round(w, 3) for w in walls], "solves_per_s": K / wall, "rcs_ok": int((rcs == 0).sum()), "times_equal": same_t,
                      "units_vs_solo_max": worst, "steps": [s["n_steps"] for s in sts][:4], "factor": [s["n_factor"] for s in sts][:4],
                      "slots": sts[0]["lu_slots"]}), flush=True)
h.close()
