"""K independent handles on K Python threads (what bench.py's concurrent_replicas times) against kin_solve_ensemble's replica
route (K solve-only copies of ONE handle on K C++ threads) in the same process: first 2 chunks of the 10k-species network."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
N = 10000
net, Ea, A = synthetic_crn(N, 5 * N)
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                   solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
for K in (4, 8):
    hs = [capi.HipNetwork.from_flat(net) for _ in range(K)]
    for i, h in enumerate(hs):
        h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(1000.0 + 10.0 * i); h.solve(p, u0)
    walls = []
    for _ in range(3):
        gate = threading.Barrier(K + 1)
        th = [threading.Thread(target=lambda i=i: (gate.wait(), hs[i].solve(p, u0))) for i in range(K)]
        for x in th: x.start()
        t0 = time.perf_counter(); gate.wait()
        for x in th: x.join()
        walls.append(time.perf_counter() - t0)
    print(json.dumps({"form": "K handles on K python threads", "K": K, "walls": [round(w, 3) for w in walls], "solves_per_s": K / min(walls)}), flush=True)
    for h in hs: h.close()
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    U0 = np.tile(u0, (K, 1)); T = 1000.0 + 10.0 * np.arange(K)
    h.solve_ensemble(p, U0, T=T)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter(); h.solve_ensemble(p, U0, T=T); walls.append(time.perf_counter() - t0)
    print(json.dumps({"form": "kin_solve_ensemble, replica route", "K": K, "walls": [round(w, 3) for w in walls], "solves_per_s": K / min(walls)}), flush=True)
    h.close()
