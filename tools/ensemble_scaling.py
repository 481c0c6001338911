"""Ensemble throughput on one GPU: K handles on K host threads, each solving the first chunks of a C3 replica.
Usage: [GPU_MAX_HW_QUEUES=n] python tools/ensemble_scaling.py [chunks=2] [Ks=1,2,4,8,16]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

nck = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]
N, R = 10000, 50000
net, Ea, A = synthetic_crn(N, R)
u0 = np.zeros(N); u0[0] = 1.0
pars = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nck, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                      ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
hs = [capi.HipNetwork.from_flat(net) for _ in range(max(Ks))]
for i, h in enumerate(hs):
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(1000.0 + 10.0 * i)
res = {}
for K in Ks:
    gate = threading.Barrier(K + 1)
    outs = [None] * K

    def work(i):
        hs[i].solve(pars, u0)
        gate.wait()
        outs[i] = hs[i].solve(pars, u0)
    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    [t.start() for t in th]
    gate.wait()
    t0 = time.perf_counter()
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    res[K] = {"wall_s": round(wall, 4), "solves_per_s": round(K / wall, 2), "steps": outs[0][3]["n_steps"]}
print(json.dumps({"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "chunks": nck, "res": res}))
