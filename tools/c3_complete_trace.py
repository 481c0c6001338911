"""C3 as ONE integration over (0, t_end) with the per-attempt trace of the host-driven integrator (KIN_TRACE_CHUNK=0 is set
here): where the corrector failures of the complete-timespan solve sit.
Usage: python tools/c3_complete_trace.py [t_end=1.0] > trace.txt"""
import json, os, sys, time
os.environ["KIN_TRACE_CHUNK"] = "0"
os.environ["KIN_RESIDENT"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

t_end = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
net, Ea, A = synthetic_crn(10000, 50000)
u0 = np.zeros(10000); u0[0] = 1.0
h = capi.HipNetwork.from_flat(net)
h.set_arrhenius(Ea, A, k_max=1e12)
h.rates_at(1000.0)
p = capi.KinParams(tspan0=0.0, tspan1=t_end, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=0,
                   ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=t_end / 10, dtmin=1e-30)
t0 = time.perf_counter()
t, u, rc, st, status = h.solve(p, u0)
print(json.dumps({"wall_s": time.perf_counter() - t0, "rc": rc, "stats": st}))
h.close()
