"""Diagnostic: C3 static solve, first chunks, GPU (kin_solve) or CPU oracle; dumps t/u/stats to an .npz
so the two can be compared offline. Usage: python tools/dev_vs_oracle.py gpu|cpu out.npz [n_chunks]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kinetica_jl_amd.synth import synthetic_crn

which, out = sys.argv[1], sys.argv[2]
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
N, R = 10000, 50000
net, Ea, A = synthetic_crn(N, R)
from oracle import oracle as orc
k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
u0 = np.zeros(N); u0[0] = 1.0
if which == "gpu":
    from kinetica_jl_amd import capi
    h = capi.HipNetwork.from_flat(net)
    h.set_rates(k)
    p = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                       ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=2.5e-4)
    t, u, rc, st, _ = h.solve(p, u0)
    h.close()
else:
    from oracle import bdf as obdf
    on = orc.OracleNetwork.from_flat(net)
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), N,
                                             dict(tspan=(0.0, 1e-3 * nch), solve_chunks=True, solve_chunkstep=1e-3, save_interval=2.5e-4),
                                             u0, k0=k)
print(which, rc, {q: st[q] for q in ("n_steps", "n_rejected", "n_factor", "n_newton_fail", "n_jac")})
np.savez(out, t=np.asarray(t), u=np.asarray(u))
