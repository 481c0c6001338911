"""One-time costs of a network: kin_network_create (topology tables), the first kin_solve (symbolic LU analysis, plans, slot
allocation) against a second identical solve. Usage: python tools/setup_cost.py N R [chunks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinetica_jl_amd import capi  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

N, R = int(sys.argv[1]), int(sys.argv[2])
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.perf_counter(); net, Ea, A = synthetic_crn(N, R); t_syn = time.perf_counter() - t0
t0 = time.perf_counter(); h = capi.HipNetwork.from_flat(net); t_create = time.perf_counter() - t0
h.set_arrhenius(Ea, A, k_max=1e12); h.rates_at(1000.0)
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                   ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3, dtmin=1e-30)
t0 = time.perf_counter(); h.solve(p, u0); t1 = time.perf_counter() - t0
t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(p, u0); t2 = time.perf_counter() - t0
print(f"N={N} R={R}: synthetic_crn {t_syn:.2f} s, kin_network_create {t_create:.2f} s, first kin_solve {t1:.2f} s, second {t2:.2f} s "
      f"(rc {rc}, {st['n_steps']} steps): one-time setup inside the first solve ~{t1 - t2:.2f} s")
