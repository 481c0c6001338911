"""Device vs oracle/bdf.py vs a 1000x tighter oracle run on small synthetic networks, in tolerance units of the default
tolerances: how far two implementations of the same algorithm sit from each other and from the truth."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
from oracle import oracle as orc, bdf as obdf
for (n, r, seed) in ((300, 1500, 12345), (300, 1500, 7), (1000, 5000, 12345)):
    net, Ea, A = synthetic_crn(n, r, seed=seed)
    h = capi.HipNetwork.from_flat(net); on = orc.OracleNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12); k = h.rates_at(1000.0)
    u0 = np.zeros(n); u0[0] = 1.0
    def P(tight): return capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10*tight, reltol=1e-8*tight, adaptive_tols=0, update_tols=0, solve_chunks=1, ban_negatives=0, solve_chunkstep=1e-3, maxiters=1000000, save_interval=5e-4, dtmin=1e-300)
    t, us, rc, st, _ = h.solve(P(1.0), u0)
    tt, ut, rct, stt, _ = h.solve(P(1e-3), u0)
    f = lambda kk: (lambda y: on.rhs(kk, y)); j = lambda kk: (lambda y: on.jac(kk, y))
    to, uo, rco, sto = obdf.solve_network_oracle(f, j, n, dict(tspan=(0.0, 2e-3), solve_chunks=True, solve_chunkstep=1e-3, save_interval=5e-4), u0, k0=k)
    to2, uo2, rco2, sto2 = obdf.solve_network_oracle(f, j, n, dict(tspan=(0.0, 2e-3), solve_chunks=True, solve_chunkstep=1e-3, save_interval=5e-4, abstol=1e-13, reltol=1e-11, dtmin=1e-300, adaptive_tols=False), u0, k0=k)
    un = lambda a, b: float((np.abs(a - b) / (1e-10 + 1e-8 * np.abs(b))).max())
    print(n, seed, "steps dev/orc", st["n_steps"], sto["n_steps"], "dev-orc", round(un(us, uo), 2), "dev-truth(orc)", round(un(us, uo2), 2),
          "orc-truth", round(un(uo, uo2), 2), "devtight-orctight", round(un(ut, uo2), 3) if ut.shape == uo2.shape else "n/a", flush=True)
    h.close()
