"""CPU experiment (oracle/bdf.py only - nothing of the product runs here): does a cap on the error estimate of the WORST species remove the
negative excursions that end in a tolerance retry? 15 collapse-prone 1 000-species solves (10 chunks of 1 ms, static rates) with the
corrector tolerance, the relative tolerance and the cap given on the command line; prints per-solve and total steps / rejections / retries.
Usage: python tools/err_cap_experiment.py <corrector tolerance, e.g. 0.03> <rtol, e.g. 1e-8> <cap in error weights, 0 = none>
Results of round 5: docs/DESIGN_HISTORY.md R5.13."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kinetica_jl_amd.synth import synthetic_crn
from oracle import oracle as orc
from oracle import bdf as ob

ntol, rtol, cap = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
_set_tols, _init = ob.OracleBDF.set_tols, ob.OracleBDF.__init__


def set_tols(self, a, r):
    _set_tols(self, a, r)
    self.newton_tol = max(10 * ob.EPS / r, ntol)


def init(self, *a, **kw):
    _init(self, *a, **kw)
    self.err_cap = cap if cap > 0 else None


ob.OracleBDF.set_tols, ob.OracleBDF.__init__ = set_tols, init
CASES = [(3, 1400.0), (6, 1000.0), (9, 800.0), (3, 1800.0), (6, 1400.0), (3, 1200.0), (12345, 1000.0), (12, 900.0), (13, 1100.0), (14, 1100.0),
         (14, 1300.0), (16, 900.0), (16, 1100.0), (17, 900.0), (17, 1100.0)]
tot = dict(steps=0, rhs=0, rejections=0, retries=0, corrector_failures=0)
for seed, T in CASES:
    net, Ea, A = synthetic_crn(1000, 5000, seed=seed)
    k = orc.arrhenius(Ea, A, T, k_max=1e12)
    on = orc.OracleNetwork.from_flat(net)
    u0 = np.zeros(1000); u0[0] = 1.0
    p = {"tspan": (0.0, 1e-2), "abstol": rtol * 1e-2, "reltol": rtol, "solve_chunks": True, "solve_chunkstep": 1e-3, "save_interval": 1e-3, "dtmin": 1e-30}
    t, u, rc, s = ob.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), 1000, p, u0, k0=k)
    tot["steps"] += s["n_steps"]; tot["rhs"] += s["n_rhs"]; tot["rejections"] += s["n_rejected"]; tot["retries"] += s.get("n_retries", 0)
    tot["corrector_failures"] += s["n_newton_fail"]
    print(json.dumps({"seed": seed, "T": T, "rc": rc, "retries": s.get("n_retries"), "steps": s["n_steps"], "rejections": s["n_rejected"], "umin": float(u.min())}), flush=True)
print("TOTAL", json.dumps({"corrector_tolerance": ntol, "rtol": rtol, "cap": cap, **tot}))
