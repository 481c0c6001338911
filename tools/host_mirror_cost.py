"""End-to-end time of the host mirror's solve_network (kinetica_jl_amd/solving.py) on a C3-size network, against the kin_solve
inside it: what the Python stand-in for the Julia host code adds (network copy, cutoff, u0, handle setup, result copy)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinetica_jl_amd import conditions as C  # noqa: E402
from kinetica_jl_amd import solving as S  # noqa: E402
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402

N, R = int(sys.argv[1]), int(sys.argv[2])
net, Ea, A = synthetic_crn(N, R)
t0 = time.perf_counter()
sd = S.SpeciesData.from_names([f"S{i}" for i in range(N)])
rd = S.RxData.from_flat(net)
t_build = time.perf_counter() - t0
calc = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e12)
pars = S.ODESimulationParams(tspan=(0.0, 2e-2), u0={"S0": 1.0}, solve_chunks=True, solve_chunkstep=1e-3, save_interval=1e-3, low_k_cutoff="none")
for rep in range(2):
    t0 = time.perf_counter()
    res = S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 1000.0}), calc), sd, rd)
    dt = time.perf_counter() - t0
    print(f"solve_network call {rep}: {dt:.3f} s total, kin_solve inside {res.sol.stats['wall_seconds']:.3f} s, retcode {res.sol.retcode}, "
          f"{len(res.sol.t)} saved points; RxData.from_flat + SpeciesData {t_build:.3f} s")
