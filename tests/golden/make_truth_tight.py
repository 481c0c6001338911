"""Truth for the tight-tolerance comparison of the two device integrators (ADVICE r4 / tests/test_gpu_resident.py): the synthetic
200-species network (seed 3) at 1000 K over (0, 10 ms), saved every millisecond, by SciPy's Radau IIA (dense Jacobian from the
oracle, corrector tolerance 0.03 as in make_truth_independent.py) at rtol 1e-11 / atol 1e-13 - ten times tighter than the
tolerances the device runs are compared at (rtol 1e-10 / atol 1e-12). `self_check` = its distance from the rtol 1e-10 run in
units of 1e-12 + 1e-10 |u|. Below rtol ~1e-11 every integrator crawls on the rounding floor of the right-hand side:
1e-10 takes 91 s and 209 k evaluations, 1e-11 tens of minutes.
    python tests/golden/make_truth_tight.py        -> tests/golden/truth_tight_200.npz"""
import os, sys, time
import numpy as np
from scipy.integrate import Radau, solve_ivp
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class RadauTol(Radau):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.newton_tol = 0.03


n, seed, T = 200, 3, 1000.0
net, Ea, A = synthetic_crn(n, 5 * n, seed=seed)
on = orc.OracleNetwork.from_flat(net)
k = orc.arrhenius(Ea, A, T, k_max=1e12)
u0 = np.zeros(n); u0[0] = 1.0
t_eval = np.arange(1, 11) * 1e-3
res = {}
for rt in (1e-10, 1e-11):
    t0 = time.time()
    sol = solve_ivp(lambda t, u: on.rhs(k, u), (0.0, 1e-2), u0, method=RadauTol, jac=lambda t, u: on.jac(k, u).toarray(), rtol=rt, atol=rt * 1e-2,
                    t_eval=t_eval, first_step=1e-22)
    assert sol.success
    res[rt] = sol.y.T
    print(f"rtol {rt:g}: {sol.nfev} rhs, {time.time() - t0:.0f} s", flush=True)
    if rt == 1e-10:
        np.savez_compressed(os.path.join(HERE, "truth_tight_200.npz"), t=t_eval, u=res[rt], rtol=rt, n=n, seed=seed, T=T, self_check=np.nan)
sc = float((np.abs(res[1e-10] - res[1e-11]) / (1e-12 + 1e-10 * np.abs(res[1e-11]))).max())
np.savez_compressed(os.path.join(HERE, "truth_tight_200.npz"), t=t_eval, u=res[1e-11], rtol=1e-11, n=n, seed=seed, T=T, self_check=sc)
print("wrote truth_tight_200.npz, 1e-10 against 1e-11:", sc, "units of 1e-12 + 1e-10 |u|")
