"""High-accuracy truth trajectories for the small known-answer networks (SciPy Radau, rtol 1e-12,
analytic Jacobian from the oracle). Run in the build container; output truth_small.npz is
committed. The reference itself cannot produce these (Julia, not runnable here)."""
import os
import sys

import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from kinetica_jl_amd.synth import from_lists, synthetic_crn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def truth(on, k, y0, ts):
    sol = solve_ivp(lambda t, y: on.rhs(k, y), (ts[0], ts[-1]), y0, method="Radau", jac=lambda t, y: on.jac(k, y).toarray(),
                    rtol=1e-12, atol=1e-16, t_eval=ts)
    assert sol.success
    return sol.y.T


def main():
    out = {}
    on = orc.OracleNetwork.from_flat(from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]],
                                                [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]]))
    ts = np.arange(11) * 4.0
    out["rober_t"] = ts
    out["rober_u"] = truth(on, np.array([0.04, 3e7, 1e4]), [1.0, 0.0, 0.0], ts)
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e3)
    u0 = np.zeros(60); u0[0] = 1.0
    ts = np.arange(17) * 0.0625
    out["syn_t"] = ts
    out["syn_u"] = truth(on, k, u0, ts)
    # temperature ramp on the same network: k switched at every 0.125 (zero-order hold)
    T = 800.0 + 400.0 * np.arange(8) * 0.125
    pieces = [u0]
    y = u0
    for i in range(8):
        ki = orc.arrhenius(Ea, A, T[i], k_max=1e3)
        seg = truth(on, ki, y, np.array([0.0, 0.0625, 0.125]))
        pieces += [seg[1], seg[2]]
        y = seg[2]
    out["ramp_t"] = ts
    out["ramp_T"] = T
    out["ramp_u"] = np.array(pieces)
    np.savez_compressed(os.path.join(HERE, "truth_small.npz"), **out)
    print("wrote truth_small.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
