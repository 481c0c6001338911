"""CPU tests: the oracle's BDF restatement pinned against SciPy's independent implementation of
the same published algorithm (step-for-step), against closed forms and against the committed
high-accuracy truth trajectories."""
import json
import os

import numpy as np
import pytest
from scipy.integrate import solve_ivp

from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import bdf as obdf
from oracle import oracle as orc

ROBER_REACS = [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]]
ROBER_PRODS = [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]]
ROBER_K = np.array([0.04, 3e7, 1e4])


def rober():
    return orc.OracleNetwork.from_flat(from_lists(3, ROBER_REACS, ROBER_PRODS))


def test_step_sequence_matches_scipy_bdf():
    on = rober()
    k = ROBER_K
    y0 = np.array([1.0, 0.0, 0.0])
    tf = 40.0
    ref = solve_ivp(lambda t, y: on.rhs(k, y), (0, tf), y0, method="BDF", jac=lambda t, y: on.jac(k, y).toarray(),
                    rtol=1e-8, atol=1e-10)
    b = obdf.OracleBDF(lambda y: on.rhs(k, y), lambda y: on.jac(k, y), 3, 1e-10, 1e-8, scipy_newton=True)
    assert b.restart(0.0, y0, tf)
    b.iters_left = 10 ** 6
    ts, ys = [0.0], [y0]
    while b.t < tf:
        assert b.step(tf) == "ok"
        ts.append(b.t); ys.append(b.D[0].copy())
        b.select_order()
    assert len(ts) == len(ref.t)
    np.testing.assert_allclose(ts, ref.t, rtol=1e-9)
    np.testing.assert_allclose(np.array(ys), ref.y.T, rtol=1e-7, atol=1e-14)


def test_first_order_decay_closed_form():
    # A -> B with k: A(t) = exp(-k t)
    on = orc.OracleNetwork.from_flat(from_lists(2, [[(0, 1)]], [[(1, 1)]]))
    k = np.array([3.0])
    pars = dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.1, save_interval=0.05)
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             2, pars, [1.0, 0.0], k0=k)
    assert rc == 0 and len(t) == 21
    np.testing.assert_allclose(t, np.arange(21) * 0.05, atol=1e-15)
    np.testing.assert_allclose(u[:, 0], np.exp(-3.0 * t), rtol=1e-6)
    np.testing.assert_allclose(u.sum(axis=1), 1.0, rtol=1e-12)


def test_reversible_and_bimolecular_closed_forms():
    # A <-> B : A(t) = Aeq + (1-Aeq) exp(-(kf+kr) t)
    on = orc.OracleNetwork.from_flat(from_lists(2, [[(0, 1)], [(1, 1)]], [[(1, 1)], [(0, 1)]]))
    kf, kr = 2.0, 0.5
    pars = dict(tspan=(0.0, 2.0), solve_chunks=False, save_interval=0.25)
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             2, pars, [1.0, 0.0], k0=np.array([kf, kr]))
    aeq = kr / (kf + kr)
    np.testing.assert_allclose(u[:, 0], aeq + (1 - aeq) * np.exp(-(kf + kr) * t), rtol=1e-6)
    # 2A -> B : A(t) = A0 / (1 + 2 k A0 t)   (rate k A^2, dA/dt = -2 k A^2)
    on = orc.OracleNetwork.from_flat(from_lists(2, [[(0, 2)]], [[(1, 1)]]))
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             2, pars, [1.0, 0.0], k0=np.array([1.5]))
    np.testing.assert_allclose(u[:, 0], 1.0 / (1.0 + 2 * 1.5 * t), rtol=1e-6)
    # A + B -> C with A0 = B0: A(t) = A0 / (1 + k A0 t)
    on = orc.OracleNetwork.from_flat(from_lists(3, [[(0, 1), (1, 1)]], [[(2, 1)]]))
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             3, pars, [1.0, 1.0, 0.0], k0=np.array([0.7]))
    np.testing.assert_allclose(u[:, 0], 1.0 / (1.0 + 0.7 * t), rtol=1e-6)


def test_discrete_rate_updates_zero_order_hold():
    # A -> B with k switched 1 -> 4 at t = 0.5: A(1) = exp(-0.5) * exp(-2)
    on = orc.OracleNetwork.from_flat(from_lists(2, [[(0, 1)]], [[(1, 1)]]))
    pars = dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.25, save_interval=0.25)
    tst = np.array([0.0, 0.5])
    ks = [np.array([1.0]), np.array([4.0])]
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             2, pars, [1.0, 0.0], tstops=tst, k_of_stop=lambda i: ks[i])
    assert rc == 0
    np.testing.assert_allclose(t, [0, 0.25, 0.5, 0.75, 1.0])
    np.testing.assert_allclose(u[:, 0], [1, np.exp(-0.25), np.exp(-0.5), np.exp(-0.5 - 1.0), np.exp(-0.5 - 2.0)], rtol=1e-6)


def test_oracle_against_committed_truth(golden_dir):
    f = os.path.join(golden_dir, "truth_small.npz")
    z = np.load(f)
    # Robertson on a log-ish grid, complete-timespan solve
    on = rober()
    pars = dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0)
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             3, pars, [1.0, 0.0, 0.0], k0=ROBER_K)
    np.testing.assert_allclose(t, z["rober_t"])
    err = np.abs(u - z["rober_u"]) / (1e-10 + 1e-8 * np.abs(z["rober_u"]))
    assert err.max() < 100.0          # stated bound: within 100 x (abstol + reltol |u|) of the truth
    # 60-species synthetic CRN, chunkwise
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e3)
    u0 = np.zeros(60); u0[0] = 1.0
    pars = dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.125, save_interval=0.0625)
    t, u, rc, st = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                             60, pars, u0, k0=k)
    np.testing.assert_allclose(t, z["syn_t"])
    err = np.abs(u - z["syn_u"]) / (1e-10 + 1e-8 * np.abs(z["syn_u"]))
    assert err.max() < 100.0


def test_explicit_oracle_is_scipy_rk45_behind_the_driver():
    """The explicit oracle (kin_solve_explicit's reference) = SciPy's RK45 run segment by segment by the same
    chunk / save-grid driver: closed form A -> B, chunkwise grid, every-step output."""
    import numpy as np
    from oracle import bdf as obdf
    k = 3.0
    fun = lambda kk: (lambda y: np.array([-kk[0] * y[0], kk[0] * y[0]]))
    t, u, rc, st = obdf.solve_network_oracle(fun, None, 2, dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.25,
                                                                save_interval=0.125, explicit=True),
                                             [1.0, 0.0], k0=np.array([k]))
    assert rc == obdf.RET_SUCCESS and np.allclose(t, np.arange(9) * 0.125)
    assert np.max(np.abs(u[:, 0] - np.exp(-k * t)) / (1e-10 + 1e-8 * np.exp(-k * t))) < 100
    assert st["n_restarts"] == 4 and st["n_factor"] == 0
    t2, u2, rc2, st2 = obdf.solve_network_oracle(fun, None, 2, dict(tspan=(0.0, 1.0), solve_chunks=False, explicit=True),
                                                 [1.0, 0.0], k0=np.array([k]))
    assert rc2 == obdf.RET_SUCCESS and t2[0] == 0.0 and t2[-1] == 1.0 and len(t2) == st2["n_steps"] + 1
