"""GPU tests of the boundary details added in round 2: `dtmin` (methods.jl:164, 232, 694, 770) with its
DtLessThanMin -> tolerance-retry path, the synchronising step-end hand-over, the Arrhenius cap at its limits,
return_integrator with continuous rate updates, and the explicit-solver guard."""
import os

import numpy as np
import pytest

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import bdf as obdf
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def kp(tspan, chunks=True, chunkstep=1e-3, save=None, abstol=1e-10, reltol=1e-8, maxiters=100000, adaptive=True, dtmin=0.0):
    return capi.KinParams(tspan0=tspan[0], tspan1=tspan[1], abstol=abstol, reltol=reltol, adaptive_tols=int(adaptive),
                          update_tols=0, solve_chunks=int(chunks), ban_negatives=0, solve_chunkstep=chunkstep,
                          maxiters=maxiters, save_interval=-1.0 if save is None else save, dtmin=dtmin)


def errscale(u, ref, abstol=1e-10, reltol=1e-8):
    return (np.abs(u - ref) / (abstol + reltol * np.abs(ref))).max()


ROBER = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
ROBER_K = np.array([0.04, 3e7, 1e4])


def oracle_rober(pars):
    on = orc.OracleNetwork.from_flat(ROBER)
    return obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), 3, pars,
                                     [1.0, 0.0, 0.0], k0=ROBER_K)


def test_dtmin_ends_in_dtlessthanmin_and_the_retry_loop_answers():
    """A dtmin far above what Robertson's initial transient needs: the first step is RAISED to dtmin (CVodeSetMinStep
    semantics), the error test pushes it below -> DtLessThanMin; adaptive_solve! (solve_utils.jl:376-424) divides the
    tolerances by 10 and retries, 5 attempts in all, then "ODE solution failed."."""
    h = capi.HipNetwork.from_flat(ROBER)
    h.set_rates(ROBER_K)
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, dtmin=1.0), [1.0, 0.0, 0.0])
    assert status == capi.KIN_ERR_SOLVE_FAILED and rc == 2 and capi.RETCODE_NAMES[rc] == "DtLessThanMin"
    assert st["n_retries"] == 4
    assert st["final_abstol"] == pytest.approx(1e-14) and st["final_reltol"] == pytest.approx(1e-12)
    # the oracle takes the same path
    to, uo, rco, sto = oracle_rober(dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0, dtmin=1.0))
    assert rco == obdf.RET_DTMIN and sto["n_retries"] == 4
    # adaptive_tols = false: no retry (solve_utils.jl:403-405)
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, dtmin=1.0, adaptive=False), [1.0, 0.0, 0.0])
    assert status == capi.KIN_ERR_SOLVE_FAILED and rc == 2 and st["n_retries"] == 0
    # a dtmin the problem can live with changes nothing but the first steps: still within the stated tolerance of the
    # default run, and the default is eps(tspan[end]) (complete) / eps(solve_chunkstep) (chunkwise), bit for bit
    t0, u0, rc0, st0, _ = h.solve(kp((0.0, 40.0), chunks=False, save=4.0), [1.0, 0.0, 0.0])
    t1, u1, rc1, st1, _ = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, dtmin=float(np.spacing(40.0))), [1.0, 0.0, 0.0])
    assert rc0 == 0 and rc1 == 0 and np.array_equal(u0, u1) and st0["n_steps"] == st1["n_steps"]
    t2, u2, rc2, st2, _ = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, dtmin=1e-7), [1.0, 0.0, 0.0])
    assert rc2 == 0 and errscale(u2, u0) < 100
    tc0, uc0, *_ = h.solve(kp((0.0, 40.0), True, 10.0, 5.0), [1.0, 0.0, 0.0])
    tc1, uc1, *_ = h.solve(kp((0.0, 40.0), True, 10.0, 5.0, dtmin=float(np.spacing(10.0))), [1.0, 0.0, 0.0])
    assert np.array_equal(uc0, uc1)
    h.close()


def test_synchronising_hand_over_gives_the_same_trajectory(monkeypatch):
    """KIN_NO_FAST_SYNC=1 forces the copy + hipStreamSynchronize hand-over at the end of every step attempt (the
    fallback of the pinned-memory sequence number): same kernels, same numbers. (Host-driven integrator forced: the
    resident kernel that otherwise takes a network of this size has no hand-over.)"""
    monkeypatch.setenv("KIN_RESIDENT", "0")
    net, Ea, A = synthetic_crn(300, 1500)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(300); u0[0] = 1.0
    h = capi.HipNetwork.from_flat(net)
    h.set_rates(k)
    t, u, rc, st, _ = h.solve(kp((0.0, 3e-3)), u0)
    h.close()
    os.environ["KIN_NO_FAST_SYNC"] = "1"
    try:
        h = capi.HipNetwork.from_flat(net)
        h.set_rates(k)
        t2, u2, rc2, st2, _ = h.solve(kp((0.0, 3e-3)), u0)
        h.close()
    finally:
        del os.environ["KIN_NO_FAST_SYNC"]
    assert rc == 0 and rc2 == 0 and np.array_equal(u, u2)
    assert all(st[q] == st2[q] for q in ("n_steps", "n_rejected", "n_factor", "n_linsolve", "n_jac"))


def test_arrhenius_cap_at_its_limits():
    """k = 1/(1/k_max + 1/k_r) (calculator.jl:225): k_r = inf gives k_max, k_r = 0 gives 0 - on the device as in the
    reference's formula (the oracle evaluates it literally)."""
    Ea = np.array([0.0, 0.0, 5.0e6, 1.0e5])
    A = np.array([1e300, 1e-320, 1e10, 1e10])           # A N_A overflows / is subnormal / exp underflows / ordinary
    for T in (300.0, 1000.0):
        with np.errstate(over="ignore", divide="ignore"):
            ref = orc.arrhenius(Ea, A, T, k_max=1e12)
        got = capi.arrhenius_eval(Ea, A, T, k_max=1e12)
        assert np.all(np.isfinite(got)) and got[0] == 1e12
        np.testing.assert_allclose(got, ref, rtol=4e-15, atol=0)
    # the table kernel agrees (its own fast arithmetic: bound of the parity test in test_gpu_parity.py)
    net = from_lists(2, [[(0, 1)]] * 4, [[(1, 1)]] * 4)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    tab = h.rate_table(np.array([300.0, 1000.0]))
    with np.errstate(over="ignore", divide="ignore"):
        ref = orc.rate_table(Ea, A, np.array([300.0, 1000.0]), k_max=1e12)
    assert np.all(np.isfinite(tab)) and np.all(tab[:, 0] == 1e12)
    np.testing.assert_allclose(tab, ref, rtol=1e-11, atol=0)
    h.close()


def test_return_integrator_with_continuous_rates_and_the_explicit_guard():
    """return_integrator=true on a continuous-rate VariableODESolve (methods.jl:363-458 with :445-449): stepping the
    integrator to the end reproduces kin_solve_continuous; solver='RK45' is refused on these paths instead of being
    silently replaced by the BDF."""
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    Ea, A = np.array([8.0e4]), np.array([1.0e-17])
    sd = S.SpeciesData.from_names(["A", "B"])
    rd = S.RxData(1, [[1]], [[2]], [[1]], [[1]])

    def method(**kw):
        cs = C.ConditionSet({"T": C.LinearGradientProfile(rate=100.0, X_start=500.0, X_end=700.0)})
        pars = S.ODESimulationParams(tspan=(0.0, 2.0), u0=[1.0, 0.0], solve_chunks=False, low_k_cutoff="none", **kw)
        return S.VariableODESolve(pars, cs, S.PrecalculatedArrheniusCalculator(Ea, A))

    res = S.solve_network(method(), sd, rd)
    assert res.sol.retcode == "Success" and set(res.sol_vcs) == {"T"}
    with S.solve_network(method(), sd, rd, return_integrator=True) as integ:
        assert integ.t == 0.0
        ts = []
        while integ.step(1) == 1:
            ts.append(integ.t)
        assert np.array_equal(np.array(ts), res.sol.t[1:])           # saveat = []: every accepted step
        assert integ.t == 2.0 and integ.retcode == "Success" and np.array_equal(integ.u, res.sol.u[-1])
    for ri in (False, True):
        with pytest.raises(ValueError):
            S.solve_network(method(solver="RK45"), sd, rd, return_integrator=ri)
    cs = C.ConditionSet({"T": 600.0})
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0, 0.0], solve_chunks=False, low_k_cutoff="none", solver="RK45")
    with pytest.raises(ValueError):
        S.solve_network(S.StaticODESolve(pars, cs, S.PrecalculatedArrheniusCalculator(Ea, A)), sd, rd, return_integrator=True)


def _autocatalytic(order):
    """A + B -> 2A (species order chosen by `order`): dA/dt = +k A B, so J_AA = k B > 0 and the Newton matrix I - c J has
    the diagonal entry 1 - c k B, which vanishes at c = 1 / (k B)."""
    a, b = order
    net = from_lists(2, [[(a, 1), (b, 1)]], [[(a, 2)]])
    return net, a, b


def test_vanishing_pivot_is_detected_and_answered():
    """Static (diagonal) pivoting meets a zero pivot: the factorisation raises the device flag (both the sparse
    elimination and the dense Gauss-Jordan check their multipliers), kin_newton_solve reports it, and kin_solve answers
    with a fresh Jacobian and half the step instead of iterating on garbage."""
    # species 0 is eliminated first (sparse round), species 1 ends in the dense Schur block: with A first the zero is the
    # sparse pivot 1 - c k B (c = 1/4 exactly); with B first it is the 1 x 1 Schur complement, det(I - c J) = 1 - 3 c
    for order, c_sing in (((0, 1), 0.25), ((1, 0), 1.0 / 3.0)):
        net, a, b = _autocatalytic(order)
        h = capi.HipNetwork.from_flat(net)
        h.set_rates([2.0])
        u = np.zeros(2); u[a] = 0.5; u[b] = 2.0                      # k B = 4, k A = 1
        rhs = np.array([1.0, -1.0])
        x = h.newton_solve(0.125, u, rhs)                            # regular
        on = orc.OracleNetwork.from_flat(net)
        M = np.eye(2) - 0.125 * on.jac(np.array([2.0]), u).toarray()
        np.testing.assert_allclose(M @ x, rhs, rtol=1e-13, atol=1e-15)
        with pytest.raises(capi.KineticaHipError) as e:
            h.newton_solve(c_sing, u, rhs)
        assert e.value.code == capi.KIN_ERR_SOLVE_FAILED and "pivot" in str(e.value)
        x2 = h.newton_solve(0.125, u, rhs)                           # the handle is usable afterwards
        np.testing.assert_array_equal(x, x2)
        h.close()
    # inside kin_solve: the response to the flag (drop the slot, fresh Jacobian at the next predictor, half the step).
    # An accuracy-controlled integration of mass-action kinetics keeps c J_ii << 1 even on an autocatalytic species (the
    # first steps have c k B ~ 3e-5 whatever k is, later ones c k B < 0.1), so the flag is raised by fault injection at
    # the 6th step attempt; everything downstream of the flag is the production path.
    net, a, b = _autocatalytic((0, 1))
    u0 = np.array([1e-3, 2.0])
    k = 3.0e3
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([k])
    t0, ur, rc0, st0, _ = h.solve(kp((0.0, 1e-3), chunks=False, save=1e-4), u0)
    h.close()
    os.environ["KIN_INJECT_BAD_PIVOT"] = "5"
    try:
        h = capi.HipNetwork.from_flat(net)
        h.set_rates([k])
        t, u, rc, st, status = h.solve(kp((0.0, 1e-3), chunks=False, save=1e-4), u0)
        h.close()
    finally:
        del os.environ["KIN_INJECT_BAD_PIVOT"]
    assert rc0 == 0 and st0["n_bad_pivot"] == 0
    assert status == capi.KIN_OK and rc == 0 and st["n_bad_pivot"] == 1
    # (the refreshed Jacobian is counted, but the two runs take different step sequences from there on, so their totals
    # are not comparable one to one)
    assert st["n_rejected"] >= 1 and st["n_jac"] >= 2
    # logistic growth of A at constant A + B: closed form. The solution grows 337-fold over the span and local errors grow
    # with it (an unstable direction), hence the wider band than for the decaying test problems
    S_ = u0.sum()
    A = S_ / (1.0 + (S_ / u0[0] - 1.0) * np.exp(-k * S_ * t))
    assert errscale(u[:, 0], A) < 1000 and errscale(ur[:, 0], A) < 1000


@pytest.mark.parametrize("mode", ["fused", "explicit", "rounds"])
def test_newton_matrix_solve_against_sparse_direct(mode):
    """(I - c J(u)) x = b through the on-device LU against SciPy's sparse direct solver, for the three solve paths of
    csrc/lu.cpp: the fused three-launch solve (default), the five-launch solve with explicit triangular inverses
    (KIN_LU_FUSED=0) and the round-by-round substitution (KIN_LU_EXPLICIT=0). 1k and 3k species (hub rows long enough
    for the whole-workgroup gather path), c from the first steps of a restart to the end of a chunk."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    env = {"fused": {}, "explicit": {"KIN_LU_FUSED": "0"}, "rounds": {"KIN_LU_EXPLICIT": "0"}}[mode]
    os.environ.update(env)
    try:
        for n, r, seed in ((1000, 5000, 3), (3000, 15000, 4)):
            net, Ea, A = synthetic_crn(n, r, seed=seed)
            h = capi.HipNetwork.from_flat(net)
            h.set_arrhenius(Ea, A, k_max=1e12)
            k = h.rates_at(1000.0)
            on = orc.OracleNetwork.from_flat(net)
            rng = np.random.default_rng(seed)
            u = 10.0 ** rng.uniform(-8, -2, n)
            J = on.jac(k, u).tocsc()
            for c in (1e-12, 1e-8, 1e-5, 1e-3):
                b = rng.standard_normal(n)
                x = h.newton_solve(c, u, b)
                M = (sp.identity(n, format="csc") - c * J).tocsc()
                xr = spl.spsolve(M, b)
                # residual in the scale of the problem, and agreement with the pivoting solver
                res = np.abs(M @ x - b).max() / (np.abs(M).dot(np.abs(x)).max() + np.abs(b).max())
                assert res < 1e-13, (mode, n, c, res)
                assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max(), (mode, n, c)
            h.close()
    finally:
        for q in env:
            del os.environ[q]


@pytest.mark.parametrize("seed,T", [(6, 1000.0), (3, 1400.0)])
def test_solves_that_collapsed_under_looser_corrector_settings(seed, T):
    """Two of the 140 solves of tools/robustness_sweep.py (1k species, static, 10 chunks) ended in DtLessThanMin at every retry
    tolerance while the corrector accepted iterates at 0.05 of the error weight from reused factorisations contracting at up
    to 0.2 per iteration (first corrections of up to one unit on a remembered rate in the other case): unconverged
    iterates pile up in the difference history and the step size collapses. With the settings in force (0.03 / 0.15 /
    0.2) both run through in ~1 200 steps without a retry."""
    net, Ea, A = synthetic_crn(1000, 5000, seed=seed)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(T)
    u0 = np.zeros(1000); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                       ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3, dtmin=1e-30)
    t, u, rc, st, status = h.solve(p, u0)
    assert status == capi.KIN_OK and rc == 0 and st["n_retries"] == 0
    assert st["n_steps"] < 2000 and st["n_factor"] < 400
    m = u @ net.mass.astype(float)
    np.testing.assert_allclose(m, m[0], rtol=5e-7, atol=0)
    h.close()


@pytest.mark.parametrize("seed,T", [(14, 900.0), (14, 1500.0), (17, 1100.0)])
def test_retry_rescues_a_chunk_that_inherited_negative_concentrations(seed, T):
    """Loose tolerances (1e-8 / 1e-6) let a chunk end with concentrations of -1e-9; mass-action kinetics is unstable under
    them and the next chunk can start on a solution with a finite-time blow-up: these three solves failed at EVERY retry
    tolerance, each retry at the same local time. The rescue path now restarts the chunk with the negative entries of
    its start state set to zero (solver.cpp, solve_entry): they end with Success (after one to three retries on the
    build this was written for - the count itself is not asserted, the excursions are chaotic)."""
    net, Ea, A = synthetic_crn(1000, 5000, seed=seed)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(T)
    u0 = np.zeros(1000); u0[0] = 1.0
    p = capi.KinParams(tspan0=0.0, tspan1=1e-2, abstol=1e-8, reltol=1e-6, adaptive_tols=1, update_tols=0, solve_chunks=1,
                       ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=1e-3, dtmin=1e-30)
    t, u, rc, st, status = h.solve(p, u0)
    assert status == capi.KIN_OK and rc == 0 and len(t) == 11
    m = u @ net.mass.astype(float)
    np.testing.assert_allclose(m, m[0], rtol=1e-5, atol=0)      # clipping -1e-9 entries moves the invariant by ~1e-9 each
    h.close()
