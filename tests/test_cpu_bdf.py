"""CPU tests pinning the compiled CPU baseline (oracle/cpu_bdf.cpp: BDF + KLU-style sparse LU) against the Python oracle
(oracle/bdf.py + SuperLU), which in turn is pinned against SciPy's BDF, closed forms and Radau truths
(tests/test_oracle_bdf.py): same algorithm, so the same step counts up to rounding, and trajectories within the stated
tolerance of each other. Both are test infrastructure; the product never loads them."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import bdf as obdf
from oracle import cpu_bdf
from oracle import oracle as orc


def oracle_solve(net, pars, u0, k0=None, tstops=None, ks=None):
    on = orc.OracleNetwork.from_flat(net)
    return obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                     net.n_species, pars, u0, k0=k0, tstops=tstops,
                                     k_of_stop=None if ks is None else (lambda i: ks[i]))


def units(u, ref):
    return (np.abs(u - ref) / (1e-10 + 1e-8 * np.abs(ref))).max()


def test_rhs_and_jacobian_match_the_c_oracle():
    net, Ea, A = synthetic_crn(300, 1500)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    on = orc.OracleNetwork.from_flat(net)
    cs = cpu_bdf.CpuSolver(net)
    u = 10.0 ** np.random.default_rng(0).uniform(-12, 0, 300)
    np.testing.assert_array_equal(cs.rhs(k, u), on.rhs(k, u))            # same loops, same order: bitwise
    J, Jo = cs.jac(k, u), on.jac(k, u)
    assert abs(J - Jo).max() <= 1e-13 * abs(Jo).max()


def test_sparse_lu_against_superlu_and_refactorisation():
    """(I - c J) x = b: first call = pivoting factorisation (AMD ordering, threshold partial pivoting), later calls reuse
    pattern and pivots (klu_refactor); both agree with SuperLU to the conditioning of the matrix, and the fill of the
    ordering is in SuperLU's class."""
    net, Ea, A = synthetic_crn(2000, 10000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    rng = np.random.default_rng(1)
    u = 10.0 ** rng.uniform(-12, 0, 2000)
    b = rng.standard_normal(2000)
    cs = cpu_bdf.CpuSolver(net)
    J = cs.jac(k, u)
    for i, c in enumerate((1e-9, 1e-6, 3e-6, 1e-4)):
        x, nnz = cs.newton_solve(c, k, u, b)
        M = (sp.identity(2000, format="csc") - c * J).tocsc()
        lu = spla.splu(M, permc_spec="MMD_AT_PLUS_A")
        xs = lu.solve(b)
        scale = np.abs(xs).max()
        assert np.abs(M @ x - b).max() <= 1e-9 * max(1.0, np.abs(M).max() * scale)
        assert np.abs(x - xs).max() <= 1e-7 * scale
        assert nnz <= 1.5 * (lu.L.nnz + lu.U.nnz)
    # a singular matrix is reported, not solved
    net2 = from_lists(2, [[(0, 1), (1, 1)]], [[(0, 2)]])
    cs2 = cpu_bdf.CpuSolver(net2)
    u2 = np.array([0.0, 2.0])                  # J = [[k B, 0], [-k B, 0]] at A = 0: I - c J singular at c = 1 / (k B)
    x, nnz = cs2.newton_solve(0.25, np.array([2.0]), u2, np.array([1.0, 1.0]))
    assert nnz == -1


def test_same_algorithm_as_the_python_oracle():
    # Robertson, complete timespan: identical step sequence (3 x 3 dense arithmetic is the same on both sides)
    rob = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
    K = np.array([0.04, 3e7, 1e4])
    pars = dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0)
    t, u, rc, st = cpu_bdf.CpuSolver(rob).solve(pars, [1.0, 0.0, 0.0], k0=K)
    to, uo, rco, sto = oracle_solve(rob, pars, [1.0, 0.0, 0.0], k0=K)
    assert rc == 0 and rco == 0 and st["n_steps"] == sto["n_steps"] and st["n_factor"] == sto["n_factor"]
    np.testing.assert_allclose(t, to)
    assert units(u, uo) < 1
    # synthetic CRN, chunkwise with the LU cache (default) and without it
    net, Ea, A = synthetic_crn(300, 1500)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(300); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    for kw in (dict(), dict(lu_band=0.0, lu_slots=0)):
        pars = dict(tspan=(0.0, 4e-3), **kw)
        t, u, rc, st = cs.solve(pars, u0, k0=k)
        to, uo, rco, sto = oracle_solve(net, pars, u0, k0=k)
        assert rc == 0 and rco == 0
        assert abs(st["n_steps"] - sto["n_steps"]) <= 0.02 * sto["n_steps"] + 2
        assert abs(st["n_factor"] - sto["n_factor"]) <= 0.1 * sto["n_factor"] + 3
        assert units(u, uo) < 100
    # discrete rate updates (zero-order hold, restart at every stop) and the save grid stitching
    tst = np.arange(8) * 0.5e-3
    ks = orc.rate_table(Ea, A, 800.0 + 50.0 * np.arange(8), k_max=1e12)
    pars = dict(tspan=(0.0, 4e-3), solve_chunkstep=1e-3, save_interval=2.5e-4)
    t, u, rc, st = cs.solve(pars, u0, tstops=tst, k_table=ks)
    to, uo, rco, sto = oracle_solve(net, pars, u0, tstops=tst, ks=ks)
    assert rc == 0 and rco == 0 and st["n_restarts"] == sto["n_restarts"] == 8
    np.testing.assert_allclose(t, to, rtol=0, atol=1e-18)
    assert units(u, uo) < 100
    # failure semantics: dtmin and maxiters end in the same retcodes after the same number of retries
    for bad in (dict(dtmin=1.0), dict(maxiters=5)):
        pars = dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0, **bad)
        t, u, rc, st = cpu_bdf.CpuSolver(rob).solve(pars, [1.0, 0.0, 0.0], k0=K)
        to, uo, rco, sto = oracle_solve(rob, pars, [1.0, 0.0, 0.0], k0=K)
        assert rc == rco != 0 and st["n_retries"] == sto["n_retries"] == 4
