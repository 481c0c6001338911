"""CPU tests of the resident integrator's controller (kinetica_jl_amd/csrc/resident_core.hpp - the code the GPU kernel of
resident.hip runs, one workgroup per trajectory) through its sequential replay (tests/native/resident_host.cpp), and of the
symbolic Newton-matrix factorisation of lu.cpp, which the replay executes numerically on the host:
  * the factorisation's three solve forms against a sparse direct solve,
  * the controller against oracle/cpu_bdf.cpp - the same algorithm written independently: same step, factorisation and
    failure counts, trajectories within the solver tolerance -, against the committed Radau truths, and its driver
    semantics (save grids, chunk stitching, discrete rate updates, retries, retcodes).
No GPU involved; the GPU tests (tests/test_gpu_resident.py) compare the device kernel with the same references."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import cpu_bdf
from oracle import oracle as orc
from tests.res_host import HostResident


def kp(t1, chunk=1e-3, save=None, chunks=True, t0=0.0, **kw):
    d = dict(tspan0=t0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1 if chunks else 0,
             ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=-1.0 if save is None else save, dtmin=0.0)
    d.update(kw)
    return capi.KinParams(**d)


def units(u, ref, atol=1e-10, rtol=1e-8):
    return (np.abs(u - ref) / (atol + rtol * np.abs(ref))).max()


ROB = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
ROB_K = np.array([0.04, 3e7, 1e4])


@pytest.mark.parametrize("env,mode", [({}, 0), ({"KIN_LU_FUSED": "0"}, 1), ({"KIN_LU_EXPLICIT": "0"}, 2)])
@pytest.mark.parametrize("n", [300, 1000])
def test_symbolic_factorisation_solves_the_newton_matrix(monkeypatch, env, mode, n):
    """(I - c J) x = b through the replayed SparseLU (independent-set rounds + dense Schur block + explicit triangular
    inverses + fused products) in its three solve forms, against SuperLU."""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    net, Ea, A = synthetic_crn(n, 5 * n)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    hr = HostResident(net)
    assert hr.info["solve_mode"] == mode and hr.info["m"] > 0 and hr.info["ns"] > 0
    rng = np.random.default_rng(1)
    u = 10.0 ** rng.uniform(-12, 0, n)
    b = rng.standard_normal(n)
    J = orc.OracleNetwork.from_flat(net).jac(k, u)
    for c in (1e-9, 1e-6, 1e-4):
        x, bad = hr.newton_solve(c, k, u, b)
        assert not bad
        M = (sp.identity(n, format="csc") - c * J).tocsc()
        xs = spla.splu(M).solve(b)
        scale = np.abs(xs).max()
        assert np.abs(M @ x - b).max() <= 1e-8 * max(1.0, np.abs(M).max() * scale)
        assert np.abs(x - xs).max() <= 1e-6 * scale
    hr.close()


def test_degenerate_structures():
    """Networks without a dense block, or with nothing but one: the three-species Robertson problem and a chain."""
    hr = HostResident(ROB)
    x, bad = hr.newton_solve(0.1, ROB_K, np.array([1.0, 1e-5, 0.1]), np.array([1.0, 2.0, 3.0]))
    J = orc.OracleNetwork.from_flat(ROB).jac(ROB_K, np.array([1.0, 1e-5, 0.1])).toarray()
    np.testing.assert_allclose((np.eye(3) - 0.1 * J) @ x, [1.0, 2.0, 3.0], rtol=1e-10)
    hr.close()
    n = 40
    chain = from_lists(n, [[(i, 1)] for i in range(n - 1)] + [[(i + 1, 1)] for i in range(n - 1)],
                       [[(i + 1, 1)] for i in range(n - 1)] + [[(i, 1)] for i in range(n - 1)])
    k = np.linspace(1.0, 3.0, 2 * (n - 1))
    hr = HostResident(chain)
    u = np.linspace(0.1, 1.0, n); b = np.cos(np.arange(n))
    x, bad = hr.newton_solve(0.3, k, u, b)
    J = orc.OracleNetwork.from_flat(chain).jac(k, u).toarray()
    np.testing.assert_allclose((np.eye(n) - 0.3 * J) @ x, b, rtol=1e-10, atol=1e-12)
    hr.close()


def test_vanishing_pivot_is_reported():
    net2 = from_lists(2, [[(0, 1), (1, 1)]], [[(0, 2)]])
    hr = HostResident(net2)
    x, bad = hr.newton_solve(0.25, np.array([2.0]), np.array([0.0, 2.0]), np.array([1.0, 1.0]))   # I - c J singular at c = 1 / (k B)
    assert bad
    hr.close()


def test_controller_takes_the_steps_of_the_independent_cpu_implementation():
    """Static chunkwise solve of a 300-species CRN: the replayed controller and oracle/cpu_bdf.cpp (same algorithm, own
    code, own LU) take the same number of steps / factorisations / corrector failures and agree within a few tolerance
    units; with the LU cache (default) and without."""
    net, Ea, A = synthetic_crn(300, 1500)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(300); u0[0] = 1.0
    hr = HostResident(net)
    cs = cpu_bdf.CpuSolver(net)
    t, u, rc, st = hr.solve(kp(4e-3), u0, k0=k)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 4e-3)), u0, k0=k)
    assert rc == 0 and rcc == 0
    np.testing.assert_array_equal(t, tc)
    assert abs(st["n_steps"] - stc["n_steps"]) <= 0.01 * stc["n_steps"] + 1
    assert abs(st["n_factor"] - stc["n_factor"]) <= 0.05 * stc["n_factor"] + 2
    assert abs(st["n_newton_fail"] - stc["n_newton_fail"]) <= 3 and st["n_restarts"] == stc["n_restarts"] == 4
    assert st["n_lu_reused"] > 0.8 * st["n_steps"]
    assert units(u, uc) < 20
    t1, u1, rc1, st1 = hr.solve(kp(4e-3), u0, k0=k, n_slots=1)       # one slot, no reuse band: a factorisation per change of c
    tc1, uc1, rcc1, stc1 = cs.solve(dict(tspan=(0.0, 4e-3), lu_band=0.0, lu_slots=0), u0, k0=k)
    assert rc1 == 0 and abs(st1["n_steps"] - stc1["n_steps"]) <= 0.01 * stc1["n_steps"] + 1
    assert st1["n_factor"] > 1.5 * st["n_factor"] and st1["n_lu_reused"] == 0
    assert abs(st1["n_factor"] - stc1["n_factor"]) <= 0.05 * stc1["n_factor"] + 2
    assert units(u1, uc1) < 20 and units(u1, u) < 100
    hr.close()


def test_corrector_tolerance_rule_is_the_same_in_every_implementation():
    """The corrector tolerance is 0.03 of the error weight down to rtol = 3.3e-9 and rises to 0.1 at 1e-9 (bdf_newton_frac,
    solver_kernels.hpp): A -> B at 1e-11 / 1e-9 and at 1e-12 / 1e-10 takes the same number of steps in the replayed resident
    controller, oracle/cpu_bdf.cpp and oracle/bdf.py (at the default tolerances tests/test_gpu_solve.py compares all four)."""
    from oracle import bdf as obdf
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    on = orc.OracleNetwork.from_flat(net)
    k = np.array([3.0]); u0 = np.array([1.0, 0.0])
    hr = HostResident(net)
    cs = cpu_bdf.CpuSolver(net)
    for atol, rtol, frac in ((1e-10, 1e-8, 0.03), (1e-10, 5e-9, 0.03), (1e-11, 2e-9, 0.05), (1e-11, 1e-9, 0.1), (1e-12, 1e-10, 0.1)):
        ob = obdf.OracleBDF(lambda y: y, lambda y: None, 2, atol, rtol)
        assert ob.newton_tol == pytest.approx(frac)
        pars = dict(tspan=(0.0, 1.0), solve_chunkstep=0.1, save_interval=0.05, abstol=atol, reltol=rtol, dtmin=1e-30)
        t, u, rc, st = hr.solve(kp(1.0, chunk=0.1, save=0.05, abstol=atol, reltol=rtol, dtmin=1e-30), u0, k0=k)
        tc, uc, rcc, stc = cs.solve(pars, u0, k0=k)
        to, uo, rco, sto = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), 2, pars, u0, k0=k)
        assert rc == rcc == rco == 0 and st["n_steps"] == stc["n_steps"] == sto["n_steps"], (rtol, st["n_steps"], stc["n_steps"], sto["n_steps"])
    hr.close()


def test_discrete_rate_updates_save_grid_and_truth(golden_dir):
    """60-species network under a temperature ramp (zero-order hold, restart at every stop, save grid finer than the stops)
    against the committed Radau truth, and against cpu_bdf on a finer grid with chunk stitching."""
    z = np.load(golden_dir + "/truth_small.npz")
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    u0 = np.zeros(60); u0[0] = 1.0
    hr = HostResident(net)
    hr.set_arrhenius(Ea, A, k_max=1e3)
    tst = np.arange(8) * 0.125
    # complete timespan with a save grid, rates from T_stops (Arrhenius on the fly)
    t, u, rc, st = hr.solve(kp(1.0, chunks=False, save=0.0625), u0, tstops=tst, T_stops=z["ramp_T"])
    assert rc == 0 and st["n_restarts"] == 8
    np.testing.assert_allclose(t, z["ramp_t"], rtol=0, atol=1e-15)
    assert units(u, z["ramp_u"]) < 100
    # chunkwise (4 chunks of 0.25), the same rates as a table
    ks = orc.rate_table(Ea, A, z["ramp_T"], k_max=1e3)
    t2, u2, rc2, st2 = hr.solve(kp(1.0, chunk=0.25, save=0.0625), u0, tstops=tst, k_table=ks)
    assert rc2 == 0 and st2["n_chunks"] == 4 and st2["n_restarts"] == 8
    np.testing.assert_allclose(t2, z["ramp_t"], rtol=0, atol=1e-15)
    assert units(u2, z["ramp_u"]) < 100
    cs = cpu_bdf.CpuSolver(net)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 1.0), solve_chunkstep=0.25, save_interval=0.0625), u0, tstops=tst, k_table=ks)
    assert rcc == 0 and abs(st2["n_steps"] - stc["n_steps"]) <= 0.02 * stc["n_steps"] + 2
    assert units(u2, uc) < 20
    # static, against the truth
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e3)
    t3, u3, rc3, _ = hr.solve(kp(1.0, chunks=False, save=0.0625), u0, k0=k)
    assert rc3 == 0 and units(u3, z["syn_u"]) < 100
    hr.close()


def test_robertson_and_failure_semantics(golden_dir):
    z = np.load(golden_dir + "/truth_small.npz")
    hr = HostResident(ROB)
    cs = cpu_bdf.CpuSolver(ROB)
    t, u, rc, st = hr.solve(kp(40.0, chunks=False, save=4.0), [1.0, 0.0, 0.0], k0=ROB_K)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0), [1.0, 0.0, 0.0], k0=ROB_K)
    assert rc == 0 and st["n_steps"] == stc["n_steps"] and st["n_factor"] == stc["n_factor"]
    np.testing.assert_allclose(t, z["rober_t"])
    assert units(u, z["rober_u"]) < 100 and units(u, uc) < 1
    # dtmin / maxiters: the same retcodes after the same number of tolerance retries (adaptive_solve!, solve_utils.jl:376-424)
    for bad, want in ((dict(dtmin=1.0), 2), (dict(maxiters=5), 1)):
        t, u, rc, st = hr.solve(kp(40.0, chunks=False, save=4.0, **bad), [1.0, 0.0, 0.0], k0=ROB_K)
        tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 40.0), solve_chunks=False, save_interval=4.0, **bad), [1.0, 0.0, 0.0], k0=ROB_K)
        assert rc == rcc == want and st["n_retries"] == stc["n_retries"] == 4
        assert st["final_abstol"] == pytest.approx(1e-14) and len(t) == 1          # only the initial point survives
    # without adaptive_tols: one attempt
    t, u, rc, st = hr.solve(kp(40.0, chunks=False, save=4.0, dtmin=1.0, adaptive_tols=0), [1.0, 0.0, 0.0], k0=ROB_K)
    assert rc == 2 and st["n_retries"] == 0
    # a save interval that does not divide the chunk: the chunk end is not a save point except on the last chunk
    t, u, rc, st = hr.solve(kp(8.0, chunk=4.0, save=1.5), [1.0, 0.0, 0.0], k0=ROB_K)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 8.0), solve_chunkstep=4.0, save_interval=1.5), [1.0, 0.0, 0.0], k0=ROB_K)
    assert rc == 0
    np.testing.assert_allclose(t, tc, rtol=0, atol=1e-15)
    assert units(u, uc) < 5
    hr.close()


def test_a_species_deep_below_zero_ends_the_segment_as_unstable():
    """2A -> B with A(0) = -1e-3 runs to -infinity in finite time (1 / (2 k |A0|) = 0.5 ms): the negative excursion of DESIGN 4 in
    two species. An ACCEPTED step that leaves a species below -1e3 error weights ends the segment as Unstable in every
    implementation (kinetica_jl_amd/csrc/solver_kernels.hpp BDF_NEG_DEEP) instead of following the blow-up down to dtmin: same
    retcode and the same handful of steps from the resident controller's CPU replay, oracle/cpu_bdf.cpp and oracle/bdf.py; with
    `adaptive_tols` the retry zeroes the negative entry of the chunk's start state and the solve ends with Success."""
    from oracle import bdf as obdf
    net = from_lists(2, [[(0, 2)]], [[(1, 1)]])
    k = np.array([1e6]); u0 = np.array([-1e-3, 1.0])
    hr = HostResident(net)
    cs = cpu_bdf.CpuSolver(net)
    on = orc.OracleNetwork.from_flat(net)
    t, u, rc, st = hr.solve(kp(1e-3, adaptive_tols=0), u0, k0=k)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 1e-3), adaptive_tols=False), u0, k0=k)
    to, uo, rco, sto = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), 2,
                                                 dict(tspan=(0.0, 1e-3), adaptive_tols=False), u0, k0=k)
    assert rc == rcc == rco == 3                                   # Unstable
    assert st["n_steps"] == stc["n_steps"] == sto["n_steps"] == 0  # the first step is never accepted
    t, u, rc, st = hr.solve(kp(1e-3), u0, k0=k)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 1e-3)), u0, k0=k)
    assert rc == 0 and rcc == 0 and st["n_retries"] == stc["n_retries"] == 1 and u[-1, 0] == 0.0 and uc[-1, 0] == 0.0
    hr.close()


def test_ban_negatives_rejects_steps_like_the_cpu_implementation():
    net, Ea, A = synthetic_crn(300, 1500)
    k = orc.arrhenius(Ea, A, 1400.0, k_max=1e12)
    u0 = np.zeros(300); u0[0] = 1.0
    hr = HostResident(net)
    cs = cpu_bdf.CpuSolver(net)
    t, u, rc, st = hr.solve(kp(2e-3, ban_negatives=1), u0, k0=k)
    tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 2e-3), ban_negatives=True), u0, k0=k)
    assert rc == 0 and rcc == 0 and u.min() >= 0.0
    assert abs(st["n_steps"] - stc["n_steps"]) <= 0.05 * stc["n_steps"] + 5
    assert units(u, uc) < 100
    hr.close()


def test_warm_continuation_across_chunk_starts():
    """kin_params.solve_chunks = 2 (extension): history, order and step size are carried across chunk starts whose rate
    constants did not change; rate updates still re-initialise. Same save times, fewer steps, and - the chunk boundaries of a
    static solve are not events - the result stays with the exact solution: compared with a 100x tighter chunkwise run."""
    net, Ea, A = synthetic_crn(200, 1000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    hr = HostResident(net)
    u0 = np.zeros(200); u0[0] = 1.0
    tc, uc, rcc, stc = hr.solve(kp(1e-2), u0, k0=k)
    tw, uw, rcw, stw = hr.solve(kp(1e-2, solve_chunks=2), u0, k0=k)
    tt, ut, rct, _ = hr.solve(kp(1e-2, abstol=1e-12, reltol=1e-10, dtmin=1e-30), u0, k0=k)
    assert rcc == 0 and rcw == 0 and rct == 0 and np.array_equal(tc, tw) and len(tw) == 11
    assert stw["n_chunks"] == 10 and stw["n_restarts"] == 10          # a segment start each, one of them cold
    assert stw["n_steps"] < 0.9 * stc["n_steps"] and stw["n_factor"] < stc["n_factor"]   # (0.85 before the re-initialisations got CVODE's first-step rules: they cost less now)
    assert units(uw, ut) <= 100 and units(uw, ut) <= units(uc, ut) + 20
    # with rate updates inside the span the integrator re-initialises at each of them: warm mode changes nothing there
    hr.set_arrhenius(Ea, A, k_max=1e12)
    tst = np.arange(10) * 1e-3
    Ts = 900.0 + 2e4 * tst
    t1, u1, rc1, st1 = hr.solve(kp(1e-2), u0, tstops=tst, T_stops=Ts)
    t2, u2, rc2, st2 = hr.solve(kp(1e-2, solve_chunks=2), u0, tstops=tst, T_stops=Ts)
    assert rc1 == 0 and rc2 == 0 and np.array_equal(u1, u2) and st1["n_steps"] == st2["n_steps"]
