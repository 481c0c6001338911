"""GPU parity tests of kin_solve: trajectories of the HIP BDF path against (1) the CPU oracle
running the same algorithm on the same inputs, (2) closed forms, (3) the committed Radau truth.
Stated tolerance (north_star: "within a stated relative tolerance"):
    |u - u_truth|  <= 100 * (abstol + reltol * |u_truth|)      at every save point,
and device-vs-oracle agreement 10x tighter than that (same algorithm, different LU / summation)."""
import os

import numpy as np
import pytest

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import bdf as obdf
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def kp(tspan, chunks=True, chunkstep=1e-3, save=None, abstol=1e-10, reltol=1e-8, maxiters=100000, adaptive=True, ban=False):
    return capi.KinParams(tspan0=tspan[0], tspan1=tspan[1], abstol=abstol, reltol=reltol, adaptive_tols=int(adaptive),
                          update_tols=0, solve_chunks=int(chunks), ban_negatives=int(ban), solve_chunkstep=chunkstep,
                          maxiters=maxiters, save_interval=-1.0 if save is None else save)


def oracle_solve(net, pars, u0, k0=None, tstops=None, ks=None):
    on = orc.OracleNetwork.from_flat(net)
    return obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)),
                                     net.n_species, pars, u0, k0=k0, tstops=tstops,
                                     k_of_stop=None if ks is None else (lambda i: ks[i]))


def errscale(u, ref, abstol=1e-10, reltol=1e-8):
    return (np.abs(u - ref) / (abstol + reltol * np.abs(ref))).max()


def test_first_order_decay_chunkwise():
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([3.0])
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), True, 0.1, 0.05), [1.0, 0.0])
    assert status == capi.KIN_OK and rc == 0 and len(t) == 21
    np.testing.assert_allclose(t, np.arange(21) * 0.05, atol=1e-15)
    assert errscale(u[:, 0], np.exp(-3.0 * t)) < 100
    np.testing.assert_allclose(u.sum(axis=1), 1.0, rtol=1e-12)
    assert st["n_chunks"] == 10 and st["n_restarts"] == 10
    # same thing through the oracle: same algorithm -> near-identical numbers and step counts
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.1, save_interval=0.05),
                                    [1.0, 0.0], k0=np.array([3.0]))
    np.testing.assert_allclose(t, to)
    assert errscale(u, uo) < 10
    assert st["n_steps"] == sto["n_steps"]
    h.close()


def test_closed_forms_complete_timespan():
    pars = kp((0.0, 2.0), chunks=False, save=0.25)
    # A <-> B
    h = capi.HipNetwork.from_flat(from_lists(2, [[(0, 1)], [(1, 1)]], [[(1, 1)], [(0, 1)]]))
    h.set_rates([2.0, 0.5])
    t, u, rc, st, status = h.solve(pars, [1.0, 0.0])
    np.testing.assert_allclose(t, np.arange(9) * 0.25)
    assert errscale(u[:, 0], 0.2 + 0.8 * np.exp(-2.5 * t)) < 100
    h.close()
    # 2A -> B : A = 1 / (1 + 2 k t)
    h = capi.HipNetwork.from_flat(from_lists(2, [[(0, 2)]], [[(1, 1)]]))
    h.set_rates([1.5])
    t, u, rc, st, status = h.solve(pars, [1.0, 0.0])
    assert errscale(u[:, 0], 1.0 / (1.0 + 3.0 * t)) < 100
    h.close()
    # A + B -> C
    h = capi.HipNetwork.from_flat(from_lists(3, [[(0, 1), (1, 1)]], [[(2, 1)]]))
    h.set_rates([0.7])
    t, u, rc, st, status = h.solve(pars, [1.0, 1.0, 0.0])
    assert errscale(u[:, 0], 1.0 / (1.0 + 0.7 * t)) < 100
    h.close()


def test_every_step_output_when_no_save_interval():
    # complete-timespan solve with save_interval = nothing -> saveat = [] -> every accepted step (methods.jl:166)
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([3.0])
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), chunks=False), [1.0, 0.0])
    assert rc == 0 and t[0] == 0.0 and t[-1] == 1.0 and len(t) == st["n_steps"] + 1
    assert np.all(np.diff(t) > 0)
    assert errscale(u[:, 0], np.exp(-3.0 * t)) < 100
    h.close()


def test_robertson_and_synthetic_against_truth(golden_dir):
    z = np.load(os.path.join(golden_dir, "truth_small.npz"))
    net = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([0.04, 3e7, 1e4])
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, save=4.0), [1.0, 0.0, 0.0])
    assert rc == 0
    np.testing.assert_allclose(t, z["rober_t"])
    assert errscale(u, z["rober_u"]) < 100
    h.close()
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e3)
    h.rates_at(1000.0)
    u0 = np.zeros(60); u0[0] = 1.0
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), True, 0.125, 0.0625), u0)
    assert rc == 0
    np.testing.assert_allclose(t, z["syn_t"])
    assert errscale(u, z["syn_u"]) < 100
    umax = h.solution_max()
    np.testing.assert_array_equal(umax, u.max(axis=0))
    h.close()


def test_discrete_rate_updates_against_truth_and_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "truth_small.npz"))
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    u0 = np.zeros(60); u0[0] = 1.0
    tst = np.arange(8) * 0.125
    T = z["ramp_T"]
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e3)
    # (a) rates generated on the device from T at every stop; chunk = 0.25 so half the stops are interior
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), True, 0.25, 0.0625), u0, tstops=tst, T_stops=T)
    assert rc == 0
    np.testing.assert_allclose(t, z["ramp_t"])
    assert errscale(u, z["ramp_u"]) < 100
    assert st["n_restarts"] == 8
    # (b) the same through a host k-table (any calculator)
    ks = orc.rate_table(Ea, A, T, k_max=1e3)
    t2, u2, rc2, st2, _ = h.solve(kp((0.0, 1.0), True, 0.25, 0.0625), u0, tstops=tst, k_table=ks)
    assert errscale(u2, u) < 1
    # (c) the oracle on the same inputs
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.25, save_interval=0.0625),
                                    u0, tstops=tst, ks=ks)
    assert errscale(u, uo) < 10
    # (d) complete-timespan discrete variant (methods.jl:655-714)
    t3, u3, rc3, st3, _ = h.solve(kp((0.0, 1.0), False, save=0.0625), u0, tstops=tst, T_stops=T)
    np.testing.assert_allclose(t3, z["ramp_t"])
    assert errscale(u3, z["ramp_u"]) < 100
    h.close()


def test_c2_synthetic_matches_oracle():
    """1k species / 5k reactions, narrow-k variant, chunkwise: device vs oracle (same algorithm)."""
    from kinetica_jl_amd.synth import narrow_k_variant
    net, Ea, A = synthetic_crn(1000, 5000)
    k = orc.arrhenius(narrow_k_variant(Ea), A, 1000.0, k_max=1e4)
    u0 = np.zeros(1000); u0[0] = 1.0
    h = capi.HipNetwork.from_flat(net)
    h.set_rates(k)
    t, u, rc, st, status = h.solve(kp((0.0, 0.004), True, 1e-3, 5e-4), u0)
    assert rc == 0 and len(t) == 9
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 0.004), solve_chunks=True, solve_chunkstep=1e-3, save_interval=5e-4),
                                    u0, k0=k)
    assert rco == 0
    assert errscale(u, uo) < 10
    assert abs(st["n_steps"] - sto["n_steps"]) <= max(2, 0.02 * sto["n_steps"])
    assert st["lu_dense_dim"] + st["lu_sparse_rows"] == 1000 and st["lu_rounds"] >= 1
    # conservation: every reaction of the generator conserves nothing in general, but total
    # positivity and boundedness hold
    assert u.min() > -1e-9
    h.close()


def test_failure_and_retry_semantics():
    net = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([0.04, 3e7, 1e4])
    # maxiters too small: MaxIters -> tolerances /10 up to 5 attempts (solve_utils.jl:406-420), then failure
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, maxiters=5), [1.0, 0.0, 0.0])
    assert status == capi.KIN_ERR_SOLVE_FAILED and rc == 1 and st["n_retries"] == 4
    assert st["final_abstol"] == pytest.approx(1e-14) and st["final_reltol"] == pytest.approx(1e-12)
    # adaptive_tols = false: no retry (solve_utils.jl:403-405)
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, save=4.0, maxiters=5, adaptive=False), [1.0, 0.0, 0.0])
    assert status == capi.KIN_ERR_SOLVE_FAILED and st["n_retries"] == 0
    # invalid parameters -> KIN_ERR_INVALID_ARG (params.jl:77-104)
    for bad in (kp((1.0, 0.5)), kp((0.0, 1.0), True, 0.3), kp((0.0, 1.0), True, 0.1, 0.2)):
        with pytest.raises(capi.KineticaHipError) as e:
            h.solve(bad, [1.0, 0.0, 0.0])
        assert e.value.code == capi.KIN_ERR_INVALID_ARG
    h.close()
