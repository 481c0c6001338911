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
    v = (np.abs(u - ref) / (abstol + reltol * np.abs(ref))).max()
    if os.environ.get("KIN_PRINT_ERRSCALE"):       # pytest -s: the measured deviations behind the thresholds below
        import inspect
        fr = inspect.stack()[1]
        print(f"errscale {fr.function}:{fr.lineno} = {v:.3f}", flush=True)
    return v


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


def test_restart_rules_take_the_same_steps_in_every_implementation(monkeypatch):
    """The (re)initialisation rules - CVODE's initial step (cvHin) rounded down to a power of ten by exact IEEE operations, growth
    cap 1e4 at the first step-size selection and 10 afterwards - are implemented four times: resident kernel /
    lockstep controller (resident_core.hpp), host-driven integrator (solver.cpp), oracle/bdf.py, oracle/cpu_bdf.cpp. Same
    problem, same step counts."""
    from oracle import cpu_bdf
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    # ... and so is the corrector tolerance, which depends on rtol (bdf_newton_frac: 0.03 at 1e-8, 0.1 from 1e-9 down)
    for atol, rtol in ((1e-10, 1e-8), (1e-11, 1e-9), (1e-12, 1e-10)):
        pars = kp((0.0, 1.0), True, 0.1, 0.05, abstol=atol, reltol=rtol)
        odict = dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.1, save_interval=0.05, abstol=atol, reltol=rtol)
        counts = []
        for resident in ("1", "0"):
            monkeypatch.setenv("KIN_RESIDENT", resident)
            h = capi.HipNetwork.from_flat(net); h.set_rates([3.0])
            t, u, rc, st, status = h.solve(pars, [1.0, 0.0])
            assert status == capi.KIN_OK and rc == 0 and errscale(u[:, 0], np.exp(-3.0 * t)) < 100
            counts.append(st["n_steps"])
            h.close()
        to, uo, rco, sto = oracle_solve(net, odict, [1.0, 0.0], k0=np.array([3.0]))
        tc, uc, rcc, stc = cpu_bdf.CpuSolver(net).solve(odict, np.array([1.0, 0.0]), k0=np.array([3.0]))
        assert rco == 0 and rcc == 0 and counts[0] == counts[1] == sto["n_steps"] == stc["n_steps"], (rtol, counts, sto["n_steps"], stc["n_steps"])


def test_a_species_deep_below_zero_ends_the_segment_as_unstable(monkeypatch):
    """2A -> B with A(0) = -1e-3 (blow-up at 0.5 ms): an accepted step that leaves a species below -1e3 error weights ends the
    segment as Unstable (solver_kernels.hpp BDF_NEG_DEEP) - resident kernel and host-driven integrator, same retcode and step
    count as the CPU implementations (tests/test_resident_replay.py has those); with `adaptive_tols` the chunk's retry zeroes the
    negative entry and the solve ends with Success."""
    net = from_lists(2, [[(0, 2)]], [[(1, 1)]])
    for resident in ("1", "0"):
        monkeypatch.setenv("KIN_RESIDENT", resident)
        h = capi.HipNetwork.from_flat(net); h.set_rates([1e6])
        t, u, rc, st, status = h.solve(kp((0.0, 1e-3), adaptive=False), [-1e-3, 1.0])
        # (a solve that ends without Success is the reference's ErrorException("ODE solution failed."): KIN_ERR_SOLVE_FAILED)
        assert status == capi.KIN_ERR_SOLVE_FAILED and capi.RETCODE_NAMES[rc] == "Unstable" and st["n_steps"] == 0, (resident, status, rc, st)
        t, u, rc, st, status = h.solve(kp((0.0, 1e-3)), [-1e-3, 1.0])
        assert status == capi.KIN_OK and rc == 0 and st["n_retries"] == 1 and u[-1, 0] == 0.0, (resident, rc, st)
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


@pytest.mark.parametrize("n,r,chunks", [(1000, 5000, 4), (10000, 50000, 2)])
def test_corrector_update_fused_into_the_solve_and_separate(n, r, chunks, monkeypatch):
    """The corrector update folded into the solve's last gather launch (stagec_newton_kernel; the default for small
    networks, forced here with KIN_FUSE_NEWTON=1 also at 10k species where the launch has 65 workgroups and a long row
    path) against the update as a launch of its own (KIN_FUSE_NEWTON=0): the same algorithm up to summation order - within
    the solver's tolerance of each other, each deterministic run to run."""
    net, Ea, A = synthetic_crn(n, r)
    u0 = np.zeros(n); u0[0] = 1.0
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("KIN_FUSE_NEWTON", mode)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        h.rates_at(1000.0)
        a = h.solve(kp((0.0, 1e-3 * chunks), True, 1e-3), u0)
        b = h.solve(kp((0.0, 1e-3 * chunks), True, 1e-3), u0)
        assert a[2] == 0 and np.array_equal(a[1], b[1]), "not reproducible run to run"
        out[mode] = a
        h.close()
    (t1, u1, _, st1, _), (t0, u0_, _, st0, _) = out["1"], out["0"]
    assert np.array_equal(t1, t0)
    # (two step sequences that part at the first rounding difference: each is within the solver's tolerance of the true
    # solution, i.e. tens of tolerance units - measured 13.5-51 at 1k species, maximum over the species)
    assert errscale(u1, u0_) < 100
    assert abs(st1["n_steps"] - st0["n_steps"]) <= max(3, 0.03 * st0["n_steps"])


def test_speculative_enqueue_of_the_next_step_changes_nothing(monkeypatch):
    """The next step's predictor and first corrector batch are enqueued behind the current step's batch, before the host
    knows how it ended (solver.cpp: enqueue_speculative; KIN_SPECULATE=0 switches it off): same arithmetic on the device,
    so every mode of the driver must return bit-identical results and counters - chunkwise on a save grid, every step
    saved, one integration over the whole span, rate updates at tstops, negative states banned, manual stepping.
    (The host-driven integrator is forced: networks of this size are otherwise integrated by the resident kernel, which has
    no host in its step chain to speculate for.)"""
    monkeypatch.setenv("KIN_RESIDENT", "0")
    net, Ea, A = synthetic_crn(300, 1500)
    u0 = np.zeros(300); u0[0] = 1.0
    tst = np.arange(0, 9) * 1e-3
    Tst = 900.0 + 2.0e4 * tst

    def run_all():
        out = []
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        h.rates_at(1000.0)
        whole = kp((0.0, 3e-3), False, 1e-3, 5e-4)
        whole.dtmin = 1e-30          # (the first step after a restart is below the reference's eps(tspan[end]))
        for pars in (kp((0.0, 4e-3), True, 1e-3, 2.5e-4), kp((0.0, 2e-3), True, 1e-3), whole, kp((0.0, 2e-3), True, 1e-3, 5e-4, ban=True)):
            t, u, rc, st, _ = h.solve(pars, u0)
            out.append((t, u, rc, st["n_steps"], st["n_rejected"], st["n_factor"], st["n_linsolve"], st["n_newton_fail"]))
        t, u, rc, st, _ = h.solve(kp((0.0, 8e-3), True, 2e-3, 1e-3), u0, tstops=tst, T_stops=Tst)
        out.append((t, u, rc, st["n_steps"], st["n_rejected"], st["n_factor"], st["n_linsolve"], st["n_restarts"]))
        h.rates_at(1000.0)
        manual = kp((0.0, 2e-3), False)
        manual.dtmin = 1e-30
        h.integrator_init(manual, u0)
        seen = []
        for n in (1, 3, 50, 0):
            h.integrator_step(n)
            ti, ui, rci, sti = h.integrator_state()
            seen.append((ti, ui.copy(), rci, sti["n_steps"]))
        out.append(seen)
        h.close()
        return out

    monkeypatch.setenv("KIN_SPECULATE", "0")
    ref = run_all()
    monkeypatch.setenv("KIN_SPECULATE", "1")
    got = run_all()
    for a, b in zip(ref[:-1], got[:-1]):
        assert a[2] == 0 and b[2] == 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[3:] == b[3:]
    for (ta, ua, ra, na), (tb, ub, rb, nb) in zip(ref[-1], got[-1]):
        assert ta == tb and ra == rb and na == nb and np.array_equal(ua, ub)


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


def test_maxiters_exit_with_every_step_saved_leaves_no_pending_accept():
    """saveat = [] (save_interval = nothing): every accepted step is saved, and the copy of a step's state into the
    solution buffer rides in the NEXT step's predictor launch (deferred accept). A MaxIters exit has no next step: the last
    saved row must still be written, and nothing of the failed call may survive into the next solve on the same handle
    (whose solution buffer is a new allocation)."""
    net = from_lists(3, [[(0, 1)], [(1, 2)], [(1, 1), (2, 1)]], [[(1, 1)], [(1, 1), (2, 1)], [(0, 1), (2, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([0.04, 3e7, 1e4])
    u0 = [1.0, 0.0, 0.0]
    t, u, rc, st, status = h.solve(kp((0.0, 40.0), chunks=False, maxiters=40, adaptive=False), u0)
    assert status == capi.KIN_ERR_SOLVE_FAILED and rc == 1 and len(t) == st["n_steps"] + 1 and st["n_steps"] >= 10
    # every saved row is a real state of the Robertson problem: finite, positive to rounding, mass 1
    assert np.all(np.isfinite(u)) and u.min() > -1e-12
    np.testing.assert_allclose(u.sum(axis=1), 1.0, rtol=1e-9)
    # the last row continues the trajectory (not stale memory): y1 keeps decreasing monotonically in this phase
    assert np.all(np.diff(u[:, 0]) < 0) and np.all(np.diff(t) > 0)
    # the same handle again, now to the end: identical to a fresh handle's result
    good = h.solve(kp((0.0, 40.0), chunks=False, save=4.0), u0)
    h2 = capi.HipNetwork.from_flat(net)
    h2.set_rates([0.04, 3e7, 1e4])
    ref = h2.solve(kp((0.0, 40.0), chunks=False, save=4.0), u0)
    assert good[2] == 0 and np.array_equal(good[0], ref[0]) and np.array_equal(good[1], ref[1])
    # and the every-step rows of the failed run are the first rows of a run that is allowed to continue
    full = h2.solve(kp((0.0, 40.0), chunks=False), u0)
    assert np.array_equal(full[1][:len(t)], u) and np.array_equal(full[0][:len(t)], t)
    h.close(); h2.close()


# ---- end-to-end through the mirrored reference interface (solve_network) ---------------------------------
def c1_network():
    """C1: hand-written methane-pyrolysis-style CRN, 15 reactions + their reverses in the block
    order duplicate_reverse produces (cde.jl:299-309: all forwards, then all reverses), sized to
    the reference's 30-entry examples/getting_started/arrhenius_params.bson. The getting-started
    CRN itself is generated by CDE at doc-build time and is not in the reference repository."""
    names = ["C", "[CH3]", "[H]", "[H][H]", "CC", "C[CH2]", "C=C", "C=[CH]", "C#C"]
    CH4, CH3, H, H2, C2H6, C2H5, C2H4, C2H3, C2H2 = range(1, 10)
    fwd = [([CH4], [CH3, H]), ([CH4, H], [CH3, H2]), ([CH3, CH3], [C2H6]), ([C2H6], [C2H5, H]),
           ([C2H6, H], [C2H5, H2]), ([C2H6, CH3], [C2H5, CH4]), ([C2H5], [C2H4, H]), ([C2H4, H], [C2H3, H2]),
           ([C2H4, CH3], [C2H3, CH4]), ([C2H3], [C2H2, H]), ([C2H5, H], [C2H4, H2]), ([H, H], [H2]),
           ([C2H3, H], [C2H2, H2]), ([C2H5, CH3], [C2H4, CH4]), ([C2H4], [C2H2, H2])]
    allr = fwd + [(p, r) for r, p in fwd]

    def side(sp):
        ids = sorted(set(sp))
        return ids, [sp.count(i) for i in ids]
    from kinetica_jl_amd.solving import RxData, SpeciesData
    ir, sr, ip, sp_ = [], [], [], []
    for r, p in allr:
        a, b = side(r); c, d = side(p)
        ir.append(a); sr.append(b); ip.append(c); sp_.append(d)
    return SpeciesData.from_names(names), RxData(len(allr), ir, ip, sr, sp_)


def c1_parameters(d):
    """Assigns the reference's 30 (Ea, A) pairs to the hand network in a chemically sensible way:
    the 8 barrierless entries go to the radical recombinations / disproportionations, the highest
    barriers to the bond fissions, the rest in ascending order to H-abstractions and their
    reverses. (An arbitrary assignment puts forward AND reverse of most reactions at the k_max
    cap, i.e. O(1) radical concentrations exchanging at 1e12 /s: f then has an absolute round-off
    floor of ~1e-4 and every Newton-based stiff integrator - SciPy's BDF included - crawls.)"""
    order = np.argsort(np.asarray(d["Ea"]), kind="stable")
    Ea = np.asarray(d["Ea"])[order]; A = np.asarray(d["A"])[order]
    # reaction indices (0-based; forwards 0..14, reverses 15..29)
    barrierless = [15 + 0, 2, 15 + 3, 15 + 6, 15 + 9, 11, 10, 12]          # X + H -> XH, CH3 + CH3, H + H, disproportionation
    fissions = [0, 3, 15 + 2, 15 + 11, 6, 9, 14, 15 + 14]                   # strongest bonds last in `order`
    rest = [r for r in range(30) if r not in barrierless and r not in fissions]
    slot = {r: i for i, r in enumerate(barrierless + rest + fissions)}
    Ea_r = np.array([Ea[slot[r]] for r in range(30)]); A_r = np.array([A[slot[r]] for r in range(30)])
    return Ea_r, A_r


def rd_to_flat(sd, rd):
    n, rp, ri, rs, pp, pi, ps = rd.flat(sd.n)
    from kinetica_jl_amd.synth import FlatNetwork
    return FlatNetwork(n, rd.nr, rp, ri - 1, rs, pp, pi - 1, ps)


def test_c1_getting_started_static_and_variable(golden_dir):
    import json
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    d = json.load(open(os.path.join(golden_dir, "arrhenius_params.json")))
    sd, rd = c1_network()
    Ea, A = c1_parameters(d)
    # --- StaticODESolve, tspan (0, 1) s, chunkwise. 800 K rather than 1000 K: with the
    # reference's rate formula (x N_A) methane fission itself reaches 1e11 /s at 1000 K, i.e. the
    # unphysical regime described in c1_parameters.
    calc = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e12)
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0}, solve_chunkstep=0.25, save_interval=0.125)
    res = S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 800.0}), calc), sd, rd)
    assert res.sol.retcode == "Success" and len(res.sol.t) == 9 and res.sol_k is None
    assert res.rd.nr <= 30 and rd.nr == 30                      # copy_network: the caller's rd is untouched ...
    assert len(calc.Ea) == res.rd.nr                            # ... but its calculator is spliced (solve_utils.jl:241)
    # oracle on the reduced network with the reduced parameters
    k = orc.arrhenius(calc.Ea, calc.A, 800.0, k_max=1e12)
    to, uo, rco, sto = oracle_solve(rd_to_flat(res.sd, res.rd), dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.25,
                                                                        save_interval=0.125), res.sol.u[0], k0=k)
    assert errscale(res.sol.u, uo) < 10
    # carbon and hydrogen atoms are conserved by every reaction of the hand network
    nC = np.array([1, 1, 0, 0, 2, 2, 2, 2, 2]); nH = np.array([4, 3, 1, 2, 6, 5, 4, 3, 2])
    np.testing.assert_allclose(res.sol.u @ nC, 1.0, rtol=1e-6)
    np.testing.assert_allclose(res.sol.u @ nH, 4.0, rtol=1e-6)
    # --- VariableODESolve: the getting-started ramp (50 K/s from 500 K, docs/src/getting-started.md:43-49),
    # stopped at 800 K; discrete updates every 0.5 s
    sd, rd = c1_network()
    calc = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e12)
    cs = C.ConditionSet({"T": C.LinearGradientProfile(rate=50.0, X_start=500.0, X_end=800.0)}, ts_update=0.5)
    pars = S.ODESimulationParams(tspan=(0.0, 6.0), u0={"C": 1.0}, solve_chunkstep=1.0, save_interval=0.5)
    res = S.solve_network(S.VariableODESolve(pars, cs, calc), sd, rd)
    assert res.sol.retcode == "Success" and len(res.sol.t) == 13 and res.sol.t[-1] == 6.0
    assert res.sol_k.u.shape == (13, res.rd.nr) and res.rd.nr <= 30
    np.testing.assert_allclose(res.sol_k.u, orc.rate_table(calc.Ea, calc.A, 500.0 + 50.0 * res.sol_k.t, k_max=1e12), rtol=2e-15)
    ks = np.asarray(res.sol_k.u)
    to, uo, rco, sto = oracle_solve(rd_to_flat(res.sd, res.rd), dict(tspan=(0.0, 6.0), solve_chunks=True, solve_chunkstep=1.0,
                                                                        save_interval=0.5), res.sol.u[0], tstops=res.sol_k.t, ks=ks)
    assert rco == 0
    np.testing.assert_allclose(res.sol.t, to)
    assert errscale(res.sol.u, uo) < 10
    np.testing.assert_allclose(res.sol.u @ nC, 1.0, rtol=1e-6)
    # linear interpolation functor res.sol(t)
    np.testing.assert_allclose(res.sol([0.25])[0], 0.5 * (res.sol.u[0] + res.sol.u[1]))


def test_solve_network_static_low_k_cutoff_and_filter():
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    net, Ea, A = synthetic_crn(40, 160, seed=5)
    sd = S.SpeciesData.from_names([f"S{i}" for i in range(40)])
    rd = S.RxData.from_flat(net)
    calc = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e3)
    # at 300 K with tspan end 1 s the :auto cutoff (1e-8) removes the high-barrier reactions
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"S0": 1.0}, solve_chunkstep=0.5)
    k_all = orc.arrhenius(Ea, A, 300.0, k_max=1e3)
    keep = orc.low_k_keep_mask(k_all, orc.low_k_cutoff_value("auto", 1e-8, 1.0), 2.0)
    res = S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 300.0}), calc), sd, rd)
    assert res.rd.nr == int(keep.sum()) < 160 and res.sol.retcode == "Success"
    to, uo, rco, sto = oracle_solve(net.subset(np.nonzero(keep)[0]), dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.5),
                                    res.sol.u[0], k0=k_all[keep])
    assert errscale(res.sol.u, uo) < 10
    # a filter that removes every bimolecular reaction
    calc2 = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e3)
    flt = S.RxFilter([lambda sd_, rd_: [sum(s) == 2 for s in rd_.stoic_reacs]])
    # removing a reaction must also remove its calculator entries: the reference leaves that to the
    # caller (setup_network! length check throws, calculator.jl:200-204)
    with pytest.raises(ValueError):
        S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 300.0}), calc2, flt), sd, rd)


def test_edge_semantics_of_the_driver():
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    h = capi.HipNetwork.from_flat(net)
    # (1) a tstop exactly on a chunk boundary switches k at the start of the next chunk (documented rule,
    #     SURVEY A9): k = 1 on [0, 0.5), 4 on [0.5, 1]; chunk = 0.5
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), True, 0.5, 0.25), [1.0, 0.0], tstops=np.array([0.0, 0.5]),
                                   k_table=np.array([[1.0], [4.0]]))
    np.testing.assert_allclose(t, [0, 0.25, 0.5, 0.75, 1.0])
    assert errscale(u[:, 0], np.array([1, np.exp(-0.25), np.exp(-0.5), np.exp(-1.5), np.exp(-2.5)])) < 100
    assert st["n_restarts"] == 2 and st["n_chunks"] == 2
    # (2) stops after the end of the span are ignored; a single stop at t = 0 equals the static solve
    t2, u2, *_ = h.solve(kp((0.0, 1.0), True, 0.5, 0.25), [1.0, 0.0], tstops=np.array([0.0, 5.0]), k_table=np.array([[1.0], [9.0]]))
    assert errscale(u2[:, 0], np.exp(-t2)) < 100
    # (3) save_interval that does not divide the chunk: collect(0:0.2:0.5) = [0, 0.2, 0.4] (methods.jl:756-758)
    h.set_rates([1.0])
    t3, u3, *_ = h.solve(kp((0.0, 1.0), True, 0.5, 0.2), [1.0, 0.0])
    # the reference drops each chunk's LAST local save point (0.4) except on the final chunk (methods.jl:829-846):
    # global grid = chunk 0: 0, 0.2 | chunk 1: 0.5, 0.7 | final point 0.9 = the last local save of the last chunk
    np.testing.assert_allclose(t3, [0.0, 0.2, 0.5, 0.7, 0.9], rtol=0, atol=1e-15)
    assert len(t3) == (3 - 1) * 2 + 1                                             # (len(saveat_local)-1)*n_chunks+1 (methods.jl:761)
    assert errscale(u3[:, 0], np.exp(-t3)) < 100
    # (4) non-increasing tstops are rejected
    with pytest.raises(capi.KineticaHipError) as e:
        h.solve(kp((0.0, 1.0), True, 0.5), [1.0, 0.0], tstops=np.array([0.0, 0.0]), k_table=np.array([[1.0], [1.0]]))
    assert e.value.code == capi.KIN_ERR_INVALID_ARG
    h.close()
    # (5) update_tols writes the tightened tolerances back into pars (solve_utils.jl:397-401); Dummy calculator
    #     through solve_network exercises the host k-table path of a VariableODESolve
    sd = S.SpeciesData.from_names(["A", "B"])
    rd = S.RxData(1, [[1]], [[2]], [[1]], [[1]])
    calc = S.DummyKineticCalculator([2.0])
    cs = C.ConditionSet({"T": C.LinearDirectProfile(rate=100.0, X_start=300.0, X_end=400.0)}, ts_update=0.5)
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0, 0.0], solve_chunkstep=0.5, save_interval=0.5, low_k_cutoff="none")
    res = S.solve_network(S.VariableODESolve(pars, cs, calc), sd, rd)
    assert res.sol_k.u.shape == (3, 1) and np.all(res.sol_k.u == 2.0)
    assert errscale(res.sol.u[:, 0], np.exp(-2.0 * res.sol.t)) < 100
    assert (pars.abstol, pars.reltol) == (1e-10, 1e-8)


def test_ban_negatives_rejects_negative_states():
    # A + B -> C with B exhausted exactly: tiny negative excursions of B appear at loose tolerances;
    # with ban_negatives = true (isoutofdomain, methods.jl:169-171) no saved state is negative
    net = from_lists(3, [[(0, 1), (1, 1)]], [[(2, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([1e6])
    p = kp((0.0, 1.0), chunks=False, save=0.01, abstol=1e-6, reltol=1e-3, ban=True)
    t, u, rc, st, status = h.solve(p, [1.0, 0.5, 0.0])
    assert rc == 0 and u.min() >= 0.0
    np.testing.assert_allclose(u[-1], [0.5, 0.0, 0.5], atol=1e-4)
    h.close()


def test_continuous_rate_updates_n3():
    """Continuous-rate VariableODESolve (methods.jl:363-653): k(t) = Arrhenius(T(t)) re-evaluated every step.
    A -> B: A(t) = exp(-int_0^t k(T(s)) ds), checked against quadrature and against the oracle."""
    from scipy.integrate import quad
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    Ea, A = np.array([8.0e4]), np.array([1.0e-17])         # k(T) = A exp(-Ea/RT) N_A : 0.03 /s at 500 K, 6 /s at 700 K
    sd = S.SpeciesData.from_names(["A", "B"])
    rd = S.RxData(1, [[1]], [[2]], [[1]], [[1]])
    calc = S.PrecalculatedArrheniusCalculator(Ea, A)
    cs = C.ConditionSet({"T": C.LinearGradientProfile(rate=100.0, X_start=500.0, X_end=700.0)})   # no ts_update -> continuous
    pars = S.ODESimulationParams(tspan=(0.0, 2.0), u0=[1.0, 0.0], solve_chunkstep=0.5, save_interval=0.25, low_k_cutoff="none")
    res = S.solve_network(S.VariableODESolve(pars, cs, calc), sd, rd)
    assert res.sol.retcode == "Success" and res.sol_k is None and len(res.sol.t) == 9
    kfun = lambda t: float(orc.arrhenius(Ea, A, 500.0 + 100.0 * t)[0])
    truth = np.array([np.exp(-quad(kfun, 0.0, tt, epsabs=1e-13, epsrel=1e-13)[0]) for tt in res.sol.t])
    assert errscale(res.sol.u[:, 0], truth) < 100
    np.testing.assert_allclose(res.sol_vcs["T"], 500.0 + 100.0 * res.sol.t, rtol=1e-12)
    assert res.sol.stats["n_restarts"] == 4                 # one per chunk, none inside
    # oracle with the same k(t)
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    on = orc.OracleNetwork.from_flat(net)
    to, uo, rco, sto = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)), lambda kk: (lambda y: on.jac(kk, y)), 2,
                                                 dict(tspan=(0.0, 2.0), solve_chunks=True, solve_chunkstep=0.5, save_interval=0.25),
                                                 [1.0, 0.0], k_of_time=lambda tg: orc.arrhenius(Ea, A, 500.0 + 100.0 * tg))
    assert errscale(res.sol.u, uo) < 10
    # a static T in a VariableODESolve-style continuous set behaves like the static solve
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A)
    t, u, rc, st, _ = h.solve_continuous(kp((0.0, 1.0), True, 0.5, 0.25), [1.0, 0.0], [0.0, 1.0], [600.0, 600.0])
    assert errscale(u[:, 0], np.exp(-kfun(1.0) * t)) < 100   # kfun(1.0) = k(600 K)
    with pytest.raises(capi.KineticaHipError):
        h.solve_continuous(kp((0.0, 1.0), True, 0.5), [1.0, 0.0], [0.0], [600.0])
    h.close()


def test_return_integrator_n1(golden_dir, monkeypatch):
    """return_integrator=true (methods.jl:175-178, 242-246, 706-709): the initialised integrator is
    stepped by the caller. Same kernels and step logic as the host-driven kin_solve, so manual stepping reproduces
    that kin_solve's every-step output exactly (a network of this size would otherwise be integrated by the resident
    kernel - same algorithm, results within the tolerance, tests/test_gpu_resident.py - and manual stepping is a
    host-driven feature)."""
    monkeypatch.setenv("KIN_RESIDENT", "0")
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    u0 = np.zeros(60); u0[0] = 1.0
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e3)
    k = orc.arrhenius(Ea, A, 900.0, k_max=1e3)
    h.set_rates(k)
    # (a) static, complete timespan: saveat = [] makes kin_solve store every accepted step
    t, u, rc, st, _ = h.solve(kp((0.0, 1.0), chunks=False), u0)
    h.integrator_init(kp((0.0, 1.0), chunks=False), u0)
    t0, uu, rc0, _ = h.integrator_state()
    assert t0 == 0.0 and rc0 == 0 and np.array_equal(uu, u0)
    ts = []
    while h.integrator_step(1) == 1:
        ts.append(h.integrator_state(with_u=False)[0])
    assert np.array_equal(np.array(ts), t[1:])
    tf, uf, rcf, stf = h.integrator_state()
    assert tf == 1.0 and rcf == 0 and np.array_equal(uf, u[-1]) and stf["n_steps"] == st["n_steps"]
    assert h.integrator_step(5) == 0                      # already at the end of the span
    # several steps per call, then solve!(integ)
    h.integrator_init(kp((0.0, 1.0), chunks=False), u0)
    assert h.integrator_step(7) == 7
    assert h.integrator_state(with_u=False)[0] == t[7]
    h.integrator_step(0)
    assert np.array_equal(h.integrator_state()[1], u[-1])
    # (b) discrete rate updates, complete timespan: the tstops fire during manual stepping
    z = np.load(os.path.join(golden_dir, "truth_small.npz"))
    tst, T = np.arange(8) * 0.125, z["ramp_T"]
    t3, u3, rc3, st3, _ = h.solve(kp((0.0, 1.0), False), u0, tstops=tst, T_stops=T)
    h.integrator_init(kp((0.0, 1.0), False), u0, tstops=tst, T_stops=T)
    n = h.integrator_step(0)
    tf, uf, rcf, stf = h.integrator_state()
    assert n == st3["n_steps"] and tf == 1.0 and np.array_equal(uf, u3[-1]) and stf["n_restarts"] == 8
    assert errscale(uf, z["ramp_u"][-1]) < 100
    # (c) chunkwise: the integrator spans the first chunk only
    t4, u4, rc4, st4, _ = h.solve(kp((0.0, 1.0), True, 0.25), u0, tstops=tst, T_stops=T)
    h.integrator_init(kp((0.0, 1.0), True, 0.25), u0, tstops=tst, T_stops=T)
    h.integrator_step(0)
    tf, uf, _, _ = h.integrator_state()
    assert tf == 0.25 and np.array_equal(uf, u4[1])
    # errors: stepping without init on a fresh handle, invalid span
    h2 = capi.HipNetwork.from_flat(net)
    with pytest.raises(capi.KineticaHipError):
        h2.integrator_step(1)
    h2.set_rates(k)
    with pytest.raises(capi.KineticaHipError):
        h2.integrator_init(kp((1.0, 0.0), chunks=False), u0)
    h2.close(); h.close()
    # (d) host interface: solve_network(...; return_integrator=true)
    sd = S.SpeciesData.from_names([f"S{i}" for i in range(60)])
    rd = S.RxData.from_flat(net)
    calc = S.PrecalculatedArrheniusCalculator(Ea, A, k_max=1e3)
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"S0": 1.0}, solve_chunks=False, low_k_cutoff="none")
    with S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 900.0}), calc), sd, rd, return_integrator=True) as integ:
        assert isinstance(integ, S.HipIntegrator) and integ.t == 0.0
        # (the calculator's rates differ from orc.arrhenius in the last bits, so no bitwise comparison here)
        assert integ.step(3) == 3 and abs(integ.t - t[3]) < 1e-6 * t[3]
        integ.solve()
        assert integ.t == 1.0 and integ.retcode == "Success" and errscale(integ.u, u[-1]) < 1


def test_explicit_solver_matches_scipy_rk45():
    """kin_solve_explicit (Dormand-Prince 5(4) on the RHS kernels; BASELINE config 2: "RHS kernel only, explicit
    solver") against SciPy's RK45 behind the same driver: same step sequence, same dense output."""
    from kinetica_jl_amd.synth import narrow_k_variant
    # closed form, chunkwise with interpolated saves
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])
    h = capi.HipNetwork.from_flat(net)
    h.set_rates([3.0])
    t, u, rc, st, status = h.solve(kp((0.0, 1.0), True, 0.25, 0.125), [1.0, 0.0], explicit=True)
    assert status == capi.KIN_OK and rc == 0 and st["n_factor"] == 0 and st["n_jac"] == 0
    np.testing.assert_allclose(t, np.arange(9) * 0.125, atol=1e-15)
    assert errscale(u[:, 0], np.exp(-3.0 * t)) < 100
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 1.0), solve_chunks=True, solve_chunkstep=0.25, save_interval=0.125,
                                              explicit=True), [1.0, 0.0], k0=np.array([3.0]))
    assert st["n_steps"] == sto["n_steps"] and errscale(u, uo) < 1e-3
    h.close()
    # C2-like: synthetic CRN, narrow-k variant (non-stiff), static 1000 K, every accepted step stored
    net, Ea, A = synthetic_crn(200, 1000, seed=6)
    k = orc.arrhenius(narrow_k_variant(Ea), A, 1000.0, k_max=1e3)
    u0 = np.zeros(200); u0[0] = 1.0
    h = capi.HipNetwork.from_flat(net)
    h.set_rates(k)
    t, u, rc, st, status = h.solve(kp((0.0, 0.05), chunks=False, abstol=1e-8, reltol=1e-6), u0, explicit=True)
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 0.05), solve_chunks=False, abstol=1e-8, reltol=1e-6, explicit=True),
                                    u0, k0=k)
    assert rc == 0 and rco == 0
    assert st["n_steps"] == sto["n_steps"] and len(t) == len(to)
    np.testing.assert_allclose(t, to, rtol=1e-9)
    assert errscale(u, uo, 1e-8, 1e-6) < 1e-2
    np.testing.assert_allclose((u * net.mass).sum(axis=1), net.mass[0], rtol=1e-9)     # mass invariant
    # against the implicit path on the same problem (two different integrators, same tolerance)
    tb, ub, rcb, stb, _ = h.solve(kp((0.0, 0.05), chunks=False, save=0.01, abstol=1e-8, reltol=1e-6), u0)
    te, ue, rce, ste, _ = h.solve(kp((0.0, 0.05), chunks=False, save=0.01, abstol=1e-8, reltol=1e-6), u0, explicit=True)
    np.testing.assert_allclose(tb, te)
    assert errscale(ue, ub, 1e-8, 1e-6) < 100
    # discrete rate updates under the explicit integrator
    tst = np.arange(5) * 0.01
    ks = np.stack([k * (1.0 + 0.1 * i) for i in range(5)])
    t, u, rc, st, _ = h.solve(kp((0.0, 0.05), True, 0.025, 0.005, abstol=1e-8, reltol=1e-6), u0, tstops=tst, k_table=ks, explicit=True)
    to, uo, rco, sto = oracle_solve(net, dict(tspan=(0.0, 0.05), solve_chunks=True, solve_chunkstep=0.025, save_interval=0.005,
                                              abstol=1e-8, reltol=1e-6, explicit=True), u0, tstops=tst, ks=ks)
    assert rc == 0 and np.allclose(t, to) and st["n_restarts"] == sto["n_restarts"]
    assert errscale(u, uo, 1e-8, 1e-6) < 1e-2
    h.close()


def test_inert_collision_partner_n4():
    """insert_inert! then solve: A + M -> B + M with M inert is first-order decay with rate k [M]; M itself must
    stay exactly constant. The species-on-both-sides records take the kernels' explicit-operand paths."""
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    sd = S.SpeciesData.from_names(["A", "B"])
    rd = S.RxData(2, [[1], [2]], [[2], [1]], [[1], [1]], [[1], [1]])
    S.insert_inert(rd, sd, ["Ar"])
    calc = S.DummyKineticCalculator([2.0, 0.5])
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"A": 1.0, "Ar": 3.0}, solve_chunks=False, save_interval=0.125,
                                 low_k_cutoff="none")
    res = S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 300.0}), calc), sd, rd)
    t, u = np.asarray(res.sol.t), np.asarray(res.sol.u)
    assert res.sol.retcode == "Success" and np.all(u[:, 2] == 3.0)
    kf, kr = 2.0 * 3.0, 0.5 * 3.0
    a = kr / (kf + kr) + (1.0 - kr / (kf + kr)) * np.exp(-(kf + kr) * t)
    assert errscale(u[:, 0], a) < 100
    # the seeds of the next exploration level come from the device-side maxima (explore_utils.jl:338-374)
    assert res.sol.umax is not None and np.array_equal(res.sol.umax, u.max(axis=0))
    assert S.identify_next_seeds(res.sol, sd, 0.5) == ["A", "B", "Ar"]
    assert S.identify_next_seeds(res.sol, sd, 0.5, ignore=["Ar"]) == ["A", "B"]
    assert S.identify_next_seeds(res.sol, sd, 0.9) == ["A", "Ar"]
    # the batched sweep on the same network
    h = capi.HipNetwork(*rd.flat(sd.n), index_base=1)
    h.set_rates([2.0, 0.5])
    U = np.array([[1.0, 0.0, 3.0], [0.3, 0.7, 2.0], [0.0, 1.0, 0.5]])
    got = h.rhs_batched(U)
    ref = np.stack([[-2.0 * x[0] * x[2] + 0.5 * x[1] * x[2], 2.0 * x[0] * x[2] - 0.5 * x[1] * x[2], 0.0] for x in U])
    np.testing.assert_allclose(got, ref, rtol=1e-14, atol=1e-300)
    h.close()


def test_solve_network_with_explicit_solver_selection():
    """`pars.solver = "RK45"` routes solve_network through kin_solve_explicit; same trajectories as the default
    BDF within the solver tolerance; an unknown solver name is refused before anything runs."""
    from kinetica_jl_amd import conditions as C
    from kinetica_jl_amd import solving as S
    from kinetica_jl_amd.synth import narrow_k_variant
    net, Ea, A = synthetic_crn(60, 300, seed=11)
    sd = S.SpeciesData.from_names([f"S{i}" for i in range(60)])
    out = {}
    for name in (None, "RK45"):
        rd = S.RxData.from_flat(net)
        calc = S.PrecalculatedArrheniusCalculator(narrow_k_variant(Ea), A, k_max=1e3)
        pars = S.ODESimulationParams(tspan=(0.0, 0.1), u0={"S0": 1.0}, solver=name, solve_chunkstep=0.05, save_interval=0.025,
                                     abstol=1e-9, reltol=1e-7, low_k_cutoff="none")
        out[name] = S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 900.0}), calc), sd, rd)
        assert out[name].sol.retcode == "Success"
    assert out["RK45"].sol.stats["n_factor"] == 0 and out[None].sol.stats["n_factor"] > 0
    np.testing.assert_allclose(out["RK45"].sol.t, out[None].sol.t)
    assert errscale(np.asarray(out["RK45"].sol.u), np.asarray(out[None].sol.u), 1e-9, 1e-7) < 100
    pars = S.ODESimulationParams(tspan=(0.0, 0.1), u0={"S0": 1.0}, solver="Rodas5")
    with pytest.raises(ValueError):
        S.solve_network(S.StaticODESolve(pars, C.ConditionSet({"T": 900.0}), S.DummyKineticCalculator(np.ones(300))), sd,
                        S.RxData.from_flat(net))
