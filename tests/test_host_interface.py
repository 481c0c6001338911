"""CPU tests of the host-side mirror of the reference interface (kinetica_jl_amd.conditions /
.solving): the reference's own condition tests (test/Main/conditions.jl) restated against this
package, constructor validation (params.jl:77-104, methods.jl:12-20, 49-57), filters,
calculators, u0, cutoff bookkeeping. Nothing here touches the GPU or the oracle."""
import json
import os

import numpy as np
import pytest

from kinetica_jl_amd import conditions as C
from kinetica_jl_amd import solving as S


def test_profile_construction_reference_values(golden_dir):
    # test/Main/conditions.jl:4-90
    kat = json.load(open(os.path.join(golden_dir, "conditions_kat.json")))
    assert C.StaticConditionProfile(10.0).value == 10.0
    nd = C.NullDirectProfile(X_start=300.0, t_end=10.0)
    assert nd.X_start == 300.0 and nd.t_end == 10.0 and nd.f(5.0, nd) == pytest.approx(300.0)
    assert len(nd.tstops) == 1 and nd.tstops[0] == pytest.approx(10.0)
    ld = C.LinearDirectProfile(rate=50.0, X_start=300.0, X_end=500.0)
    assert (ld.rate, ld.X_start, ld.X_end) == (50.0, 300.0, 500.0)
    assert ld.t_end == pytest.approx(4.0) and ld.f(2.0, ld) == pytest.approx(400.0)
    assert len(ld.tstops) == 1 and ld.tstops[0] == pytest.approx(4.0)
    ng = C.NullGradientProfile(X_start=300.0, t_end=10.0)
    assert ng.grad(5.0, ng) == 0.0 and ng.tstops[0] == pytest.approx(10.0)
    lg = C.LinearGradientProfile(rate=50.0, X_start=300.0, X_end=500.0)
    assert lg.t_end == pytest.approx(4.0) and lg.grad(2.0, lg) == 50.0 and lg.grad(5.0, lg) == 0.0
    dr = C.DoubleRampGradientProfile(**kat["doubleramp"]["args"])
    assert dr.t_blend == 0.0 and dr.t_end == pytest.approx(48.0)
    np.testing.assert_allclose(dr.tstops, [5.0, 25.0, 28.0, 43.0, 48.0])
    for t, v in kat["doubleramp"]["grad_at"]:
        assert dr.grad(t, dr) == v
    db = C.DoubleRampGradientProfile(**kat["doubleramp_blended"]["args"])
    assert db.t_blend == 0.1
    np.testing.assert_allclose(db.tstops, [4.9, 5.1, 24.9, 25.1, 27.9, 28.1, 42.9, 43.1, 48.0])
    with pytest.raises(RuntimeError):
        C.LinearDirectProfile(rate=-50.0, X_start=300.0, X_end=500.0)


def _cs(ts_update=None):
    return C.ConditionSet({
        "T": C.LinearDirectProfile(rate=50.0, X_start=300.0, X_end=500.0),
        "P": C.DoubleRampGradientProfile(X_start=1e5, t_start_plateau=1.0, rate1=1e3, X_mid=2e5, t_mid_plateau=10.0,
                                         rate2=-1e3, X_end=1e5, t_end_plateau=1.0, t_blend=0.1),
        "V": 1e3}, ts_update=ts_update)


def test_condition_set_construction_reference_values():
    # test/Main/conditions.jl:92-135
    csc = _cs()
    assert set(csc.symbols) == {"T", "P", "V"} and len(csc.profiles) == 3
    assert csc.discrete_updates is False and csc.ts_update is None
    csd = _cs(1e-3)
    assert csd.discrete_updates is True and csd.ts_update == pytest.approx(1e-3)
    with pytest.raises(ValueError):
        C.ConditionSet({"X": "abc"})
    assert C.isstatic(csc, "V") and C.isvariable(csc, "T") and not C.isstatic(csc) and not C.isvariable(csc)
    ts = C.get_tstops(csd)
    assert ts[0] == 0.0 and np.all(np.diff(ts) > 0) and ts[-1] == pytest.approx(C.get_t_final(csd))
    with pytest.raises(RuntimeError):
        C.get_tstops(C.ConditionSet({"T": 300.0}))


def test_discrete_tstops_and_profile_solutions():
    p = C.LinearGradientProfile(rate=50.0, X_start=500.0, X_end=1200.0)
    cs = C.ConditionSet({"T": p}, ts_update=1e-3)
    ts = C.get_tstops(cs)
    assert len(ts) == 14001 and ts[1234] == 1.234 and ts[-1] == 14.0
    pars = S.ODESimulationParams(tspan=(0.0, 14.0), u0={"C": 1.0}, solve_chunkstep=1e-2, save_interval=5e-3)
    C.solve_variable_conditions(cs, pars)
    np.testing.assert_allclose(p.sol(ts), 500.0 + 50.0 * ts, rtol=1e-12)
    assert p.minimum() == pytest.approx(500.0) and p.maximum() == pytest.approx(1200.0)
    ld = C.LinearDirectProfile(rate=50.0, X_start=300.0, X_end=500.0)
    cs2 = C.ConditionSet({"T": ld}, ts_update=0.5)
    np.testing.assert_allclose(ld.tstops, np.arange(9) * 0.5)
    pars2 = S.ODESimulationParams(tspan=(0.0, 5.0), u0=[1.0], solve_chunks=False)
    C.solve_variable_conditions(cs2, pars2)
    assert ld.sol([2.0])[0] == pytest.approx(400.0) and ld.sol([4.5])[0] == pytest.approx(500.0)
    with pytest.raises(ValueError):
        C.ConditionSet({"T": C.LinearDirectProfile(rate=50.0, X_start=300.0, X_end=500.0)}, ts_update=10.0)


def test_params_validation():
    ok = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0})
    assert (ok.abstol, ok.reltol, ok.solve_chunks, ok.solve_chunkstep, ok.maxiters) == (1e-10, 1e-8, True, 1e-3, 100000)
    assert ok.low_k_cutoff == "auto" and ok.low_k_maxconc == 2.0 and ok.save_interval is None
    for bad in (dict(tspan=(1.0, 1.0)), dict(tspan=(0.0, 1.0), low_k_cutoff="sometimes"), dict(tspan=(0.0, 1.0), low_k_cutoff=-1.0),
                dict(tspan=(0.0, 1.0), solve_chunkstep=0.3), dict(tspan=(0.0, 1.0), solve_chunkstep=0.1, save_interval=0.2)):
        with pytest.raises(ValueError):
            S.ODESimulationParams(u0=[1.0], **bad)
    S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0], solve_chunks=False, solve_chunkstep=0.3)   # only checked when chunking
    kp = ok.to_kin_params()
    assert kp.save_interval == -1.0 and kp.solve_chunks == 1 and kp.maxiters == 100000


def test_solver_selection():
    """`pars.solver` (params.jl:9: any SciML algorithm in the reference): None / "BDF" -> the implicit path,
    "RK45" / "DP5" / "explicit" -> kin_solve_explicit; anything else is refused, never silently replaced."""
    assert S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0}).explicit is False
    assert S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0}, solver="BDF").explicit is False
    for name in ("RK45", "dp5", "explicit"):
        assert S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0}, solver=name).explicit is True
    with pytest.raises(ValueError):
        S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0}, solver="Rodas5").explicit


def test_solver_sentinels_carry_dtmin():
    """`pars.solver = HIPBDF(dtmin=...)`: the option lives on the solver object (the reference's ODESimulationParams has no
    dtmin field, params.jl:3-27; it hands eps(solve_chunkstep) / eps(tspan[end]) to `init`, methods.jl:164, 232, 694, 770)."""
    base = dict(tspan=(0.0, 1.0), u0={"A": 1.0}, solve_chunkstep=0.5)
    p = S.ODESimulationParams(solver=S.HIPBDF(), **base)
    assert not p.explicit and p.solver_dtmin is None and p.to_kin_params().dtmin == 0.0        # 0 = the reference's value
    p = S.ODESimulationParams(solver=S.HIPBDF(dtmin=1e-30), **base)
    assert not p.explicit and p.to_kin_params().dtmin == 1e-30
    p = S.ODESimulationParams(solver=S.HIPRK45(dtmin=1e-12), **base)
    assert p.explicit and p.to_kin_params().dtmin == 1e-12
    p = S.ODESimulationParams(solver=S.HIPBDF(), dtmin=1e-20, **base)                           # the older extension field still works
    assert p.to_kin_params().dtmin == 1e-20
    p = S.ODESimulationParams(solver=S.HIPBDF(dtmin=1e-30), dtmin=1e-20, **base)               # the sentinel wins
    assert p.to_kin_params().dtmin == 1e-30


def test_warm_chunk_starts_travel_as_solve_chunks_2():
    """`HIPBDF(warm_chunks=True)` (extension): chunkwise solve without re-initialisation at chunk starts whose rates did not
    change = kin_params.solve_chunks 2; plain chunkwise 1, complete timespan 0 whatever the option says."""
    base = dict(tspan=(0.0, 1.0), u0={"A": 1.0}, solve_chunkstep=0.5)
    assert S.ODESimulationParams(solver=S.HIPBDF(), **base).to_kin_params().solve_chunks == 1
    assert S.ODESimulationParams(solver=S.HIPBDF(warm_chunks=True), **base).to_kin_params().solve_chunks == 2
    assert S.ODESimulationParams(solver=S.HIPBDF(warm_chunks=True), solve_chunks=False, **base).to_kin_params().solve_chunks == 0
    assert S.ODESimulationParams(solver=None, **base).to_kin_params().solve_chunks == 1
    assert S.ODESimulationParams(solver=S.HIPRK45(), **base).to_kin_params().solve_chunks == 1


def test_discrete_stop_temperatures_is_the_interpolation_half_of_calculate_discrete_rates():
    cs = C.ConditionSet({"T": C.LinearGradientProfile(rate=50.0, X_start=500.0, X_end=510.0)}, ts_update=0.05)
    pars = S.ODESimulationParams(tspan=(0.0, 0.2), u0={"A": 1.0}, solve_chunkstep=0.1)
    C.solve_variable_conditions(cs, pars)
    tst, T = S.discrete_stop_temperatures(cs)
    np.testing.assert_allclose(tst, [0.0, 0.05, 0.1, 0.15, 0.2], atol=1e-15)
    np.testing.assert_allclose(T, 500.0 + 50.0 * tst, rtol=1e-9)
    static = C.ConditionSet({"T": 900.0}, ts_update=None)
    with pytest.raises(RuntimeError):
        S.discrete_stop_temperatures(static)                      # continuous / static sets have no discrete stops


def test_solve_method_constructors():
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0])
    calc = S.PrecalculatedArrheniusCalculator([1.0], [1.0])
    S.StaticODESolve(pars, C.ConditionSet({"T": 300.0}), calc)
    with pytest.raises(ValueError):   # variable condition in a static solve (methods.jl:13-14)
        S.StaticODESolve(pars, C.ConditionSet({"T": C.NullDirectProfile(X_start=300.0, t_end=1.0)}), calc)
    with pytest.raises(ValueError):   # Arrhenius knows only T (calculator.jl:234-236)
        S.StaticODESolve(pars, C.ConditionSet({"T": 300.0, "V": 1.0}), calc)
    S.StaticODESolve(pars, C.ConditionSet({"T": 300.0, "V": 1.0}), S.DummyKineticCalculator([1.0]))
    assert S.allows_continuous(calc) and S.has_conditions(calc, ["T"])
    with pytest.raises(RuntimeError):
        S.PrecalculatedLindemannCalculator([1.0], [1.0], [1.0])(T=300.0)


def test_filters_network_and_u0():
    sd = S.SpeciesData.from_names(["A", "B", "C"])
    rd = S.RxData(3, [[1], [2], [1, 2]], [[2], [3], [3]], [[1], [1], [1, 1]], [[1], [1], [1]])
    assert not get_mask(S.RxFilter(), sd, rd).any()
    bimol = lambda sd_, rd_: [len(r) == 2 for r in rd_.id_reacs]
    assert list(get_mask(S.RxFilter([bimol]), sd, rd)) == [False, False, True]
    assert list(get_mask(S.RxFilter([bimol], keep_filtered=True), sd, rd)) == [True, True, False]
    rd.splice([2])
    assert rd.nr == 2 and rd.id_reacs == [[1], [2]]
    n, rp, ri, rs, pp, pi, ps = rd.flat(3)
    assert list(rp) == [0, 1, 2] and list(ri) == [1, 2] and list(pi) == [2, 3]
    pars = S.ODESimulationParams(tspan=(0.0, 1.0), u0={"C": 1.0})
    assert list(S.make_u0(sd, pars)) == [0.0, 0.0, 1.0]
    with pytest.raises(RuntimeError):
        S.make_u0(sd, S.ODESimulationParams(tspan=(0.0, 1.0), u0={"X": 1.0}))
    with pytest.raises(RuntimeError):
        S.make_u0(sd, S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0]))
    assert list(S.make_u0(sd, S.ODESimulationParams(tspan=(0.0, 1.0), u0=[1.0], allow_short_u0=True))) == [1.0, 0.0, 0.0]


def get_mask(rf, sd, rd):
    return S.get_filter_mask(rf, sd, rd)


def test_dummy_calculator_and_low_k_cutoff_bookkeeping():
    calc = S.DummyKineticCalculator([1e3, 1e-12, 5.0, 1e-11], t_unit="ms")
    np.testing.assert_allclose(calc(T=300.0), np.array([1e3, 1e-12, 5.0, 1e-11]) * 1e-3)
    capped = S.DummyKineticCalculator([1e3, 1e13], k_max=1e12)
    np.testing.assert_allclose(capped(T=1.0), 1.0 / (1e-12 + 1.0 / np.array([1e3, 1e13])))
    sd = S.SpeciesData.from_names(["A", "B"])
    rd = S.RxData(4, [[1], [2], [1], [2]], [[2], [1], [2], [1]], [[1]] * 4, [[1]] * 4)
    pars = S.ODESimulationParams(tspan=(0.0, 10.0), u0=[1.0, 0.0])          # :auto -> cutoff = 1e-8 / 10
    cs = C.ConditionSet({"T": 300.0})
    calc = S.DummyKineticCalculator([1e3, 1e-12, 5.0, 2e-10])
    S.setup_network(sd, rd, calc)
    removed = S.apply_low_k_cutoff(rd, calc, pars, cs)      # removed iff k * 2^2 < 1e-9
    assert removed == 2 and rd.nr == 2 and list(calc.rates) == [1e3, 5.0]
    with pytest.raises(ValueError):
        S.setup_network(sd, rd, S.DummyKineticCalculator([1.0]))
    pars_none = S.ODESimulationParams(tspan=(0.0, 10.0), u0=[1.0, 0.0], low_k_cutoff="none")
    assert S.apply_low_k_cutoff(rd, calc, pars_none, cs) == 0
    assert S.tconvert("ms", "s") == 1e-3 and S.tconvert(2.0, "mins", "s") == 120.0


def test_insert_inert_topology():
    """insert_inert! (solve_utils.jl:126-192): unimolecular reactions gain the inert species on both sides; one copy
    per additional inert species; bimolecular reactions and 2A -> ... are left alone."""
    sd = S.SpeciesData.from_names(["A", "B", "C"])
    rd = S.RxData(3, [[1], [1, 2], [1]], [[2], [3], [2, 3]], [[1], [1, 1], [2]], [[1], [1], [1, 1]], dH=[0.1, 0.2, 0.3])
    S.insert_inert(rd, sd, ["N#N"])
    assert sd.n == 4 and sd.toInt["N#N"] == 4 and sd.toStr[4] == "N#N"
    assert rd.nr == 3 and rd.id_reacs == [[1, 4], [1, 2], [1]] and rd.id_prods == [[2, 4], [3], [2, 3]]
    assert rd.stoic_reacs == [[1, 1], [1, 1], [2]] and rd.stoic_prods == [[1, 1], [1], [1, 1]]
    # two inert species, one of them already in the network: a copy for the first, the original takes the last
    sd = S.SpeciesData.from_names(["A", "B", "Ar"])
    rd = S.RxData(1, [[1]], [[2]], [[1]], [[1]], dH=[0.5])
    S.insert_inert(rd, sd, ["Ar", "He"])
    assert sd.n == 4 and rd.nr == 2 and rd.dH == [0.5, 0.5]
    assert rd.id_reacs == [[1, 4], [1, 3]] and rd.id_prods == [[2, 4], [2, 3]]


def test_identify_next_seeds_selection_and_seeds_out(tmp_path):
    """identify_next_seeds (explore_utils.jl:338-406) on a host-side solution (no device maxima attached): threshold,
    `ignore`, `elim_small_na`, the method without a threshold, and the seeds.out table (explore_utils.jl:363-372:
    count, header padded to the longest SMILES, `rpad(sid, 5)`, Julia's Float64 printing)."""
    sd = S.SpeciesData.from_names(["C", "CC", "[H][H]", "C=C", "CCCC"], n_atoms=[5, 8, 2, 6, 14])
    u = np.array([[1.0, 0.0, 0.0, 0.0, 0.0],
                  [0.6, 0.2, 0.05, 1e-7, 0.004],
                  [0.3, 0.15, 0.3, 2.5e-7, 0.001]])
    sol = S.ODESolution(np.array([0.0, 1.0, 2.0]), u, "Success")
    assert S.identify_next_seeds(sol, sd, 0.1) == ["C", "CC", "[H][H]"]
    assert S.identify_next_seeds(sol, sd, 0.2) == ["C", "CC", "[H][H]"]          # >= : a maximum AT the threshold counts
    assert S.identify_next_seeds(sol, sd, 0.1, ignore=["CC"]) == ["C", "[H][H]"]
    assert S.identify_next_seeds(sol, sd, 0.1, elim_small_na=5) == ["C", "CC"]
    assert S.identify_next_seeds(sol, sd) == ["C", "CC", "[H][H]", "C=C", "CCCC"]
    assert S.identify_next_seeds(sol, sd, elim_small_na=6, ignore=["CCCC"]) == ["CC", "C=C"]
    out = tmp_path / "seeds.out"
    S.identify_next_seeds(sol, sd, 1e-7, saveto=str(out))
    assert out.read_text().splitlines() == [
        "5",
        "SID   SMILES   Max. Conc.",
        "1     C        1.0",
        "2     CC       0.2",
        "3     [H][H]   0.3",
        "4     C=C      2.5e-7",
        "5     CCCC     0.004",
    ]
    with pytest.raises(ValueError):
        S.identify_next_seeds(sol, S.SpeciesData.from_names(["a", "b", "c", "d", "e"]), 0.1, elim_small_na=3)
    with pytest.raises(ValueError):
        S.identify_next_seeds(sol, S.SpeciesData.from_names(["a", "b"]), 0.1)
    # a device-reduced maximum, when attached, is what is read (the trajectory is not scanned)
    sol2 = S.ODESolution(np.array([0.0]), u[:1], "Success", umax=np.array([0.0, 0.0, 0.0, 0.0, 7.0]))
    assert S.identify_next_seeds(sol2, sd, 0.1) == ["CCCC"]
