"""CPU tests: the oracle against every piece of known-answer material the reference holds for
the solve path; runs without a GPU."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from kinetica_jl_amd.synth import from_lists, synthetic_crn


@pytest.fixture(scope="module")
def arr(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "arrhenius_params.json")))
    return np.array(d["Ea"]), np.array(d["A"])


def test_arrhenius_file_shape(arr):
    Ea, A = arr
    assert len(Ea) == 30 and len(A) == 30
    assert (Ea == 0).sum() == 8 and Ea.max() == pytest.approx(595362.9, rel=1e-6)
    assert 5.9e8 < A.min() < 6.0e8 and 2.0e12 < A.max() < 2.1e12


def test_arrhenius_known_ranges(arr):
    # ranges recorded in SURVEY.md 8(c) for k_max = 1e12 (calculator.jl:223-226)
    Ea, A = arr
    k500 = orc.arrhenius(Ea, A, 500.0, k_max=1e12)
    k1200 = orc.arrhenius(Ea, A, 1200.0, k_max=1e12)
    assert k500.min() == pytest.approx(4.66e-30, rel=5e-3) and k500.max() == pytest.approx(1e12, rel=1e-9)
    assert k1200.min() == pytest.approx(8.9e6, rel=1e-2) and k1200.max() == pytest.approx(1e12, rel=1e-9)
    # :auto cutoff for a 14 s span at reltol 1e-8 (solve_utils.jl:221-236)
    cutoff = orc.low_k_cutoff_value("auto", 1e-8, 14.0)
    assert (~orc.low_k_keep_mask(k500, cutoff, 2.0)).sum() == 8
    assert (~orc.low_k_keep_mask(k1200, cutoff, 2.0)).sum() == 0


def test_arrhenius_formula_independent(arr):
    # independent numpy evaluation of the same expression, with and without k_max
    Ea, A = arr
    for T in (300.0, 500.0, 1000.0):
        kr = A * np.exp(-Ea / (8.314462618 * T)) * 6.02214076e23 * 1.0
        np.testing.assert_allclose(orc.arrhenius(Ea, A, T), kr, rtol=4e-16)
        np.testing.assert_allclose(orc.arrhenius(Ea, A, T, k_max=1e12), 1.0 / ((1.0 / 1e12) + (1.0 / kr)), rtol=4e-16)
    # t_mult: rates per ms are 1e-3 of rates per s before the cap (calculator.jl:196, 224)
    np.testing.assert_allclose(orc.arrhenius(Ea, A, 800.0, t_mult=orc.tconvert("ms", "s")),
                               orc.arrhenius(Ea, A, 800.0) * 1e-3, rtol=1e-15)


def test_dummy_calculator_order_of_cap_and_tmult():
    # calculator.jl:130-132 applies t_mult AFTER the cap; :144-146 without cap
    r = np.array([1e3, 1e13])
    np.testing.assert_allclose(orc.dummy_rates(r, k_max=1e12, t_mult=1e-3), 1.0 / ((1.0 / 1e12) + (1.0 / r)) * 1e-3)
    np.testing.assert_allclose(orc.dummy_rates(r, t_mult=60.0), r * 60.0)


def test_doc_crn_rhs_matches_written_odes(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "doc_crn.json")))
    net = from_lists(5, d["reacs"], d["prods"])
    on = orc.OracleNetwork.from_flat(net)
    rng = np.random.default_rng(0)
    for _ in range(5):
        k = rng.uniform(0.1, 3.0, 6)
        u = rng.uniform(0.0, 2.0, 5)
        du = on.rhs(k, u)
        for i, name in enumerate(d["species"]):
            want = sum(sgn * k[ki] * np.prod(u[fac]) for sgn, ki, fac in d["odes"][name])
            assert du[i] == pytest.approx(want, rel=1e-14, abs=1e-15)


def test_no_combinatoric_factor_and_stoichiometry():
    # 2A -> B : rate = k A^2 (no 1/2!), dA = -2 k A^2, dB = + k A^2   (combinatoric_ratelaws=false)
    net = from_lists(2, [[(0, 2)]], [[(1, 1)]])
    on = orc.OracleNetwork.from_flat(net)
    du = on.rhs([3.0], [0.5, 0.0])
    np.testing.assert_allclose(du, [-2 * 3.0 * 0.25, 3.0 * 0.25])
    # inert collider A + M -> B + M (solve_utils.jl:178-183): M multiplies the rate, net 0
    net = from_lists(3, [[(0, 1), (2, 1)]], [[(1, 1), (2, 1)]])
    du = orc.OracleNetwork.from_flat(net).rhs([2.0], [0.5, 0.0, 4.0])
    np.testing.assert_allclose(du, [-4.0, 4.0, 0.0])


def test_oracle_jacobian_matches_finite_differences():
    net, Ea, A = synthetic_crn(60, 300, seed=7)
    on = orc.OracleNetwork.from_flat(net)
    rng = np.random.default_rng(1)
    k = rng.uniform(0.5, 2.0, net.n_reactions)
    u = rng.uniform(0.1, 1.0, 60)
    J = on.jac(k, u).toarray()
    for j in range(0, 60, 7):
        h = 1e-6 * u[j]
        up, um = u.copy(), u.copy()
        up[j] += h; um[j] -= h
        fd = (on.rhs(k, up) - on.rhs(k, um)) / (2 * h)
        np.testing.assert_allclose(J[:, j], fd, rtol=1e-6, atol=1e-6)


def test_conservation_of_atoms_like_invariant():
    # A -> B + C ; B -> D : total "mass" with weights w(A)=2,w(B)=1,w(C)=1,w(D)=1 is conserved
    net = from_lists(4, [[(0, 1)], [(1, 1)]], [[(1, 1), (2, 1)], [(3, 1)]])
    du = orc.OracleNetwork.from_flat(net).rhs([1.3, 0.7], [0.9, 0.4, 0.1, 0.0])
    assert abs(np.dot(du, [2, 1, 1, 1])) < 1e-15


def test_condition_profile_known_answers(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "conditions_kat.json")))
    p = orc.linear_direct(**kat["lineardirect"]["args"])
    assert p["t_end"] == pytest.approx(kat["lineardirect"]["t_end"])
    for t, v in kat["lineardirect"]["f_at"]:
        assert orc.profile_f(p, t) == pytest.approx(v)
    np.testing.assert_allclose(p["tstops"], kat["lineardirect"]["tstops"])
    p = orc.null_direct(**kat["nulldirect"]["args"])
    assert orc.profile_f(p, 5.0) == pytest.approx(300.0) and list(p["tstops"]) == [10.0]
    p = orc.null_gradient(**kat["nullgradient"]["args"])
    assert orc.profile_grad(p, 5.0) == 0.0 and list(p["tstops"]) == [10.0]
    p = orc.linear_gradient(**kat["lineargradient"]["args"])
    assert p["t_end"] == pytest.approx(4.0)
    for t, v in kat["lineargradient"]["grad_at"]:
        assert orc.profile_grad(p, t) == v
    p = orc.double_ramp_gradient(**kat["doubleramp"]["args"])
    assert p["t_end"] == pytest.approx(48.0) and p["t_blend"] == 0.0
    np.testing.assert_allclose(p["tstops"], kat["doubleramp"]["tstops"])
    for t, v in kat["doubleramp"]["grad_at"]:
        assert orc.profile_grad(p, t) == v
    p = orc.double_ramp_gradient(**kat["doubleramp_blended"]["args"])
    np.testing.assert_allclose(p["tstops"], kat["doubleramp_blended"]["tstops"])


def test_gradient_profile_solution_is_the_ramp():
    # getting-started ramp: 500 -> 1200 K at 50 K/s => t_end = 14 s (docs/src/getting-started.md:43-49)
    p = orc.linear_gradient(rate=50.0, X_start=500.0, X_end=1200.0)
    assert p["t_end"] == pytest.approx(14.0)
    orc.create_discrete_tstops(p, 1e-3)
    assert len(p["tstops"]) == 14001 and p["tstops"][0] == 0.0 and p["tstops"][-1] == 14.0
    orc.solve_variable_condition(p, (0.0, 14.0), None)
    T = orc.interp_linear(p["sol_t"], p["sol_u"], p["tstops"])
    np.testing.assert_allclose(T, 500.0 + 50.0 * p["tstops"], rtol=1e-12)
    lo, hi = orc.profile_minmax(p)
    assert lo == pytest.approx(500.0) and hi == pytest.approx(1200.0)
    # double ramp integrates to its plateaus
    p = orc.double_ramp_gradient(300.0, 5.0, 10.0, 500.0, 3.0, -20.0, 200.0, 5.0)
    orc.solve_variable_condition(p, (0.0, 48.0), None)
    assert orc.interp_linear(p["sol_t"], p["sol_u"], [26.0])[0] == pytest.approx(500.0)
    assert orc.interp_linear(p["sol_t"], p["sol_u"], [48.0])[0] == pytest.approx(200.0)
    assert orc.interp_linear(p["sol_t"], p["sol_u"], [15.0])[0] == pytest.approx(400.0)


def test_julia_range_and_savepoints():
    r = orc.julia_range(0.0, 0.001, 14.0)
    assert len(r) == 14001 and r[7] == 0.007 and r[-1] == 14.0 and r[1234] == 1.234
    r = orc.julia_range(0.0, 0.005, 0.01)
    assert list(r) == [0.0, 0.005, 0.01]
    s = orc.create_savepoints(0.0, 4.0, 1.5)
    assert list(s) == [0.0, 1.5, 3.0, 4.0]          # end point appended (utils.jl:111-113)
    n, grid, size_final = orc.chunk_grids(14.0, 1e-2, 5e-3)
    assert (n, len(grid), size_final) == (1400, 3, 2801)
    with pytest.raises(ValueError):
        orc.chunk_grids(1.0, 0.3, None)


def test_make_u0():
    u = orc.make_u0(4, {"C": 1.0}, {"A": 0, "C": 2})
    assert list(u) == [0, 0, 1.0, 0]
    assert list(orc.make_u0(3, [1.0], allow_short_u0=True)) == [1.0, 0, 0]
    with pytest.raises(ValueError):
        orc.make_u0(3, [1.0])
    with pytest.raises(KeyError):
        orc.make_u0(3, {"X": 1.0}, {"A": 0})


def test_synthetic_crn_is_seeded_and_valid():
    n1, Ea1, A1 = synthetic_crn(200, 1000)
    n2, Ea2, A2 = synthetic_crn(200, 1000)
    assert np.array_equal(n1.reac_idx, n2.reac_idx) and np.array_equal(Ea1, Ea2) and np.array_equal(A1, A2)
    assert n1.n_reactions == 1000
    # forward/reverse pairs
    for r in range(0, 1000, 2):
        assert n1.reaction(r)[0] == n1.reaction(r + 1)[1] and n1.reaction(r)[1] == n1.reaction(r + 1)[0]
    # molecularity <= 2 per side, every species appears
    assert np.add.reduceat(n1.reac_sto, n1.reac_ptr[:-1]).max() <= 2
    assert np.add.reduceat(n1.prod_sto, n1.prod_ptr[:-1]).max() <= 2
    assert len(np.unique(np.concatenate([n1.reac_idx, n1.prod_idx]))) == 200
    assert (Ea1 == 0).mean() == pytest.approx(0.25, abs=0.05) and Ea1.max() < 6.5e5
    assert 10 ** 8.8 <= A1.min() and A1.max() <= 10 ** 12.3
    # mass conservation: every reaction balances, so sum_i m_i du_i = 0 for any state and any k
    on = orc.OracleNetwork.from_flat(n1)
    rng = np.random.default_rng(3)
    k = rng.uniform(0.1, 10.0, 1000); u = rng.uniform(0.0, 1.0, 200)
    assert abs(np.dot(n1.mass, on.rhs(k, u))) <= 1e-12 * np.dot(n1.mass, on.abs_rhs(k, u))
    # detailed balance before the cap: k_f / k_r = exp(-dG / RT) with a common prefactor
    assert np.array_equal(A1[0::2], A1[1::2]) and np.all(np.minimum(Ea1[0::2], Ea1[1::2]) >= 0)
    assert np.all((Ea1[0::2] == 0) | (Ea1[1::2] == 0) | (np.minimum(Ea1[0::2], Ea1[1::2]) > 0))


def test_ensemble_rhs_equals_single_state_rhs():
    """orc_rhs_many (OpenMP over states, bench.py's all-core CPU baseline) = orc_rhs state by state."""
    from kinetica_jl_amd.synth import synthetic_crn
    net, Ea, A = synthetic_crn(200, 1000, seed=2)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 900.0, k_max=1e12)
    rng = np.random.default_rng(3)
    U = 10.0 ** rng.uniform(-12, 0, (17, 200))
    K = k[None, :] * rng.uniform(0.5, 2.0, (17, 1))
    for kk, D in ((k, on.rhs_many(k, U)), (K, on.rhs_many(K, U))):
        for b in range(17):
            assert np.array_equal(D[b], on.rhs(kk if kk.ndim == 1 else kk[b], U[b]))


def test_config_truths_are_cross_checked_by_an_independent_integrator(golden_dir):
    """Every truth file comes from oracle/cpu_bdf.cpp (the device integrator's own algorithm family at 100-1000x tighter
    tolerances). tests/golden/make_truth_independent.py integrated the same problems with SciPy's Radau IIA - nothing shared
    but the right-hand side - and stored how far it lands from the committed truths, in units of the DEFAULT tolerances: a
    semantic error common to device and mirror (restart rule, zero-order hold of the rate constants, chunk stitching) would
    show there. What was checked per file (`independent_points`): truth_c3 2 chunk ends; truth_c4 the first chunk (10 rate
    intervals); truth_c3_mid 5 and 10 ms, one integration from t = 0; truth_c3_long all ten stored chunk ends, ONE Radau integration from t = 0 over the 100 ms (4.3 h of one core): 4.4 / 8.5 / 7.6 / 3.2 / 1.1 / 3.7 / 10.0 / 18.3 / 30.5 / 60.5 units at 10 ... 100 ms, rms 0.59 - at 100 ms single species of this truth are known to ~60 units, the size of its own self_check (40.7), which is what the `major` bounds of tests/test_gpu_configs.py leave room for; truth_c4_long
    the first two chunk ends (20 rate intervals); truth_c5 the first rate update (1 ms - a pair of SuperLU factorisations takes
    40 s at 50k species)."""
    import os
    bounds = {"c3": (2.0, 0.2), "c4": (2.0, 0.2), "c3_mid": (5.0, 0.5), "c3_long": (65.0, 0.7), "c4_long": (5.0, 0.5)}
    z5 = np.load(os.path.join(golden_dir, "truth_c5.npz"))
    if "self_check_independent" in z5:          # (the Radau run over the first rate interval at 50k species: hours; stored when it is there)
        bounds["c5"] = (5.0, 0.5)
    else:                                       # until then: the regenerated file's own check (x1e-3 stored, x1e-2 against it; make_truth_configs.py c5)
        assert float(z5["self_check"]) < 10.0 and "u_early" in z5
    for name, (mx, rms) in bounds.items():
        z = np.load(os.path.join(golden_dir, f"truth_{name}.npz"))
        assert "self_check_independent" in z, name
        assert float(z["self_check_independent"]) < mx, (name, float(z["self_check_independent"]))
        if rms is not None:
            assert float(z["self_check_independent_rms"]) < rms, name
        assert "Radau" in str(z["independent_method"])
