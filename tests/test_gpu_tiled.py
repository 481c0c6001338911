"""GPU parity of the tiled batched sweep (library order, kinetica_jl_amd/csrc/tiled_kernels.hip) through the C ABI:
the k-stream form against the oracle's RHS, the temperature form (rate constants formed inside the sweep) against
the oracle's RHS with the oracle's Arrhenius rate constants, the library-order rate table against the plain one."""
import numpy as np
import pytest
import torch

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-13
R_GAS = 8.314462618


def _dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")


def _sync():
    """The library launches on the handle's own non-blocking stream: whatever torch has queued on ITS stream (fills,
    arithmetic on the inputs) must have finished before a library call reads or writes those buffers."""
    torch.cuda.synchronize()


def _states(B, n, seed):
    return 10.0 ** np.random.default_rng(seed).uniform(-12, 0, (B, n))


def _tiled_k(h, U, K):
    """du[b] through kin_rates_to_lib_dev / kin_states_to_lib_dev / kin_rhs_tiled_dev / kin_states_from_lib_dev."""
    B, N = U.shape
    lay = h.lib_layout()
    d_u, d_k = _dev(U), _dev(K)
    d_ul, d_kl = torch.empty_like(d_u), torch.empty((B, lay["k_len"]), dtype=torch.float64, device="cuda")
    d_dul, d_du = torch.full_like(d_u, float("nan")), torch.full_like(d_u, float("nan"))
    _sync()
    h.states_to_lib_dev(B, d_u.data_ptr(), d_ul.data_ptr())
    h.rates_to_lib_dev(B, d_k.data_ptr(), d_kl.data_ptr())
    _sync()
    h.rhs_tiled_dev(B, d_ul.data_ptr(), d_dul.data_ptr(), d_k_lib=d_kl.data_ptr())
    h.states_from_lib_dev(B, d_dul.data_ptr(), d_du.data_ptr())
    # the same through the drop-in entry: states in the caller's order, rate constants in slot order (kin_rhs_batched_klib_dev)
    d_du2 = torch.full_like(d_u, float("nan"))
    _sync()
    h.rhs_batched_klib_dev(B, d_u.data_ptr(), d_kl.data_ptr(), d_du2.data_ptr())
    torch.cuda.synchronize()
    # the conversions are permutations: checked against the layout the library reports
    assert np.array_equal(d_ul.cpu().numpy(), U[:, lay["species_of_lib"]])
    kl = np.zeros((B, lay["k_len"])); kl[:, lay["slot_of_reaction"]] = K
    assert np.array_equal(d_kl.cpu().numpy(), kl)
    lay["du_klib"] = d_du2.cpu().numpy()
    return d_du.cpu().numpy(), lay


def _check_against_oracle(net, du, U, K, rows, tol=TOL):
    on = orc.OracleNetwork.from_flat(net)
    worst = 0.0
    for b in rows:
        k = K[b] if K.ndim == 2 else K
        ref, scale = on.rhs(k, U[b]), on.abs_rhs(k, U[b])
        err = np.abs(du[b] - ref) / np.maximum(scale, 1e-300)
        worst = max(worst, err.max())
    assert worst <= tol, worst
    return worst


@pytest.mark.parametrize("n,r,B", [(300, 1500, 7), (1000, 5000, 33), (3000, 15000, 9), (6000, 30000, 5)],
                         ids=["300", "C2_1k", "3k_bs512", "6k_bs1024"])
def test_k_stream_form_matches_oracle(n, r, B):
    net, Ea, A = synthetic_crn(n, r)
    h = capi.HipNetwork.from_flat(net)
    U = _states(B, n, 1)
    K = 10.0 ** np.random.default_rng(2).uniform(-3, 3, (B, r))
    du, lay = _tiled_k(h, U, K)
    assert lay["identity"] and lay["windows"] == 1
    _check_against_oracle(net, du, U, K, range(B))
    _check_against_oracle(net, lay["du_klib"], U, K, range(B))      # kin_rhs_batched_klib_dev
    h.close()


def test_windows_on_a_small_network(monkeypatch):
    monkeypatch.setenv("KIN_TILED_ENTRIES", "1400")     # read when the handle builds its layout
    net, Ea, A = synthetic_crn(3000, 15000)
    h = capi.HipNetwork.from_flat(net)
    B = 300                                            # more states than workgroups: the state loop and its prefetches
    U = _states(B, 3000, 3)
    K = 10.0 ** np.random.default_rng(4).uniform(-3, 3, (B, 15000))
    du, lay = _tiled_k(h, U, K)
    assert lay["windows"] > 1 and not lay["identity"]
    _check_against_oracle(net, du, U, K, [0, 1, 17, 255, 256, 299])
    _check_against_oracle(net, lay["du_klib"], U, K, [0, 1, 17, 255, 256, 299])      # kin_rhs_batched_klib_dev
    # temperature form on the same layout, caller-order entry point (converts on the way in and out)
    h.set_arrhenius(Ea, A, k_max=1e12)
    T = np.linspace(500.0, 1200.0, B)
    d_u, d_T = _dev(U), _dev(T)
    d_du = torch.full_like(d_u, float("nan"))
    _sync()
    h.rhs_batched_T_dev(B, d_u.data_ptr(), d_T.data_ptr(), d_du.data_ptr())
    torch.cuda.synchronize()
    duT = d_du.cpu().numpy()
    on = orc.OracleNetwork.from_flat(net)
    for b in (0, 150, 299):
        k = orc.arrhenius(Ea, A, T[b], k_max=1e12)
        bound = (2 * (Ea.max() / (R_GAS * T[b])) + 16) * 2.0 ** -53
        assert np.all(np.abs(duT[b] - on.rhs(k, U[b])) <= (bound + TOL) * on.abs_rhs(k, U[b]) + 1e-300)
    h.close()


def test_special_stoichiometries_and_unpaired_reactions():
    reacs = [[(0, 2)], [(1, 1), (2, 1)], [(3, 1)], [(4, 2)], [(0, 1), (5, 1)], [(1, 1), (5, 1)], [(2, 1)], [(2, 1)], [(2, 1)], [(5, 2)]]
    prods = [[(1, 1), (2, 1)], [(0, 2)], [(4, 2)], [(3, 1)], [(1, 1), (5, 1)], [(0, 1), (5, 1)], [(3, 1), (4, 1)], [(0, 1)], [(0, 1)], [(5, 1), (1, 1)]]
    net = from_lists(6, reacs, prods)
    h = capi.HipNetwork.from_flat(net)
    U = _states(5, 6, 5)
    K = 10.0 ** np.random.default_rng(6).uniform(-2, 2, (5, 10))
    du, lay = _tiled_k(h, U, K)
    assert lay["records"] == 7
    _check_against_oracle(net, du, U, K, range(5))
    _check_against_oracle(net, lay["du_klib"], U, K, range(5))      # kin_rhs_batched_klib_dev
    h.close()


def test_degenerate_networks():
    """No reactions at all; one irreversible reaction; a single state; fewer states than workgroups."""
    h0 = capi.HipNetwork.from_flat(from_lists(3, [], []))
    d_u = _dev(np.ones((4, 3)))
    d_du = torch.full_like(d_u, float("nan"))
    d_k = torch.zeros((4, 2), dtype=torch.float64, device="cuda")
    assert h0.lib_layout()["k_len"] == 0
    _sync()
    h0.rhs_tiled_dev(4, d_u.data_ptr(), d_du.data_ptr(), d_k_lib=d_k.data_ptr())
    torch.cuda.synchronize()
    assert np.all(d_du.cpu().numpy() == 0.0)
    h0.close()
    net = from_lists(2, [[(0, 1)]], [[(1, 1)]])          # A -> B, no reverse
    h1 = capi.HipNetwork.from_flat(net)
    U = np.array([[2.0, 5.0]])
    K = np.array([[3.0]])
    du, lay = _tiled_k(h1, U, K)
    assert lay["k_len"] == 2 and np.array_equal(du, [[-6.0, 6.0]])
    h1.set_arrhenius(np.array([0.0]), np.array([1.0 / 6.02214076e23]))        # k = 1 at any temperature
    d_u, d_T = _dev(U), _dev(np.array([700.0]))
    d_du = torch.full_like(d_u, float("nan"))
    _sync()
    h1.rhs_batched_T_dev(1, d_u.data_ptr(), d_T.data_ptr(), d_du.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(d_du.cpu().numpy(), [[-2.0, 2.0]], rtol=1e-15)
    h1.close()


@pytest.mark.parametrize("k_max", [1e12, None], ids=["kmax", "nokmax"])
def test_temperature_form_and_library_order_table(k_max):
    """C2-size network: du from T[b] (no k anywhere) against the oracle; the library-order rate table against the plain
    table kernel (same arithmetic: equal bit for bit) and against the oracle within the table kernel's stated bound."""
    n, r, B = 1000, 5000, 40
    net, Ea, A = synthetic_crn(n, r)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=k_max)
    U = _states(B, n, 7)
    T = np.linspace(500.0, 1500.0, B)
    lay = h.lib_layout()
    d_u, d_T = _dev(U), _dev(T)
    d_du = torch.full_like(d_u, float("nan"))
    _sync()
    h.rhs_batched_T_dev(B, d_u.data_ptr(), d_T.data_ptr(), d_du.data_ptr())
    torch.cuda.synchronize()
    du = d_du.cpu().numpy()
    on = orc.OracleNetwork.from_flat(net)
    for b in range(0, B, 3):
        k = orc.arrhenius(Ea, A, T[b], k_max=k_max)
        bound = (2 * (Ea.max() / (R_GAS * T[b])) + 16) * 2.0 ** -53
        assert np.all(np.abs(du[b] - on.rhs(k, U[b])) <= (bound + TOL) * on.abs_rhs(k, U[b]) + 1e-300)
    # table in library order
    d_tl = torch.full((B, lay["k_len"]), float("nan"), dtype=torch.float64, device="cuda")
    d_t = torch.empty((B, r), dtype=torch.float64, device="cuda")
    _sync()
    h.rate_table_lib_dev(T, d_tl.data_ptr())
    _sync()
    h.rate_table_dev(T, d_t.data_ptr())
    tl, t = d_tl.cpu().numpy(), d_t.cpu().numpy()
    for b in (0, B // 2, B - 1):
        ko = orc.arrhenius(Ea, A, T[b], k_max=k_max)
        bound = (2 * np.abs(Ea / (R_GAS * T[b])) + 8) * 2.0 ** -53
        assert np.all(np.abs(tl[b][lay["slot_of_reaction"]] - ko) <= bound * ko)
    assert np.array_equal(tl[:, lay["slot_of_reaction"]], t)
    # the k-stream form fed with that table reproduces the temperature form's arithmetic up to the exp table size
    d_du2 = torch.full_like(d_u, float("nan"))
    _sync()
    h.rhs_tiled_dev(B, d_u.data_ptr(), d_du2.data_ptr(), d_k_lib=d_tl.data_ptr())
    torch.cuda.synchronize()
    du2 = d_du2.cpu().numpy()
    scale = np.stack([on.abs_rhs(orc.arrhenius(Ea, A, T[b], k_max=k_max), U[b]) for b in range(B)])
    assert np.all(np.abs(du2 - du) <= 2e-13 * scale + 1e-300)
    h.close()


def test_argument_errors():
    net, Ea, A = synthetic_crn(300, 1500)
    h = capi.HipNetwork.from_flat(net)
    d = torch.zeros((2, 300), dtype=torch.float64, device="cuda")
    with pytest.raises(capi.KineticaHipError):       # neither k nor T
        _sync()
        h.rhs_tiled_dev(2, d.data_ptr(), d.data_ptr())
    with pytest.raises(capi.KineticaHipError) as e:  # T form without Arrhenius parameters
        _sync()
        h.rhs_tiled_dev(2, d.data_ptr(), d.data_ptr(), d_T=d.data_ptr())
    assert e.value.code == capi.KIN_ERR_STATE
    # a reaction with three product molecules has no fixed-role record: the layout is refused, the plain sweep still works
    net3 = from_lists(3, [[(0, 1)], [(1, 1)]], [[(1, 3)], [(2, 1)]])
    h3 = capi.HipNetwork.from_flat(net3)
    with pytest.raises(capi.KineticaHipError) as e:
        h3.lib_layout()
    assert e.value.code == capi.KIN_ERR_UNSUPPORTED
    assert h3.rhs_batched(np.ones((1, 3)), np.ones((1, 2))).shape == (1, 3)
    h.close(); h3.close()


def test_full_size_c3_elementwise_and_c5_tiled_sweep():
    """BASELINE sizes. C3 (10k / 50k): B = 300 states (> CU count) with per-state k, element-wise against the oracle on
    sampled states. C5 (50k / 250k, 9 windows): sampled states against the oracle, linearity in k, mass conservation,
    run-to-run reproducibility of the sampled rows, and the temperature form on the same states."""
    for (n, r, B) in ((10000, 50000, 300), (50000, 250000, 264)):
        net, Ea, A = synthetic_crn(n, r)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        lay = h.lib_layout()
        U = _states(B, n, 11)
        T = np.linspace(500.0, 1200.0, B)
        d_ul = _dev(U[:, lay["species_of_lib"]])
        d_kl = torch.empty((B, lay["k_len"]), dtype=torch.float64, device="cuda")
        _sync()
        h.rate_table_lib_dev(T, d_kl.data_ptr())
        d_dul = torch.full_like(d_ul, float("nan"))
        _sync()
        h.rhs_tiled_dev(B, d_ul.data_ptr(), d_dul.data_ptr(), d_k_lib=d_kl.data_ptr())
        torch.cuda.synchronize()
        dul = d_dul.cpu().numpy()
        assert np.all(np.isfinite(dul))
        du = np.empty_like(dul); du[:, lay["species_of_lib"]] = dul
        kl = d_kl.cpu().numpy()
        on = orc.OracleNetwork.from_flat(net)
        sample = [0, 1, 63, 128, 255, 256, 257, B - 1]
        # two summation orders of n_i terms differ by ~sqrt(n_i) roundings: the top hub of the 50k network collects
        # 58 000 contributions (measured 1.1e-13 of sum |nu rate| there, against <= 1e-14 everywhere else)
        n_terms = np.bincount(np.concatenate([net.reac_idx, net.prod_idx]), minlength=n)
        tol = np.maximum(TOL, 8.0 * np.sqrt(n_terms) * 2.0 ** -53)
        for b in sample:
            k = kl[b][lay["slot_of_reaction"]]
            err = np.abs(du[b] - on.rhs(k, U[b])) / np.maximum(on.abs_rhs(k, U[b]), 1e-300)
            assert np.all(err <= tol), (n, b, err.max())
            # mass conservation (the synthetic CRN conserves sum m_i u_i)
            assert abs(du[b] @ net.mass) <= 1e-12 * (on.abs_rhs(k, U[b]) @ net.mass)
        # the drop-in entry on the same states in the CALLER's species order (kin_rhs_batched_klib_dev: at C5 the species permutation
        # goes through LDS on the way in and out), against the library-order call
        d_uc, d_duc = _dev(U), torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
        _sync()
        h.rhs_batched_klib_dev(B, d_uc.data_ptr(), d_kl.data_ptr(), d_duc.data_ptr())
        torch.cuda.synchronize()
        duc = d_duc.cpu().numpy()
        assert np.all(np.isfinite(duc))
        for b in sample:
            k = kl[b][lay["slot_of_reaction"]]
            assert np.all(np.abs(duc[b] - du[b]) <= 4e-13 * on.abs_rhs(k, U[b]) + 1e-300), (n, b)
        del d_uc, d_duc
        # linearity in k: doubling k doubles du exactly (power of two)
        d_k2 = d_kl * 2.0
        d_du2 = torch.empty_like(d_dul)
        _sync()
        h.rhs_tiled_dev(B, d_ul.data_ptr(), d_du2.data_ptr(), d_k_lib=d_k2.data_ptr())
        # temperature form on the same states
        d_T = _dev(T)
        d_duT = torch.full_like(d_ul, float("nan"))
        _sync()
        h.rhs_tiled_dev(B, d_ul.data_ptr(), d_duT.data_ptr(), d_T=d_T.data_ptr())
        torch.cuda.synchronize()
        du2 = d_du2.cpu().numpy()
        for b in sample:      # (the order of the LDS atomics differs from run to run: equal up to the summation order)
            k = kl[b][lay["slot_of_reaction"]]
            scale = on.abs_rhs(k, U[b])[lay["species_of_lib"]]
            assert np.all(np.abs(du2[b] - 2.0 * dul[b]) <= 4e-13 * scale + 1e-300)
        duT = np.empty_like(dul); duT[:, lay["species_of_lib"]] = d_duT.cpu().numpy()
        for b in (0, 128, B - 1):
            k = orc.arrhenius(Ea, A, T[b], k_max=1e12)
            bound = (2 * (Ea.max() / (R_GAS * T[b])) + 16) * 2.0 ** -53
            assert np.all(np.abs(duT[b] - on.rhs(k, U[b])) <= (bound + TOL) * on.abs_rhs(k, U[b]) + 1e-300), (n, b)
        h.close()


@pytest.mark.parametrize("n,entries", [(3000, None), (3000, "1400"), (1000, None), (6000, None), (6000, "2600")],
                         ids=["one_window_bs512", "windows_bs512", "bs256", "bs1024", "windows_bs1024"])
def test_one_slot_records_after_the_low_k_cutoff(monkeypatch, n, entries):
    """A network that lost 30 % of its reactions (apply_low_k_cutoff!, solve_utils.jl:213-245): the records without a reverse
    take one rate-constant slot each at the end of their window - the kernel's SINGLES instantiation. k-stream form through
    the layout conversions, the library-order rate table against the plain one, the temperature form, all against the oracle;
    the plain two-slot layout of the same network (KIN_TILED_SINGLES=0) passes the same check."""
    if entries:
        monkeypatch.setenv("KIN_TILED_ENTRIES", entries)
    r, B = 5 * n, 37
    net0, Ea0, A0 = synthetic_crn(n, r)
    keep = np.sort(np.random.default_rng(21).choice(r, int(0.7 * r), replace=False))
    net, Ea, A = net0.subset(keep), Ea0[keep], A0[keep]
    R2 = net.n_reactions
    U = _states(B, n, 3)
    K = 10.0 ** np.random.default_rng(4).uniform(-3, 3, (B, R2))
    h = capi.HipNetwork.from_flat(net)
    du, lay = _tiled_k(h, U, K)
    assert R2 <= lay["k_len"] < R2 + 130 * lay["windows"] and lay["k_len"] < 2 * lay["records"]
    assert (lay["windows"] > 1) == bool(entries)
    _check_against_oracle(net, du, U, K, range(B))
    _check_against_oracle(net, lay["du_klib"], U, K, range(B))      # kin_rhs_batched_klib_dev
    # rate table in library order = the plain table, permuted; the sweep fed with it against the temperature form
    h.set_arrhenius(Ea, A, k_max=1e12)
    T = np.linspace(600.0, 1400.0, B)
    d_tl = torch.full((B, lay["k_len"]), float("nan"), dtype=torch.float64, device="cuda")
    d_t = torch.empty((B, R2), dtype=torch.float64, device="cuda")
    _sync()
    h.rate_table_lib_dev(T, d_tl.data_ptr())
    _sync()
    h.rate_table_dev(T, d_t.data_ptr())
    tl, t = d_tl.cpu().numpy(), d_t.cpu().numpy()
    assert np.array_equal(tl[:, lay["slot_of_reaction"]], t)
    unowned = np.ones(lay["k_len"], bool); unowned[lay["slot_of_reaction"]] = False
    assert np.all(tl[:, unowned] == 0.0)                     # missing reverses and padding slots: written, zero
    d_ul = _dev(U[:, lay["species_of_lib"]])
    d_a, d_b, d_T = torch.full_like(d_ul, float("nan")), torch.full_like(d_ul, float("nan")), _dev(T)
    _sync()
    h.rhs_tiled_dev(B, d_ul.data_ptr(), d_a.data_ptr(), d_k_lib=d_tl.data_ptr())
    h.rhs_tiled_dev(B, d_ul.data_ptr(), d_b.data_ptr(), d_T=d_T.data_ptr())
    torch.cuda.synchronize()
    on = orc.OracleNetwork.from_flat(net)
    scale = np.stack([on.abs_rhs(t[b], U[b]) for b in range(B)])[:, lay["species_of_lib"]]
    assert np.all(np.abs(d_a.cpu().numpy() - d_b.cpu().numpy()) <= 2e-13 * scale + 1e-300)
    h.close()
    # the two-slot layout of the same network
    monkeypatch.setenv("KIN_TILED_SINGLES", "0")
    h2 = capi.HipNetwork.from_flat(net)
    du2, lay2 = _tiled_k(h2, U, K)
    assert lay2["k_len"] == 2 * lay2["records"] and lay2["records"] == lay["records"]
    _check_against_oracle(net, du2, U, K, range(B))
    h2.close()
