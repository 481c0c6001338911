"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs. FP64 throughout; only the summation order (and FMA contraction) differs, so the bound
is 1e-13 relative to the natural scale sum_r |nu_ir rate_r| of each component."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-13


def _state(n, seed=0):
    rng = np.random.default_rng(seed)
    return 10.0 ** rng.uniform(-12, 0, n)   # LogUniform(1e-12, 1) for RHS microbenchmarks


@pytest.fixture(scope="module", params=[(300, 1500), (1000, 5000)], ids=["300x1500", "C2_1k_5k"])
def case(request):
    n, r = request.param
    net, Ea, A = synthetic_crn(n, r)
    h = capi.HipNetwork.from_flat(net)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    h.set_rates(k)
    yield net, Ea, A, h, on, k
    h.close()


def test_rhs_matches_oracle(case):
    net, Ea, A, h, on, k = case
    for seed in range(3):
        u = _state(net.n_species, seed)
        err = np.abs(h.rhs(u) - on.rhs(k, u)) / (on.abs_rhs(k, u) + 1e-300)
        assert err.max() < TOL


def test_rhs_edge_states(case):
    net, Ea, A, h, on, k = case
    u = np.zeros(net.n_species); u[0] = 1.0               # the solve's u0 (make_u0: one-hot)
    np.testing.assert_allclose(h.rhs(u), on.rhs(k, u), rtol=1e-14, atol=0)
    assert np.all(h.rhs(np.zeros(net.n_species)) == 0.0)


def test_jacobian_matches_oracle(case):
    net, Ea, A, h, on, k = case
    rowptr, col = h.jac_pattern()
    n = net.n_species
    assert rowptr[0] == 0 and np.all(np.diff(rowptr) >= 1)
    for i in range(n):                                   # sorted columns, diagonal present
        c = col[rowptr[i]:rowptr[i + 1]]
        assert np.all(np.diff(c) > 0) and i in c
    u = _state(n, 3)
    Jd = sp.csr_matrix((h.jac_values(u), col, rowptr), shape=(n, n))
    Jo = on.jac(k, u)
    D = abs(Jd - Jo)
    assert D.max() <= TOL * abs(Jo).max()
    # finite-difference check of the device Jacobian against the device RHS (independent of the oracle)
    # (mass action is at most quadratic in u_j, so a central difference is exact up to round-off)
    j = 5
    hstep = 0.25 * u[j]
    up, um = u.copy(), u.copy(); up[j] += hstep; um[j] -= hstep
    fd = (h.rhs(up) - h.rhs(um)) / (2 * hstep)
    noise = 1e-12 * (on.abs_rhs(k, up) + on.abs_rhs(k, um)) / hstep
    assert np.all(np.abs(Jd[:, j].toarray().ravel() - fd) <= noise + 1e-12 * np.abs(fd))


def test_arrhenius_matches_oracle(case, golden_dir):
    net, Ea, A, h, on, k = case
    h.set_arrhenius(Ea, A, k_max=1e12)
    for T in (500.0, 1000.0, 1200.0):
        np.testing.assert_allclose(h.rates_at(T), orc.arrhenius(Ea, A, T, k_max=1e12), rtol=2e-15)
    h.set_arrhenius(Ea, A)      # k_max = nothing dispatch (calculator.jl:229-232)
    np.testing.assert_allclose(h.rates_at(800.0), orc.arrhenius(Ea, A, 800.0), rtol=2e-15)
    h.set_rates(k)
    # the reference's own parameter file
    d = json.load(open(os.path.join(golden_dir, "arrhenius_params.json")))
    for T in (500.0, 1200.0):
        np.testing.assert_allclose(capi.arrhenius_eval(d["Ea"], d["A"], T, k_max=1e12),
                                   orc.arrhenius(d["Ea"], d["A"], T, k_max=1e12), rtol=2e-15)


def test_rate_table_matches_oracle(case):
    net, Ea, A, h, on, k = case
    h.set_arrhenius(Ea, A, k_max=1e12, t_mult=orc.tconvert("ms", "s"))
    T = np.linspace(500.0, 1200.0, 37)
    got = h.rate_table(T)
    ref = orc.rate_table(Ea, A, T, k_max=1e12, t_mult=1e-3)
    # the table kernel forms Ea/RT and the k_max cap without IEEE divisions: one ulp in the argument of
    # exp is amplified by |Ea/RT|, so the bound is elementwise (2 |Ea/RT| + 8) * 2^-53 (see kernels.hip)
    arg = np.abs(Ea)[None, :] / (8.314462618 * T[:, None])
    assert (np.abs(got - ref) <= (2.0 * arg + 8.0) * 2.0 ** -53 * np.abs(ref) + 5e-324).all()
    assert h.rate_table(T[:0]).shape == (0, net.n_reactions)     # empty table
    h.set_rates(k)


def test_batched_rhs_matches_oracle(case):
    net, Ea, A, h, on, k = case
    rng = np.random.default_rng(5)
    for B in (1, 7, 130):                                  # ragged batch sizes (padding lanes)
        U = np.stack([_state(net.n_species, 10 + b) for b in range(B)])
        K = k[None, :] * rng.uniform(0.5, 2.0, (B, 1))
        got_shared = h.rhs_batched(U)
        got_own = h.rhs_batched(U, K)
        for b in range(B):
            sc = on.abs_rhs(k, U[b]) + 1e-300
            assert (np.abs(got_shared[b] - on.rhs(k, U[b])) / sc).max() < TOL
            sc = on.abs_rhs(K[b], U[b]) + 1e-300
            assert (np.abs(got_own[b] - on.rhs(K[b], U[b])) / sc).max() < TOL


def test_special_stoichiometries_on_device():
    # 2A -> B (no 1/2!), A -> 2B, inert collider A + M -> B + M, product = reactant species
    reacs = [[(0, 2)], [(0, 1)], [(0, 1), (2, 1)], [(1, 1)]]
    prods = [[(1, 1)], [(1, 2)], [(1, 1), (2, 1)], [(0, 1), (1, 1)]]
    net = from_lists(3, reacs, prods)
    h = capi.HipNetwork.from_flat(net)
    on = orc.OracleNetwork.from_flat(net)
    k = np.array([3.0, 0.5, 2.0, 0.25]); u = np.array([0.5, 0.3, 4.0])
    h.set_rates(k)
    np.testing.assert_allclose(h.rhs(u), on.rhs(k, u), rtol=1e-15)
    rowptr, col = h.jac_pattern()
    Jd = sp.csr_matrix((h.jac_values(u), col, rowptr), shape=(3, 3)).toarray()
    np.testing.assert_allclose(Jd, on.jac(k, u).toarray(), rtol=1e-15, atol=1e-15)
    h.close()


def test_julia_index_base():
    net, Ea, A = synthetic_crn(50, 200, seed=3)
    h0 = capi.HipNetwork.from_flat(net)
    h1 = capi.HipNetwork(net.n_species, net.reac_ptr, net.reac_idx + 1, net.reac_sto, net.prod_ptr,
                         net.prod_idx + 1, net.prod_sto, index_base=1)
    k = np.linspace(1, 2, 200); u = _state(50)
    h0.set_rates(k); h1.set_rates(k)
    assert np.array_equal(h0.rhs(u), h1.rhs(u))
    r0, c0 = h0.jac_pattern(); r1, c1 = h1.jac_pattern(index_base=1)
    assert np.array_equal(r0 + 1, r1) and np.array_equal(c0 + 1, c1)
    h0.close(); h1.close()


def test_state_errors_are_reported():
    net, Ea, A = synthetic_crn(50, 200, seed=3)
    h = capi.HipNetwork.from_flat(net)
    with pytest.raises(capi.KineticaHipError) as e:
        h.rhs(np.ones(50))                                  # rates never set
    assert e.value.code == capi.KIN_ERR_STATE
    with pytest.raises(capi.KineticaHipError) as e:
        h.rates_at(500.0)                                   # no Arrhenius parameters
    assert e.value.code == capi.KIN_ERR_STATE
    h.close()


def test_full_size_properties_c3():
    """BASELINE sizes (10k / 50k): size-independent properties instead of a full oracle sweep:
    linearity in k, the conservation invariant sum_i du_i * w_i for reaction-wise balanced weights,
    determinism (bitwise reproducible), and an oracle spot check."""
    net, Ea, A = synthetic_crn(10000, 50000)
    h = capi.HipNetwork.from_flat(net)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u = _state(10000, 1)
    h.set_rates(k)
    du = h.rhs(u)
    assert np.array_equal(du, h.rhs(u))                     # fixed summation order
    h.set_rates(2.0 * k)
    assert np.array_equal(h.rhs(u), 2.0 * du)               # linear in k, exactly (power of two)
    h.set_rates(k)
    sc = on.abs_rhs(k, u) + 1e-300
    assert (np.abs(du - on.rhs(k, u)) / sc).max() < TOL
    # forward/reverse pairs cancel exactly at k_f * prod(u_reac) == k_r * prod(u_prod): use u = 1, k = 1
    h.set_rates(np.ones(50000))
    assert np.all(np.abs(h.rhs(np.ones(10000))) < 1e-9)
    # mass conservation (every generated reaction balances): sum_i m_i du_i = 0 up to round-off
    h.set_rates(k)
    assert abs(np.dot(net.mass, du)) <= 1e-11 * np.dot(net.mass, on.abs_rhs(k, u))
    B = 64
    U = np.stack([_state(10000, 100 + b) for b in range(B)])
    dU = h.rhs_batched(U)
    scale = np.array([np.dot(net.mass, on.abs_rhs(k, U[b])) for b in (0, B - 1)])
    assert abs(np.dot(dU[0], net.mass)) <= 1e-11 * scale[0] and abs(np.dot(dU[B - 1], net.mass)) <= 1e-11 * scale[1]
    # the batched sweep accumulates with relaxed LDS atomics (ds_add_f64): the ORDER of the additions is not fixed, so two
    # launches may differ in the last bits - by no more than the rounding of the accumulation itself
    dU2 = h.rhs_batched(U)
    for b in (0, B // 2, B - 1):
        assert (np.abs(dU2[b] - dU[b]) / (on.abs_rhs(k, U[b]) + 1e-300)).max() < 1e-14
    # The exact instantiation bench.py times (sweep_reg_kernel<8, 4, false, 1024>: 10k / 50k, more states than compute units,
    # every state with its OWN rate constants - Arrhenius at its own temperature, as the bench builds them), element by
    # element against the oracle on sampled states: the first and last state of the first wave of workgroups, states of
    # later trips of the state loop (their u and first k batch were prefetched across the state boundary), the last state
    B3 = 300
    U3 = np.stack([_state(10000, 500 + b) for b in range(B3)])
    T3 = np.linspace(500.0, 1200.0, B3)
    K3 = np.stack([orc.arrhenius(Ea, A, T, k_max=1e12) for T in T3])
    dU3 = h.rhs_batched(U3, K3)
    for b in (0, 1, 255, 256, 257, 280, B3 - 1):
        sc = on.abs_rhs(K3[b], U3[b]) + 1e-300
        assert (np.abs(dU3[b] - on.rhs(K3[b], U3[b])) / sc).max() < TOL, b
    h.close()


def test_full_size_properties_c5_tiled_sweep():
    """C5 size (50k / 250k): the batched sweep takes the species-tiled path (state too large for
    LDS); checked against the single-state kernel, the oracle (spot states) and for linearity."""
    net, Ea, A = synthetic_crn(50000, 250000)
    h = capi.HipNetwork.from_flat(net)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    h.set_rates(k)
    B = 5
    U = np.stack([_state(50000, 40 + b) for b in range(B)])
    rng = np.random.default_rng(9)
    K = k[None, :] * rng.uniform(0.5, 2.0, (B, 1))
    got = h.rhs_batched(U, K)
    for b in (0, B - 1):
        sc = on.abs_rhs(K[b], U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(K[b], U[b])) / sc).max() < TOL
    single = h.rhs(U[2])
    shared = h.rhs_batched(U[2:3])[0]
    sc = on.abs_rhs(k, U[2]) + 1e-300
    assert (np.abs(single - shared) / sc).max() < TOL
    assert (np.abs(single - on.rhs(k, U[2])) / sc).max() < TOL
    # more states than workgroups (shared k): every workgroup re-uses its scratch row for several
    # states, so a stale tail-u / net-rate value from the previous state would show up here
    B2 = 600
    U2 = np.stack([_state(50000, 100 + (b % 7)) * (1.0 + 0.001 * b) for b in range(B2)])
    got2 = h.rhs_batched(U2)
    for b in (0, 255, 256, 511, 512, B2 - 1):
        sc = on.abs_rhs(k, U2[b]) + 1e-300
        assert (np.abs(got2[b] - on.rhs(k, U2[b])) / sc).max() < TOL
    # the same 600 states through the LIBRARY-ORDER sweep (round 3: hubs + windows of species in LDS, one pass over k; the
    # full set of its tests is tests/test_gpu_tiled.py): converted on the device, rate constants from the library-order
    # table kernel at 1000 K, back in the caller's order; against the caller-order kernel's result and the oracle
    import torch
    lay = h.lib_layout()
    assert lay["windows"] > 1 and not lay["identity"] and lay["k_len"] == 250000
    h.set_arrhenius(Ea, A, k_max=1e12)
    d_u = torch.tensor(U2, dtype=torch.float64, device="cuda")
    d_ul, d_dul, d_du = torch.empty_like(d_u), torch.empty_like(d_u), torch.empty_like(d_u)
    d_kl = torch.empty((B2, lay["k_len"]), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    h.rate_table_lib_dev(np.full(B2, 1000.0), d_kl.data_ptr())
    h.states_to_lib_dev(B2, d_u.data_ptr(), d_ul.data_ptr())
    h.rhs_tiled_dev(B2, d_ul.data_ptr(), d_dul.data_ptr(), d_k_lib=d_kl.data_ptr())
    h.states_from_lib_dev(B2, d_dul.data_ptr(), d_du.data_ptr())
    torch.cuda.synchronize()
    got3 = d_du.cpu().numpy()
    n_terms = np.bincount(np.concatenate([net.reac_idx, net.prod_idx]), minlength=50000)
    tol3 = np.maximum(TOL, 8.0 * np.sqrt(n_terms) * 2.0 ** -53)       # the top hub sums 58 000 terms in another order
    k_tab = d_kl[0].cpu().numpy()[lay["slot_of_reaction"]]             # the table kernel's k (within its stated bound of the oracle's)
    for b in (0, 255, 256, 511, 512, B2 - 1):
        sc = on.abs_rhs(k_tab, U2[b]) + 1e-300
        assert np.all(np.abs(got3[b] - on.rhs(k_tab, U2[b])) / sc <= tol3), b
        assert np.all(np.abs(got3[b] - got2[b]) / sc <= tol3 + 300 * 2.0 ** -53), b     # k differs by the table kernel's <= 2 |Ea/RT| ulps
    # Jacobian at this size: oracle comparison
    rowptr, col = h.jac_pattern()
    h.set_rates(k)
    Jd = sp.csr_matrix((h.jac_values(U[0]), col, rowptr), shape=(50000, 50000))
    Jo = on.jac(k, U[0])
    assert abs(Jd - Jo).max() <= TOL * abs(Jo).max()
    h.close()


def test_batched_sweep_general_record_layouts():
    """The sweep's other code paths: reverse reactions listed in a separate block (the order
    duplicate_reverse produces, cde.jl:299-309 -> index-based k loads), unpaired reactions, a collider
    on both sides (explicit operands), 2A / A->2B stoichiometries."""
    rng = np.random.default_rng(21)
    # block order: forwards 0..3, reverses 4..7, plus an unpaired reaction and a collider reaction
    fwd = [([(0, 1)], [(1, 1), (2, 1)]), ([(0, 1), (1, 1)], [(3, 1), (4, 1)]), ([(2, 2)], [(3, 1), (5, 1)]), ([(4, 1)], [(5, 2)])]
    reacs = [r for r, p in fwd] + [p for r, p in fwd] + [[(5, 1)], [(0, 1), (6, 1)]]
    prods = [p for r, p in fwd] + [r for r, p in fwd] + [[(6, 1)], [(1, 1), (6, 1)]]
    net = from_lists(7, reacs, prods)
    h = capi.HipNetwork.from_flat(net)
    on = orc.OracleNetwork.from_flat(net)
    k = rng.uniform(0.5, 2.0, net.n_reactions)
    h.set_rates(k)
    B = 9
    U = rng.uniform(0.1, 1.0, (B, 7))
    K = k[None, :] * rng.uniform(0.5, 2.0, (B, 1))
    for got, kk in ((h.rhs_batched(U), None), (h.rhs_batched(U, K), K)):
        for b in range(B):
            kb = k if kk is None else kk[b]
            np.testing.assert_allclose(got[b], on.rhs(kb, U[b]), rtol=1e-13, atol=1e-14)
    h.close()
    # odd number of reactions / large batch with ragged tail on the 300-species network in block order
    net2, Ea, A = synthetic_crn(300, 1500)
    order = list(range(0, 1500, 2)) + list(range(1, 1500, 2))       # all forwards, then all reverses
    blk = net2.subset(order)
    h = capi.HipNetwork.from_flat(blk)
    on = orc.OracleNetwork.from_flat(blk)
    k = orc.arrhenius(Ea[order], A[order], 900.0, k_max=1e12)
    h.set_rates(k)
    U = np.stack([_state(300, 70 + b) for b in range(33)])
    got = h.rhs_batched(U)
    for b in (0, 17, 32):
        sc = on.abs_rhs(k, U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(k, U[b])) / sc).max() < TOL
    h.close()


def test_large_n_sweep_with_explicit_operands_and_block_order():
    """The large-N sweep (state too large for LDS) on a network that also holds colliders on both sides
    (explicit operands, slow path), unpaired reactions and an odd reaction count (index-based k loads);
    more states than workgroups so every workgroup re-uses its scratch rows."""
    net, Ea, A = synthetic_crn(12000, 60000, seed=3)
    reacs = [net.reaction(r)[0] for r in range(net.n_reactions)]
    prods = [net.reaction(r)[1] for r in range(net.n_reactions)]
    # colliders: a rarely referenced (tail) species and a hub on both sides; an unpaired decay; 2A -> B + C
    extra = [([(7, 1), (11999, 1)], [(8, 1), (11999, 1)]), ([(11998, 1), (0, 1)], [(11997, 1), (0, 1)]),
             ([(11990, 1)], [(11991, 1)]), ([(11980, 2)], [(11981, 1), (3, 1)]), ([(5, 1), (11970, 1)], [(11970, 2)])]
    for r, p in extra:
        reacs.append(r); prods.append(p)
    net2 = from_lists(12000, reacs, prods)
    assert net2.n_reactions % 2 == 1
    h = capi.HipNetwork.from_flat(net2)
    on = orc.OracleNetwork.from_flat(net2)
    rng = np.random.default_rng(5)
    k = rng.uniform(0.5, 2.0, net2.n_reactions)
    h.set_rates(k)
    B = 300
    U = np.stack([_state(12000, 200 + (b % 5)) * (1.0 + 0.01 * b) for b in range(B)])
    got = h.rhs_batched(U)
    for b in (0, 1, 255, 256, 257, B - 1):
        sc = on.abs_rhs(k, U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(k, U[b])) / sc).max() < TOL
    K = k[None, :] * rng.uniform(0.5, 2.0, (4, 1))
    got = h.rhs_batched(U[:4], K)
    for b in range(4):
        sc = on.abs_rhs(K[b], U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(K[b], U[b])) / sc).max() < TOL
    sc = on.abs_rhs(k, U[0]) + 1e-300
    assert (np.abs(h.rhs(U[0]) - on.rhs(k, U[0])) / sc).max() < TOL
    h.close()


def test_batched_sweep_block_order_pairs():
    """Reactions in the order duplicate_reverse produces (cde.jl:299-309): all forward reactions first, their
    reverses in the same order behind them. The register-resident sweep pairs reaction p with reaction P + p
    (two coalesced k streams); more states than workgroups so the cross-state pipeline is exercised."""
    net, Ea, A = synthetic_crn(2000, 10000, seed=4)
    order = np.concatenate([np.arange(0, 10000, 2), np.arange(1, 10000, 2)])
    netb = net.subset(order)
    h = capi.HipNetwork.from_flat(netb)
    on = orc.OracleNetwork.from_flat(netb)
    rng = np.random.default_rng(8)
    k = rng.uniform(0.5, 2.0, 10000)
    h.set_rates(k)
    B = 520
    U = np.stack([_state(2000, 300 + (b % 9)) * (1.0 + 0.002 * b) for b in range(B)])
    got = h.rhs_batched(U)
    for b in (0, 1, 255, 256, 300, B - 1):
        sc = on.abs_rhs(k, U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(k, U[b])) / sc).max() < TOL
    K = k[None, :] * rng.uniform(0.5, 2.0, (5, 1))
    got = h.rhs_batched(U[:5], K)
    for b in range(5):
        sc = on.abs_rhs(K[b], U[b]) + 1e-300
        assert (np.abs(got[b] - on.rhs(K[b], U[b])) / sc).max() < TOL
    h.close()


def _random_net(rng, n, n_pairs, layout):
    """Small random CRN on n species: n_pairs forward reactions (A->B, A->B+C, A+B->C+D, 2A->B+C, A->2B) with
    their reverses, laid out as adjacent pairs, as forward block + reverse block, or with some reverses dropped."""
    fwd = []
    while len(fwd) < n_pairs:
        kind = rng.integers(5)
        sp = rng.choice(n, size=4, replace=False) if n >= 4 else None
        if sp is None:
            a, b = (0, 1) if rng.random() < 0.5 else (1, 0)
            fwd.append(([(a, 1)], [(b, 1)]))
            continue
        a, b, c, d = (int(x) for x in sp)
        fwd.append([([(a, 1)], [(b, 1)]), ([(a, 1)], sorted([(b, 1), (c, 1)])), (sorted([(a, 1), (b, 1)]), sorted([(c, 1), (d, 1)])),
                    ([(a, 2)], sorted([(b, 1), (c, 1)])), ([(a, 1)], [(b, 2)])][kind])
    if layout == "adjacent":
        reacs = [x for r, p in fwd for x in (r, p)]
        prods = [x for r, p in fwd for x in (p, r)]
    elif layout == "block":
        reacs = [r for r, p in fwd] + [p for r, p in fwd]
        prods = [p for r, p in fwd] + [r for r, p in fwd]
    else:   # irregular: adjacent order with every third reverse missing
        reacs, prods = [], []
        for i, (r, p) in enumerate(fwd):
            reacs.append(r); prods.append(p)
            if i % 3:
                reacs.append(p); prods.append(r)
    return from_lists(n, reacs, prods)


@pytest.mark.parametrize("layout", ["adjacent", "block", "irregular"])
def test_batched_sweep_size_boundaries(layout):
    """Species counts around every dispatch boundary of the batched sweep (LDS capacity of the register-resident
    and general kernels, odd N, the large-N kernel), few reactions, batch sizes below / above the CU count."""
    rng = np.random.default_rng({"adjacent": 1, "block": 2, "irregular": 3}[layout])
    for n in (2, 3, 63, 64, 65, 1023, 1024, 1025, 2559, 2560, 2561, 2562, 5119, 5120, 5121, 5122, 10111, 10112, 10175, 10176, 10177,
              10178, 10239, 10240, 10241, 12001, 55471, 55472, 60001):     # > 55 471: tail operands by label again
        net = _random_net(rng, n, 150 if n >= 4 else 1, layout)
        h = capi.HipNetwork.from_flat(net)
        on = orc.OracleNetwork.from_flat(net)
        k = rng.uniform(0.5, 2.0, net.n_reactions)
        h.set_rates(k)
        for B in (1, 3, 259):
            U = 10.0 ** rng.uniform(-6, 0, (B, n))
            got = h.rhs_batched(U)
            for b in {0, B // 2, B - 1}:
                sc = on.abs_rhs(k, U[b]) + 1e-300
                assert (np.abs(got[b] - on.rhs(k, U[b])) / sc).max() < TOL, (layout, n, B, b)
        K = k[None, :] * rng.uniform(0.5, 2.0, (2, 1))
        got = h.rhs_batched(U[:2], K)
        for b in range(2):
            sc = on.abs_rhs(K[b], U[b]) + 1e-300
            assert (np.abs(got[b] - on.rhs(K[b], U[b])) / sc).max() < TOL, (layout, n, "per-state k", b)
        h.close()


def test_batched_sweep_beyond_16_bit_species_ids():
    """Networks of 65 535 species and more do not fit the packed sweep records (16-bit ids): the batched entry point
    then runs the single-state kernels per state instead of refusing the call. Same answers as the oracle."""
    rng = np.random.default_rng(11)
    for n in (65534, 65535, 70001):
        net = _random_net(rng, n, 4000, "adjacent")
        h = capi.HipNetwork.from_flat(net)
        on = orc.OracleNetwork.from_flat(net)
        k = rng.uniform(0.5, 2.0, net.n_reactions)
        h.set_rates(k)
        U = 10.0 ** rng.uniform(-6, 0, (3, n))
        K = k[None, :] * rng.uniform(0.5, 2.0, (3, 1))
        for got, kk in ((h.rhs_batched(U), [k, k, k]), (h.rhs_batched(U, K), K)):
            for b in range(3):
                sc = on.abs_rhs(kk[b], U[b]) + 1e-300
                assert (np.abs(got[b] - on.rhs(kk[b], U[b])) / sc).max() < TOL, (n, b)
        h.close()
