"""CPU tests of the tiled sweep's library order (kinetica_jl_amd/csrc/tiled.cpp): the tables are replayed on the host
(tests/tiled_replay.py follows the kernel's arithmetic) and compared with the oracle's RHS - no GPU involved."""
import os

import numpy as np
import pytest

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import from_lists, synthetic_crn
from oracle import oracle as orc
from tests.tiled_replay import k_to_lib, replay


def _check(net, seed=0, hubs=0):
    L = capi.lib_layout_host(net, hubs)
    N, R = net.n_species, net.n_reactions
    rng = np.random.default_rng(seed)
    u = 10.0 ** rng.uniform(-6, 0, N)
    k = 10.0 ** rng.uniform(-2, 2, R)
    # layout invariants
    sp = L["species_of_lib"]
    assert sorted(sp) == list(range(N))
    slots = L["slot_of_reaction"]
    assert len(set(slots)) == R and slots.min() >= 0 and slots.max() < L["k_len"] <= 2 * L["P"] + 2 * L["T"]
    assert L["win_cnt"].sum() + L["h"] == N
    assert all(q % 2 == 0 for q in L["seg_q"]) and all(b - a >= 8 for a, b in zip(L["seg_q"][:-1], L["seg_q"][1:]))
    du_lib = replay(L, u[sp], k_to_lib(L, k))
    du = np.empty(N)
    du[sp] = du_lib
    on = orc.OracleNetwork.from_flat(net)
    ref, scale = on.rhs(k, u), on.abs_rhs(k, u)
    assert np.all(np.abs(du - ref) <= 1e-13 * np.maximum(scale, 1e-300))
    return L


def test_small_network_keeps_the_callers_species_order():
    net, _, _ = synthetic_crn(300, 1500)
    L = _check(net)
    assert L["T"] == 1 and L["h"] == 300 and np.array_equal(L["species_of_lib"], np.arange(300))
    assert L["P"] == 750          # every reaction paired with its reverse


def test_special_stoichiometries_and_unpaired_reactions():
    # 2A -> B + C, A -> 2B, a collider on both sides (A + M -> B + M), its reverse, a reaction without reverse,
    # two identical reactions, a reaction whose products repeat a reactant (2M -> M + B)
    reacs = [[(0, 2)], [(1, 1), (2, 1)], [(3, 1)], [(4, 2)], [(0, 1), (5, 1)], [(1, 1), (5, 1)], [(2, 1)], [(2, 1)], [(2, 1)], [(5, 2)]]
    prods = [[(1, 1), (2, 1)], [(0, 2)], [(4, 2)], [(3, 1)], [(1, 1), (5, 1)], [(0, 1), (5, 1)], [(3, 1), (4, 1)], [(0, 1)], [(0, 1)], [(5, 1), (1, 1)]]
    net = from_lists(6, reacs, prods)
    L = _check(net, seed=3)
    assert L["P"] == 7            # three pairs, four single reactions


def test_windows_on_a_small_network(monkeypatch):
    monkeypatch.setenv("KIN_TILED_ENTRIES", "1400")
    net, _, _ = synthetic_crn(3000, 15000)
    L = _check(net, seed=1)
    assert L["T"] > 1 and L["h"] < 3000 and L["E"] <= 1400
    # every record of a segment touches no other window: implied by the replay (foreign entries are NaN there)


def test_low_k_cutoff_leftovers_stay_tileable(monkeypatch):
    """What apply_low_k_cutoff! (solve_utils.jl:213-245) leaves behind: many reactions without their reverse. They come last
    in their window and take one rate-constant slot each, so the k row is (nearly) as long as the reactions are many."""
    net, _, _ = synthetic_crn(1000, 5000)
    keep = np.sort(np.random.default_rng(5).choice(5000, 3500, replace=False))
    L = _check(net.subset(keep), seed=2)
    n_single = 2 * L["P"] - 3500              # records without a reverse
    assert L["has_singles"] and n_single > 500
    assert 3500 <= L["k_len"] <= 3500 + 2 * 64 * L["T"]
    assert np.all(L["seg_k"][:, 1] % 64 == 0)
    # the plain layout (two slots for every record) on request: same records, longer row
    monkeypatch.setenv("KIN_TILED_SINGLES", "0")
    L0 = _check(net.subset(keep), seed=2)
    assert not L0["has_singles"] and L0["k_len"] == 2 * L0["P"] and L0["P"] == L["P"]


def test_c5_size_layout():
    net, _, _ = synthetic_crn(50000, 250000)
    L = _check(net, seed=4)
    assert L["BS"] == 1024 and L["T"] <= 12 and L["P"] == 125000
    # every window fits next to the hubs
    assert L["wbase"] + L["win_cnt"].max() <= L["E"] <= 10176


@pytest.mark.parametrize("cap", [1399, 1400, 1401, 1402, 1405])
@pytest.mark.parametrize("hubs", [0, 601, 602])
def test_window_capacity_follows_the_rounded_window_base(monkeypatch, cap, hubs):
    """the window capacity is what is left behind the EVEN window base - whatever the parities of the hub count, the
    split-hub copies and the capacity, a full window ends inside the label space the launch sizes its LDS for."""
    monkeypatch.setenv("KIN_TILED_ENTRIES", str(cap))
    net, _, _ = synthetic_crn(3000, 15000)
    L = capi.lib_layout_host(net, hubs)
    assert L["T"] > 1
    assert L["wbase"] % 2 == 0 and L["E"] % 2 == 0
    assert L["wbase"] + int(L["win_cnt"].max()) <= L["E"] <= cap


def _group_cycles(L):
    """LDS cycles per (16-lane group, field) of the records' ds_add_f64 under the bank rule tools/lds_group_probe.hip
    measures: labels of one group that share a bank (label mod 16) serialise."""
    w = L["rec"].astype(np.uint64)
    fl = ((w >> np.uint64(56)) & np.uint64(7)).astype(np.int64)
    sq, rt = L["seg_q"], L["rowtab"]
    cycles = groups = 0
    for s in range(len(sq) - 1):
        base, n = int(rt[sq[s]][0]), int(rt[sq[s]:sq[s + 1], 1].sum())
        for j, sh in enumerate((0, 14, 28, 42)):
            lab = ((w[base:base + n] >> np.uint64(sh)) & np.uint64(0x3fff)).astype(np.int64)
            f = fl[base:base + n]
            use = (f & 4) == 0
            if j == 1:
                use &= (f & 1) != 0
            if j == 3:
                use &= (f & 2) != 0
            for g in range(0, n, 16):
                b = lab[g:g + 16][use[g:g + 16]] % 16
                if len(b):
                    cycles += np.bincount(b, minlength=16).max()
                    groups += 1
    return cycles / groups


def test_record_order_spreads_a_lane_groups_labels_over_the_banks(monkeypatch):
    """The library order places the records (and picks reactant / product order, the forward role of a pair and the split
    hubs' accumulator entries) so that the 16 lanes of a group hit different LDS banks: 3 cycles per group and field in the
    plain order of the synthetic CRN, 2 scheduled inside chunks of 64 records (the default), 1.5 from a reservoir. Both orders replay to the oracle's RHS."""
    net, _, _ = synthetic_crn(10000, 50000)
    L1 = _check(net, seed=7)
    monkeypatch.setenv("KIN_TILED_SCHEDULE", "0")
    L0 = _check(net, seed=7)
    c1, c0 = _group_cycles(L1), _group_cycles(L0)
    assert c0 > 2.7 and c1 < 2.25, (c0, c1)
    # a reservoir of candidates instead of chunks of 64 records: fewer conflicts still (and a wider scramble of the order)
    monkeypatch.setenv("KIN_TILED_SCHEDULE", "1"); monkeypatch.setenv("KIN_TILED_CHUNKED", "0"); monkeypatch.setenv("KIN_TILED_SCAN", "256")
    assert _group_cycles(_check(net, seed=7)) < 1.75
    # the same records, reordered: every reaction keeps exactly one rate-constant slot
    assert L0["P"] == L1["P"] and sorted(L0["slot_of_reaction"]) == sorted(L1["slot_of_reaction"])


def test_scheduled_order_with_windows_and_unpaired_reactions(monkeypatch):
    monkeypatch.setenv("KIN_TILED_ENTRIES", "1400")
    net, _, _ = synthetic_crn(3000, 15000)
    keep = np.sort(np.random.default_rng(11).choice(15000, 10500, replace=False))
    L = _check(net.subset(keep), seed=8)
    assert L["T"] > 1
    assert _group_cycles(L) < 2.4


@pytest.mark.parametrize("seed", range(12))
def test_random_networks_through_every_layout_variant(monkeypatch, seed):
    """Random small networks with every reaction shape the record format takes (A -> B, A -> 2B, A -> B + C, 2A -> ..., A + B ->
    ..., no products, colliders, duplicates), a random share of their reverses present, Zipf-like species popularity: the library
    order in all its variants (windows or not, bank-aware order from chunks / from a reservoir / plain, one- or two-slot records)
    replays to the oracle's RHS."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 400))
    nr = int(rng.integers(n, 6 * n))
    w = 1.0 / np.arange(1, n + 1) ** 1.1
    w /= w.sum()
    pick = lambda k: [int(x) for x in rng.choice(n, size=k, replace=False, p=w)]   # noqa: E731
    reacs, prods = [], []
    p_rev = rng.uniform(0.2, 1.0)
    while len(reacs) < nr:
        shape = rng.integers(0, 8)
        a, b, c, d = pick(4)
        lhs, rhs = {0: ([(a, 1)], [(b, 1)]), 1: ([(a, 1)], [(b, 2)]), 2: ([(a, 1)], [(b, 1), (c, 1)]), 3: ([(a, 2)], [(b, 1)]),
                    4: ([(a, 2)], [(b, 1), (c, 1)]), 5: ([(a, 1), (b, 1)], [(c, 1)]), 6: ([(a, 1), (b, 1)], [(c, 1), (d, 1)]),
                    7: ([(a, 1)], [])}[int(shape)]
        if rng.random() < 0.05 and int(shape) == 0:                       # a collider on both sides (A + M -> B + M)
            lhs, rhs = lhs + [(d, 1)], rhs + [(d, 1)]
        reacs.append(lhs); prods.append(rhs)
        if rhs and sum(c_ for _, c_ in rhs) <= 2 and rng.random() < p_rev:
            reacs.append(rhs); prods.append(lhs)
        if rng.random() < 0.03:                                           # an exact duplicate
            reacs.append(lhs); prods.append(rhs)
    net = from_lists(n, reacs, prods)
    variants = [{}, {"KIN_TILED_ENTRIES": "320"}, {"KIN_TILED_CHUNKED": "0", "KIN_TILED_SCAN": "200"}, {"KIN_TILED_SCHEDULE": "0"},
                {"KIN_TILED_SINGLES": "0", "KIN_TILED_ENTRIES": "320"}]
    done = 0
    for env in variants:
        for k_ in ("KIN_TILED_ENTRIES", "KIN_TILED_CHUNKED", "KIN_TILED_SCAN", "KIN_TILED_SCHEDULE", "KIN_TILED_SINGLES"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v in env.items():
            monkeypatch.setenv(k_, v)
        try:
            _check(net, seed=seed)
            done += 1
        except capi.KineticaHipError as e:     # (a tail that does not decompose into windows of 100-odd entries: reported, not wrong)
            assert e.code == capi.KIN_ERR_UNSUPPORTED and "KIN_TILED_ENTRIES" in env
    assert done >= 3
