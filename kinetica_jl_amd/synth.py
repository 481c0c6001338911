"""Seeded synthetic chemical reaction networks (CRNs) for the solve path.

The reference ships no CRN generator (its networks come out of the CDE binary at doc-build
time), so the build defines one. It follows SURVEY.md section 8(d) -

* seed 12345 (the literal the reference's test runner seeds with, test/runtests.jl:5);
* R/2 forward reactions, each immediately followed by its exact reverse (mirrors the reverse
  duplication of src/exploration/cde.jl:299-309);
* forward types respecting max_molecularity = 2 (src/exploration/network.jl:275-279):
  25 % A->B, 35 % A->B+C, 5 % A->2B, 30 % A+B->C+D, 5 % 2A->B+C;
* species drawn with Zipf(1.1) popularity, so a few hub species touch thousands of reactions;
  A==B no-ops (network.jl:269-272) and duplicate reactions are rejected; every species is the
  first reactant of at least one reaction;
* Arrhenius parameters in the ranges of examples/getting_started/arrhenius_params.bson
  (Ea 0 ... 6e5 J/mol with ~25 % exact zeros, A = 10**U(8.8, 12.3))

- with two physical constraints the survey's recipe lacks and without which the ODE system is
  not a chemical one (found the hard way: randomly wired `A -> B + C` networks create mass, blow
  up exponentially to concentrations ~1e9 and defeat every stiff integrator, SciPy's included):

* **mass conservation**: every species carries an integer mass (1..24, "number of heavy
  atoms"); a reaction is only generated if both sides have the same total mass, so
  sum_i m_i u_i is an invariant of the ODEs (returned as `net.mass`);
* **detailed balance**: every species carries a free energy G_i; a pair has the intrinsic barrier
  E0 (50 % exactly 0) and Ea_forward = max(0, dG) + E0, Ea_reverse = max(0, -dG) + E0 with the
  same prefactor, so k_f / k_r = exp(-dG / RT) before the k_max cap.

The network is returned in the flat ragged form of `RxData`
(src/exploration/network.jl:193-203): id_reacs / stoic_reacs / id_prods / stoic_prods as
(ptr, idx, sto) triplets, 0-based.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

SEED = 12345  # test/runtests.jl:5
N_MASS = 24

# (reactant template, product template); letters are distinct species
_TYPES = [
    ("A", "B"),      # A -> B
    ("A", "BC"),     # A -> B + C
    ("A", "BB"),     # A -> 2B
    ("AB", "CD"),    # A + B -> C + D
    ("AA", "BC"),    # 2A -> B + C
]
_TYPE_P = np.array([0.25, 0.35, 0.05, 0.30, 0.05])


@dataclass
class FlatNetwork:
    """Flat ragged CRN (the four ragged vectors of RxData + counts)."""
    n_species: int
    n_reactions: int
    reac_ptr: np.ndarray  # int64[R+1]
    reac_idx: np.ndarray  # int64[nnz_reac]   0-based species ids
    reac_sto: np.ndarray  # int64[nnz_reac]
    prod_ptr: np.ndarray
    prod_idx: np.ndarray
    prod_sto: np.ndarray
    mass: Optional[np.ndarray] = None   # conserved weights: sum_i mass_i * du_i/dt == 0

    def reaction(self, r):
        a, b = self.reac_ptr[r], self.reac_ptr[r + 1]
        c, d = self.prod_ptr[r], self.prod_ptr[r + 1]
        return (list(zip(self.reac_idx[a:b], self.reac_sto[a:b])),
                list(zip(self.prod_idx[c:d], self.prod_sto[c:d])))

    def subset(self, keep):
        """Network restricted to reactions `keep` (index array), cf. splice!(rd, rids)
        (src/exploration/network.jl:514-529, which removes the complement)."""
        keep = np.asarray(keep, dtype=np.int64)
        sub = from_lists(self.n_species,
                         [self.reaction(r)[0] for r in keep],
                         [self.reaction(r)[1] for r in keep])
        sub.mass = self.mass
        return sub


def from_lists(n_species, reacs, prods):
    """Build a FlatNetwork from per-reaction [(species, stoich), ...] lists."""
    R = len(reacs)
    rp = np.zeros(R + 1, np.int64)
    pp = np.zeros(R + 1, np.int64)
    ri, rs, pi, ps = [], [], [], []
    for r in range(R):
        for s, c in reacs[r]:
            ri.append(s); rs.append(c)
        for s, c in prods[r]:
            pi.append(s); ps.append(c)
        rp[r + 1] = len(ri)
        pp[r + 1] = len(pi)
    return FlatNetwork(n_species, R, rp, np.array(ri, np.int64), np.array(rs, np.int64),
                       pp, np.array(pi, np.int64), np.array(ps, np.int64))


def _side(template, assign):
    """'AA' -> [(a,2)], 'BC' -> [(b,1),(c,1)] with species sorted for canonical form."""
    out = {}
    for ch in template:
        out[assign[ch]] = out.get(assign[ch], 0) + 1
    return sorted(out.items())


class _Sampler:
    """Zipf-weighted draws, globally and inside one mass bucket."""

    def __init__(self, weights, mass, rng):
        self.rng, self.mass = rng, mass
        self.cdf = np.cumsum(weights / weights.sum())
        self.n = len(weights)
        self.bucket = {}
        for m in range(1, N_MASS + 1):
            ids = np.nonzero(mass == m)[0]
            if len(ids):
                w = weights[ids]
                self.bucket[m] = (ids, np.cumsum(w / w.sum()))
        self.pool = np.empty(0, np.int64)
        self.pos = 0

    def any(self):
        if self.pos >= len(self.pool):
            self.pool = np.minimum(np.searchsorted(self.cdf, self.rng.random(65536)), self.n - 1)
            self.pos = 0
        self.pos += 1
        return int(self.pool[self.pos - 1])

    def of_mass(self, m):
        if m not in self.bucket:
            return None
        ids, cdf = self.bucket[m]
        return int(ids[min(np.searchsorted(cdf, self.rng.random()), len(ids) - 1)])


def _draw(kind, forced_a, smp, mass):
    """One mass-balanced assignment {letter: species} for forward type `kind`, or None to retry."""
    rt, pt = _TYPES[kind]
    a = forced_a if forced_a is not None else smp.any()
    ma = int(mass[a])
    if (rt, pt) == ("A", "B"):
        b = smp.of_mass(ma)
        return None if b is None else {"A": a, "B": b}
    if (rt, pt) == ("A", "BC"):
        if ma < 2:
            return None
        b = smp.any()
        if mass[b] >= ma:
            return None
        c = smp.of_mass(ma - int(mass[b]))
        return None if c is None else {"A": a, "B": b, "C": c}
    if (rt, pt) == ("A", "BB"):
        if ma % 2:
            return None
        b = smp.of_mass(ma // 2)
        return None if b is None else {"A": a, "B": b}
    if (rt, pt) == ("AB", "CD"):
        b = smp.any()
        tot = ma + int(mass[b])
        c = smp.any()
        md = tot - int(mass[c])
        if md < 1 or md > N_MASS:
            return None
        d = smp.of_mass(md)
        return None if d is None else {"A": a, "B": b, "C": c, "D": d}
    # 2A -> B + C
    b = smp.any()
    mc = 2 * ma - int(mass[b])
    if mc < 1 or mc > N_MASS:
        return None
    c = smp.of_mass(mc)
    return None if c is None else {"A": a, "B": b, "C": c}


def synthetic_crn(n_species: int, n_reactions: int, seed: int = SEED, zipf_s: float = 1.1):
    """Generate the seeded synthetic CRN. Returns (FlatNetwork, Ea[R], A[R]); `net.mass` holds
    the conserved species masses."""
    if n_reactions % 2:
        raise ValueError("n_reactions must be even (forward/reverse pairs)")
    rng = np.random.default_rng(seed)
    N, F = n_species, n_reactions // 2
    w = 1.0 / np.arange(1, N + 1) ** zipf_s
    # masses: light species are more numerous (m = 1 + floor(24 r^2)); species 0 - the feedstock
    # that carries the initial concentration in the solve configurations - is mid-weight (16), so it
    # can both decompose and combine
    mass = 1 + np.minimum((N_MASS * rng.random(N) ** 2).astype(np.int64), N_MASS - 1)
    mass[0] = 16
    G = rng.uniform(0.0, 2.0e5, N)           # species free energies, J/mol
    smp = _Sampler(w, mass, rng)

    types = rng.choice(len(_TYPES), size=F, p=_TYPE_P)
    seen = set()
    reacs, prods, dG = [], [], []
    for f in range(F):
        forced = f if f < N else None            # every species reacts at least once
        kind = int(types[f])
        tries = 0
        while True:
            tries += 1
            if tries % 64 == 0:                   # this species cannot play this role: change the type
                kind = int(rng.choice(len(_TYPES), p=_TYPE_P))
            assign = _draw(kind, forced, smp, mass)
            if assign is None:
                continue
            rt, pt = _TYPES[kind]
            letters = sorted(set(rt + pt))
            if len(set(assign[ch] for ch in letters)) != len(letters):
                continue  # letters must be distinct species (no A->A no-ops)
            rside, pside = _side(rt, assign), _side(pt, assign)
            key = tuple(sorted([tuple(rside), tuple(pside)]))
            if key in seen:
                continue
            seen.add(key)
            reacs.append(rside); prods.append(pside)   # forward
            reacs.append(pside); prods.append(rside)   # exact reverse
            dG.append(sum(G[s] * c for s, c in pside) - sum(G[s] * c for s, c in rside))
            break

    net = from_lists(N, reacs, prods)
    net.mass = mass.astype(np.float64)
    dG = np.asarray(dG)
    E0 = np.where(rng.random(F) < 0.5, 0.0, rng.uniform(0.0, 2.5e5, F))
    Apair = 10.0 ** rng.uniform(8.8, 12.3, F)
    Ea = np.empty(2 * F); A = np.empty(2 * F)
    Ea[0::2] = np.maximum(0.0, dG) + E0
    Ea[1::2] = np.maximum(0.0, -dG) + E0
    A[0::2] = Apair
    A[1::2] = Apair
    return net, Ea, A


def narrow_k_variant(Ea):
    """C2's explicit-solver variant: Ea squeezed to [0, 5e4] J/mol."""
    return Ea * (5.0e4 / 6.5e5)
