"""Seeded synthetic chemical reaction networks (CRNs) for the solve path.

The reference ships no CRN generator (its networks come out of the CDE binary
at doc-build time), so the build defines one, following SURVEY.md section 8(d):

* seed 12345 (the literal the reference's test runner seeds with,
  test/runtests.jl:5);
* R/2 forward reactions, each immediately followed by its exact reverse
  (mirrors the reverse duplication of src/exploration/cde.jl:299-309);
* forward types respecting max_molecularity = 2 (src/exploration/network.jl:275-279):
  25 % A->B, 35 % A->B+C, 5 % A->2B, 30 % A+B->C+D, 5 % 2A->B+C;
* species drawn with Zipf(1.1) popularity, so a few hub species touch thousands
  of reactions; A==B no-ops (network.jl:269-272) and duplicate reactions are
  rejected; every species is the first reactant of at least one reaction;
* Ea: 25 % exact zeros, else U(0, 6e5) J/mol; A: 10**U(8.8, 12.3)
  (ranges of examples/getting_started/arrhenius_params.bson).

The network is returned in the flat ragged form of `RxData`
(src/exploration/network.jl:193-203): id_reacs / stoic_reacs / id_prods /
stoic_prods as (ptr, idx, sto) triplets, 0-based.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SEED = 12345  # test/runtests.jl:5

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

    def reaction(self, r):
        a, b = self.reac_ptr[r], self.reac_ptr[r + 1]
        c, d = self.prod_ptr[r], self.prod_ptr[r + 1]
        return (list(zip(self.reac_idx[a:b], self.reac_sto[a:b])),
                list(zip(self.prod_idx[c:d], self.prod_sto[c:d])))

    def subset(self, keep):
        """Network restricted to reactions `keep` (index array), cf. splice!(rd, rids)
        (src/exploration/network.jl:514-529, which removes the complement)."""
        keep = np.asarray(keep, dtype=np.int64)
        return from_lists(self.n_species,
                          [self.reaction(r)[0] for r in keep],
                          [self.reaction(r)[1] for r in keep])


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


def synthetic_crn(n_species: int, n_reactions: int, seed: int = SEED, zipf_s: float = 1.1):
    """Generate the seeded synthetic CRN of SURVEY.md section 8(d).

    Returns (FlatNetwork, Ea[R], A[R]).
    """
    if n_reactions % 2:
        raise ValueError("n_reactions must be even (forward/reverse pairs)")
    rng = np.random.default_rng(seed)
    N, F = n_species, n_reactions // 2
    w = 1.0 / np.arange(1, N + 1) ** zipf_s
    cdf = np.cumsum(w / w.sum())

    def draw(k):
        return np.minimum(np.searchsorted(cdf, rng.random(k)), N - 1)

    types = rng.choice(len(_TYPES), size=F, p=_TYPE_P)
    seen = set()
    reacs, prods = [], []
    pool = draw(8 * F)
    pos = 0
    for f in range(F):
        rt, pt = _TYPES[types[f]]
        letters = sorted(set(rt + pt))
        while True:
            if pos + 4 > len(pool):
                pool = draw(8 * F)
                pos = 0
            cand = pool[pos:pos + len(letters)]
            pos += len(letters)
            assign = {ch: int(s) for ch, s in zip(letters, cand)}
            if f < N:
                assign["A"] = f  # every species reacts at least once
            if len(set(assign.values())) != len(letters):
                continue  # letters must be distinct species (no A->A no-ops)
            rside, pside = _side(rt, assign), _side(pt, assign)
            key = tuple(sorted([tuple(rside), tuple(pside)]))
            if key in seen:
                continue
            seen.add(key)
            reacs.append(rside); prods.append(pside)   # forward
            reacs.append(pside); prods.append(rside)   # exact reverse
            break

    net = from_lists(N, reacs, prods)
    R = n_reactions
    Ea = np.where(rng.random(R) < 0.25, 0.0, rng.uniform(0.0, 6.0e5, R))
    A = 10.0 ** rng.uniform(8.8, 12.3, R)
    return net, Ea, A


def narrow_k_variant(Ea):
    """C2's explicit-solver variant: Ea squeezed to [0, 5e4] J/mol."""
    return Ea * (5.0e4 / 6.0e5)
