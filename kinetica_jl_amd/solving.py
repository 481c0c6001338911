"""Host-side mirror of the reference's solve interface (src/solving/*.jl, src/analysis/io.jl):
ODESimulationParams / StaticODESolve / VariableODESolve / solve_network / ODESolveOutput, the
kinetic calculators, RxFilter and the CRN containers the solve reads. Same names, argument
meaning and error behaviour as the reference; all numerics go through the C ABI
(include/kinetica_hip.h) into libkinetica_hip.so - there is no CPU path here.

A Julia user keeps the reference's own types and calls the same library through the `ccall`
shim in INTEGRATION.md; this module is the runnable twin used by the parity tests.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np

from . import capi
from .conditions import (ConditionSet, get_initial_conditions, get_static_conditions, get_tstops, isstatic,
                         isvariable, solve_variable_conditions)
from .synth import FlatNetwork, from_lists

__all__ = ["SpeciesData", "RxData", "RxFilter", "get_filter_mask", "DummyKineticCalculator",
           "PrecalculatedArrheniusCalculator", "PrecalculatedLindemannCalculator", "allows_continuous",
           "has_conditions", "setup_network", "ODESimulationParams", "StaticODESolve", "VariableODESolve",
           "HIPBDF", "HIPRK45", "ArrheniusRates", "solve_network", "identify_next_seeds", "insert_inert", "ODESolveOutput", "ODESolution", "tconvert", "make_u0", "apply_low_k_cutoff",
           "get_max_rates", "get_initial_rates", "calculate_discrete_rates"]

_T_UNIT = {  # src/utils.jl:77-97
    "picoseconds": 1.0e-12, "ps": 1.0e-12, "nanoseconds": 1.0e-9, "ns": 1.0e-9, "microseconds": 1.0e-6, "us": 1.0e-6,
    "milliseconds": 1.0e-3, "ms": 1.0e-3, "seconds": 1.0, "s": 1.0, "minutes": 60.0, "mins": 60.0, "hours": 3600.0,
    "hrs": 3600.0, "days": 86400.0, "months": 2.6297368e06, "mts": 2.6297368e06, "years": 3.15576e07, "yrs": 3.15576e07,
}


def tconvert(*args):
    """tconvert(t, from, to) / tconvert(from, to) (src/utils.jl:21-43)."""
    t, fu, tu = (1.0, *args) if len(args) == 2 else args
    if fu not in _T_UNIT or tu not in _T_UNIT:
        raise RuntimeError("Unknown unit specified in time conversion!")
    return float(t) * _T_UNIT[fu] / _T_UNIT[tu]


# ---- CRN containers (src/exploration/network.jl:1-8, 193-203): only what the solve reads -------
@dataclass
class SpeciesData:
    toInt: Dict[str, int]      # SMILES -> species id (1-based, as in the reference)
    toStr: Dict[int, str]
    n: int
    xyz: Optional[Dict[int, dict]] = None     # per-species geometry frames; the solve side only reads "N_atoms"

    @classmethod
    def from_names(cls, names: Sequence[str], n_atoms: Optional[Sequence[int]] = None):
        xyz = None if n_atoms is None else {i + 1: {"N_atoms": int(a)} for i, a in enumerate(n_atoms)}
        return cls({s: i + 1 for i, s in enumerate(names)}, {i + 1: s for i, s in enumerate(names)}, len(names), xyz)


@dataclass
class RxData:
    nr: int
    id_reacs: List[List[int]]      # 1-based species ids
    id_prods: List[List[int]]
    stoic_reacs: List[List[int]]
    stoic_prods: List[List[int]]
    dH: Optional[List[float]] = None

    @classmethod
    def from_flat(cls, net: FlatNetwork):
        ir, ip, sr, sp = [], [], [], []
        for r in range(net.n_reactions):
            re, pr = net.reaction(r)
            ir.append([int(s) + 1 for s, _ in re]); sr.append([int(c) for _, c in re])
            ip.append([int(s) + 1 for s, _ in pr]); sp.append([int(c) for _, c in pr])
        return cls(net.n_reactions, ir, ip, sr, sp)

    def __deepcopy__(self, memo):
        # deepcopy(rd) of solve_network's copy_network (methods.jl:107-111): rows are lists of ints, so a row-wise slice copy
        # is a full copy - 20x faster than the generic deepcopy on a 50k-reaction network (0.46 s -> 0.02 s)
        rows = lambda L: [r[:] for r in L]
        return RxData(self.nr, rows(self.id_reacs), rows(self.id_prods), rows(self.stoic_reacs), rows(self.stoic_prods),
                      None if self.dH is None else list(self.dH))

    def splice(self, rids):
        """splice!(rd, rids): removes the reactions at the (0-based here) positions `rids`
        (src/exploration/network.jl:514-529)."""
        if len(rids) == 0:
            return
        kill = set(int(i) for i in rids)
        keep = [i for i in range(self.nr) if i not in kill]
        for name in ("id_reacs", "id_prods", "stoic_reacs", "stoic_prods"):
            setattr(self, name, [getattr(self, name)[i] for i in keep])
        if self.dH is not None:
            self.dH = [self.dH[i] for i in keep]
        self.nr = len(keep)

    def flat(self, n_species):
        """Flat arrays for kin_network_create (index_base = 1)."""
        rp, pp = [0], [0]
        for r in range(self.nr):
            rp.append(rp[-1] + len(self.id_reacs[r])); pp.append(pp[-1] + len(self.id_prods[r]))
        cat = lambda L: np.array([x for row in L for x in row], dtype=np.int64)
        return (n_species, np.array(rp, np.int64), cat(self.id_reacs), cat(self.stoic_reacs),
                np.array(pp, np.int64), cat(self.id_prods), cat(self.stoic_prods))


# ---- reaction filters (src/solving/filters.jl) ----------------------------------------------------
def insert_inert(rd: "RxData", sd: "SpeciesData", inert_species: Sequence[str]):
    """insert_inert!(rd, sd, inert_species) (src/solving/solve_utils.jl:126-192), topology part: every unimolecular
    reaction (one reactant with stoichiometry 1) becomes bimolecular with the inert species as a bystander on both
    sides; with several inert species the reaction is copied once per additional collision partner (new reactions
    appended, the original modified for the last partner). The reference also builds 3D geometries for new species
    and re-hashes the reactions (OpenBabel, stable_hash): chemistry metadata the solve path does not read."""
    ids = []
    for name in inert_species:
        if name not in sd.toInt:
            sd.n += 1
            sd.toInt[name] = sd.n
            sd.toStr[sd.n] = name
        ids.append(sd.toInt[name])
    uni = [i for i in range(rd.nr) if len(rd.id_reacs[i]) == 1 and rd.stoic_reacs[i][0] == 1]
    for pos, sid in enumerate(ids):
        if pos < len(ids) - 1:
            for rid in uni:
                rd.id_reacs.append(rd.id_reacs[rid] + [sid]); rd.id_prods.append(rd.id_prods[rid] + [sid])
                rd.stoic_reacs.append(rd.stoic_reacs[rid] + [1]); rd.stoic_prods.append(rd.stoic_prods[rid] + [1])
                if rd.dH is not None:
                    rd.dH.append(rd.dH[rid])
                rd.nr += 1
        else:
            for rid in uni:
                rd.id_reacs[rid] = rd.id_reacs[rid] + [sid]; rd.id_prods[rid] = rd.id_prods[rid] + [sid]
                rd.stoic_reacs[rid] = rd.stoic_reacs[rid] + [1]; rd.stoic_prods[rid] = rd.stoic_prods[rid] + [1]


class RxFilter:
    def __init__(self, filters: Optional[List[Callable]] = None, keep_filtered: bool = False):
        self.filters = [lambda sd, rd: [False] * rd.nr] if filters is None else filters
        self.keep_filtered = keep_filtered


def get_filter_mask(rf: RxFilter, sd, rd):
    """filters.jl:40-52"""
    if len(rf.filters) == 0:
        raise RuntimeError("RxFilter has not filter functions defined.")
    mask = np.zeros(rd.nr, bool)
    for f in rf.filters:
        mask |= np.asarray(f(sd, rd), dtype=bool)
    return ~mask if rf.keep_filtered else mask


# ---- kinetic calculators (src/solving/calculator.jl) ------------------------------------------------
class AbstractKineticCalculator:
    pass


class DummyKineticCalculator(AbstractKineticCalculator):
    """calculator.jl:72-158: always returns the stored rates (scaled to the time unit; note the
    cap is applied BEFORE t_mult here, calculator.jl:130-132)."""

    def __init__(self, rates, k_max=None, t_unit="s"):
        self.rates = np.asarray(rates, dtype=float).copy()
        self.k_max, self.t_unit, self.t_mult = k_max, t_unit, tconvert(t_unit, "s")

    def __call__(self, T=None, V=None):
        if T is None and V is None:
            raise TypeError("DummyKineticCalculator needs T and/or V")   # MethodError in the reference
        if self.k_max is None:
            return self.rates * self.t_mult
        return 1.0 / ((1.0 / self.k_max) + (1.0 / self.rates)) * self.t_mult

    def splice(self, rids):
        self.rates = np.delete(self.rates, np.asarray(rids, dtype=int))


class PrecalculatedArrheniusCalculator(AbstractKineticCalculator):
    """calculator.jl:164-238. k = A exp(-Ea/RT) N_A t_mult, optionally capped; evaluated on the device
    (kin_arrhenius_eval), so the cutoff decision and the solve see the same numbers."""

    def __init__(self, Ea, A, k_max=None, t_unit="s"):
        self.Ea = np.asarray(Ea, dtype=float).copy()
        self.A = np.asarray(A, dtype=float).copy()
        self.k_max, self.t_unit, self.t_mult = k_max, t_unit, tconvert(t_unit, "s")

    def __call__(self, T):
        return capi.arrhenius_eval(self.Ea, self.A, float(T), self.k_max, self.t_mult)

    def splice(self, rids):
        rids = np.asarray(rids, dtype=int)
        self.Ea = np.delete(self.Ea, rids)
        self.A = np.delete(self.A, rids)


class PrecalculatedLindemannCalculator(AbstractKineticCalculator):
    """calculator.jl:244-320: a stub in the reference (its functor throws), mirrored as such."""

    def __init__(self, Ea, A_0, A_inf, k_max=None, t_unit="s"):
        self.Ea, self.A_0, self.A_inf = map(lambda x: np.asarray(x, dtype=float), (Ea, A_0, A_inf))
        self.k_max, self.t_unit, self.t_mult = k_max, t_unit, tconvert(t_unit, "s")

    def __call__(self, **kw):
        raise RuntimeError("Lindemann calculator is not implemented yet.")   # calculator.jl:308, 313


def setup_network(sd, rd, calc):
    """setup_network! (calculator.jl:102-106, 200-204)"""
    if isinstance(calc, DummyKineticCalculator):
        if len(calc.rates) != rd.nr:
            raise ValueError(f"Number of rates ({len(calc.rates)}) does not match number of reactions in `RxData` ({rd.nr})")
    elif isinstance(calc, PrecalculatedArrheniusCalculator):
        if len(calc.Ea) != rd.nr or len(calc.A) != rd.nr:
            raise ValueError(f"Number of parameters (Ea: {len(calc.Ea)}, A: {len(calc.A)}) does not match number of reactions in `RxData` ({rd.nr})")


def has_conditions(calc, symbols):
    """calculator.jl:154-156, 234-236"""
    allowed = {"T", "V"} if isinstance(calc, DummyKineticCalculator) else {"T"}
    return all(str(s) in allowed for s in symbols)


def allows_continuous(calc):
    """calculator.jl:158, 238"""
    return isinstance(calc, (DummyKineticCalculator, PrecalculatedArrheniusCalculator))


def _call_calc(calc, conds: dict):
    if isinstance(calc, DummyKineticCalculator):
        return calc(T=conds.get("T"), V=conds.get("V"))
    return calc(T=conds["T"])


# ---- solver sentinels: what `pars.solver` holds to select this backend (HIPBDF / HIPRK45) -------------
@dataclass
class HIPBDF:
    """`ODESimulationParams(solver=HIPBDF(), ...)`: the library's variable-order BDF with the on-device sparse LU
    (kin_solve). The reference hands `pars.solver` to `init` together with `dtmin = eps(solve_chunkstep)` or
    `eps(tspan[end])` (methods.jl:164, 232, 694, 770) and `ODESimulationParams` has no field for it (params.jl:3-27):
    a solver-level option therefore lives on the solver object. `dtmin=None` keeps the reference's value.
    `warm_chunks=True` (extension): a chunkwise solve keeps difference history, order and step size across chunk starts whose
    rate constants did not change (kin_params.solve_chunks = 2) instead of re-initialising the integrator there as the
    reference does (`reinit!`, methods.jl:819): the same exact solution - chunk boundaries of a StaticODESolve are not events -
    in ~2/3 of the steps and closer to it (C3, 30 chunks: 58 against 170 tolerance units); rate updates still re-initialise."""
    dtmin: Optional[float] = None
    warm_chunks: bool = False


@dataclass
class HIPRK45:
    """`ODESimulationParams(solver=HIPRK45(), ...)`: the explicit Dormand-Prince 5(4) pair (kin_solve_explicit)."""
    dtmin: Optional[float] = None


# ---- ODESimulationParams (src/solving/params.jl) ---------------------------------------------------
@dataclass
class ODESimulationParams:
    tspan: tuple
    u0: Union[Dict[str, float], Sequence[float]]
    solver: object = None          # None / "BDF": the library's implicit BDF; "RK45" (or "DP5", "explicit"): its explicit
                                   # Dormand-Prince 5(4) pair (kin_solve_explicit); the reference takes any SciML algorithm here
    jac: bool = True
    sparse: bool = True
    abstol: float = 1.0e-10
    reltol: float = 1.0e-8
    adaptive_tols: bool = True
    update_tols: bool = False
    solve_chunks: bool = True
    solve_chunkstep: float = 1e-3
    maxiters: int = 100000
    ban_negatives: bool = False
    progress: bool = False
    save_interval: Optional[float] = None
    low_k_cutoff: Union[float, str] = "auto"     # :auto / :none / number
    low_k_maxconc: float = 2.0
    allow_short_u0: bool = False
    dtmin: Optional[float] = None   # EXTENSION kept for earlier callers (not a field of the reference's struct; prefer
                                    # `solver=HIPBDF(dtmin=...)`): None = what the reference hard-codes,
                                    # eps(solve_chunkstep) / eps(tspan[end]) (methods.jl:164, 232, 694, 770)

    def __post_init__(self):
        # validation of the keyword constructor (params.jl:77-104); ArgumentError -> ValueError
        if self.tspan[0] >= self.tspan[1]:
            raise ValueError(f"Invalid time span: Start = {self.tspan[0]}, End = {self.tspan[1]}")
        if isinstance(self.low_k_cutoff, str):
            if self.low_k_cutoff not in ("auto", "none"):
                raise ValueError("low_k_cutoff must be a numerical value or one of [:auto, :none]")
        elif self.low_k_cutoff < 0:
            raise ValueError("low_k_cutoff must be a positive number or one of [:auto, :none]")
        if self.solve_chunks:
            q = self.tspan[1] / self.solve_chunkstep
            if q != math.floor(q):
                raise ValueError("Simulation timespan is not divisible by requested chunkwise simulation step size")
        if self.solve_chunks and self.save_interval is not None and self.save_interval > self.solve_chunkstep:
            raise ValueError("Solution save interval must be less than chunkwise simulation step size")

    @property
    def explicit(self):
        """True when `solver` selects the explicit integrator."""
        if isinstance(self.solver, HIPRK45):
            return True
        if isinstance(self.solver, HIPBDF):
            return False
        if self.solver is None or (isinstance(self.solver, str) and self.solver.upper() in ("BDF", "CVODE_BDF", "IMPLICIT")):
            return False
        if isinstance(self.solver, str) and self.solver.upper() in ("RK45", "DP5", "EXPLICIT"):
            return True
        raise ValueError(f"solver must be None / 'BDF' / HIPBDF() or 'RK45' / 'DP5' / 'explicit' / HIPRK45(), got {self.solver!r}")

    @property
    def solver_dtmin(self):
        """dtmin of the solver sentinel, else the extension field, else None (the reference's own choice)."""
        d = getattr(self.solver, "dtmin", None)
        return self.dtmin if d is None else d

    def to_kin_params(self):
        return capi.KinParams(tspan0=self.tspan[0], tspan1=self.tspan[1], abstol=self.abstol, reltol=self.reltol,
                              adaptive_tols=int(self.adaptive_tols), update_tols=int(self.update_tols),
                              solve_chunks=(2 if getattr(self.solver, "warm_chunks", False) else 1) if self.solve_chunks else 0,
                              ban_negatives=int(self.ban_negatives),
                              solve_chunkstep=self.solve_chunkstep, maxiters=int(self.maxiters),
                              save_interval=-1.0 if self.save_interval is None else self.save_interval,
                              dtmin=0.0 if self.solver_dtmin is None else float(self.solver_dtmin))


# ---- solve methods (src/solving/methods.jl:7-58) -----------------------------------------------------
class StaticODESolve:
    def __init__(self, pars, conditions, calculator, filter=None):
        if not isstatic(conditions):
            raise ValueError("All conditions must be static to run a StaticODESolve.")
        if not has_conditions(calculator, conditions.symbols):
            raise ValueError("Calculator does not support all of the provided conditions.")
        self.pars, self.conditions, self.calculator = pars, conditions, calculator
        self.filter = RxFilter() if filter is None else filter


class VariableODESolve:
    def __init__(self, pars, conditions, calculator, filter=None):
        if not has_conditions(calculator, conditions.symbols):
            raise ValueError("Calculator does not support all of the provided conditions.")
        if not conditions.discrete_updates and not allows_continuous(calculator):
            raise ValueError("Calculator does not support continuous rate updates in simulations.")
        self.pars, self.conditions, self.calculator = pars, conditions, calculator
        self.filter = RxFilter() if filter is None else filter


# ---- pre-solve pipeline (src/solving/solve_utils.jl) ---------------------------------------------------
def get_max_rates(conditions, calculator):
    """solve_utils.jl:19-54: enumerate min/max corners of the variable conditions (binary order,
    minimum first), keep the corner with the greatest mean rate (first maximum)."""
    static = get_static_conditions(conditions)
    var = [(s, p.minimum(), p.maximum()) for s, p in zip(conditions.symbols, conditions.profiles) if not isstatic(p)]
    if not var:
        return _call_calc(calculator, static)
    best, best_mean = None, -math.inf
    for bits in range(2 ** len(var)):
        conds = dict(static)
        for j, (s, lo, hi) in enumerate(var):
            conds[s] = hi if (bits >> (len(var) - 1 - j)) & 1 else lo
        k = _call_calc(calculator, conds)
        if np.mean(k) > best_mean:
            best, best_mean = k, np.mean(k)
    return best


def get_initial_rates(conditions, calculator):
    """solve_utils.jl:62-73"""
    return _call_calc(calculator, get_initial_conditions(conditions))


def apply_low_k_cutoff(rd, calc, pars, conditions):
    """apply_low_k_cutoff! (solve_utils.jl:213-245). Mutates rd AND the calculator, as the
    reference does. Returns the number of removed reactions."""
    if pars.low_k_cutoff == "none":
        return 0
    k_cutoff = pars.reltol / pars.tspan[-1] if pars.low_k_cutoff == "auto" else float(pars.low_k_cutoff)
    max_rates = get_max_rates(conditions, calc) * pars.low_k_maxconc ** 2
    low = [i for i, rate in enumerate(max_rates) if rate < k_cutoff]
    rd.splice(low)
    calc.splice(low)
    return len(low)


def make_u0(sd, pars):
    """solve_utils.jl:262-297"""
    if not isinstance(pars.u0, dict):
        u0 = np.asarray(pars.u0, dtype=float)
        if len(u0) != sd.n:
            if pars.allow_short_u0:
                out = np.zeros(sd.n)
                out[:len(u0)] = u0
                return out
            raise RuntimeError("Length of supplied initial concentration vector does not match with number of species in system.")
        return u0.copy()
    out = np.zeros(sd.n)
    for spec, conc in pars.u0.items():
        if spec not in sd.toInt:
            raise RuntimeError(f"Species {spec} not in SpeciesData. Check pars.u0 is correct.")
        out[sd.toInt[spec] - 1] = conc
    return out


def discrete_stop_temperatures(conditions):
    """(tstops, T(tstops)): the condition-interpolation half of calculate_discrete_rates (solve_utils.jl:91-104) for a
    calculator that reads :T only - what kin_solve needs instead of a rate table."""
    if not conditions.discrete_updates:
        raise RuntimeError("Cannot calculate discrete rates for a continuous ConditionSet.")
    tstops = get_tstops(conditions)
    prof = conditions.profiles[conditions.symbols.index("T")]
    T = np.full(len(tstops), float(prof.value)) if isstatic(prof) else np.asarray(prof.sol(tstops), dtype=float)
    return tstops, T


def calculate_discrete_rates(conditions, calculator, nr, handle=None):
    """solve_utils.jl:91-109: k_precalc[s] = calculator(conditions interpolated at tstop_s).
    With the Arrhenius calculator and a live network handle the S x R table is generated by the
    device kernel (kin_rate_table); otherwise by S calculator calls. Returns (tstops, T or None, table)."""
    if not conditions.discrete_updates:
        raise RuntimeError("Cannot calculate discrete rates for a continuous ConditionSet.")
    tstops = get_tstops(conditions)
    static = get_static_conditions(conditions)
    var = {s: p.sol(tstops) for s, p in zip(conditions.symbols, conditions.profiles) if isvariable(p)}
    if isinstance(calculator, PrecalculatedArrheniusCalculator) and handle is not None:
        T = var["T"] if "T" in var else np.full(len(tstops), static["T"])
        return tstops, T, handle.rate_table(T)
    table = np.empty((len(tstops), nr))
    for i in range(len(tstops)):
        conds = dict(static)
        conds.update({s: v[i] for s, v in var.items()})
        table[i] = _call_calc(calculator, conds)
    return tstops, None, table


# ---- results (src/analysis/io.jl:3-48; SciMLBase solution fields the callers read) ----------------------
@dataclass
class ODESolution:
    t: np.ndarray
    u: np.ndarray            # [len(t)][n_species]
    retcode: str
    k: Optional[object] = None
    stats: dict = field(default_factory=dict)
    umax: Optional[np.ndarray] = None    # max over saved times per species, reduced on the device (kin_solution_max)

    def __call__(self, tq):
        """Linear interpolation res.sol(t) (docs/src/getting-started.md:232-236)."""
        tq = np.atleast_1d(np.asarray(tq, dtype=float))
        i = np.clip(np.searchsorted(self.t, tq, side="left"), 1, len(self.t) - 1)
        th = ((tq - self.t[i - 1]) / (self.t[i] - self.t[i - 1]))[:, None]
        return (1 - th) * self.u[i - 1] + th * self.u[i]


@dataclass
class DiscreteRates:
    """sol_k: DiffEqArray(k_precalc, tstops) (methods.jl:739, io.jl:38)."""
    t: np.ndarray
    u: np.ndarray            # [S][R']


class ArrheniusRates:
    """sol_k of a discrete-update solve with the Arrhenius calculator: the same DiffEqArray(k_precalc, tstops) seen from
    outside (`t`, `u[s]`, `len`), but no S x R table exists anywhere - not on the host, not on the device (C4: 14 001 x
    50 000 doubles = 5.6 GB). The solve received the temperatures at the stops and formed each stop's rate constants on
    the device when it reached it (kin_solve with T_stops); a row asked for here is evaluated by the same device functor
    (kin_arrhenius_eval: calculator.jl:223-232 literally), so `u[s]` is bit for bit what the integrator used.
    `u` of the whole object (`np.asarray(sol_k.u)`, `save_output`) materialises all rows on demand."""

    def __init__(self, t, T, Ea, A, k_max, t_mult):
        self.t = np.asarray(t, dtype=float)
        self.T = np.asarray(T, dtype=float)
        self._Ea, self._A, self._k_max, self._t_mult = np.array(Ea, dtype=float), np.array(A, dtype=float), k_max, t_mult
        self._full = None

    def __len__(self):
        return len(self.t)

    def row(self, s):
        if self._full is not None:
            return self._full[s]
        return capi.arrhenius_eval(self._Ea, self._A, float(self.T[s]), self._k_max, self._t_mult)

    __getitem__ = row

    @property
    def u(self):
        return _LazyRows(self)

    def materialize(self):
        if self._full is None:
            self._full = np.stack([self.row(s) for s in range(len(self.t))]) if len(self.t) else np.empty((0, len(self._Ea)))
        return self._full


class _LazyRows:
    """`sol_k.u`: rows on demand (`u[s]`), the full [S][R'] array when converted (`np.asarray(u)`, `u.shape`)."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return len(self._o)

    def __getitem__(self, s):
        if isinstance(s, (int, np.integer)):
            return self._o.row(int(s))
        return self._o.materialize()[s]

    def __array__(self, dtype=None, copy=None):
        a = self._o.materialize()
        return a if dtype is None else a.astype(dtype)

    @property
    def shape(self):
        return (len(self._o), len(self._o._Ea))


@dataclass
class ODESolveOutput:
    sd: SpeciesData
    rd: RxData
    sol: ODESolution
    sol_k: Optional[DiscreteRates]
    sol_vcs: Optional[object]
    pars: ODESimulationParams
    conditions: ConditionSet


# ---- solve_network (src/solving/methods.jl:105-130, 330-360) ----------------------------------------------
class HipIntegrator:
    """What `solve_network(...; return_integrator=true)` hands back (methods.jl:98-100, 175-178): the
    initialised integrator, living on the device. `step()` = step!(integ), `solve()` = solve!(integ),
    `t` / `u` = integ.t / integ.u. It spans the whole tspan (solve_chunks=false) or the first chunk
    (solve_chunks=true: "integrators for chunkwise solutions require significant work to fully solve
    outside of their intended solution methods", methods.jl:244). Close it (or use `with`) to free the GPU."""

    def __init__(self, handle, sd, rd, pars):
        self._h, self.sd, self.rd, self.pars = handle, sd, rd, pars

    def step(self, n=1):
        """n accepted steps; returns how many were taken (fewer at the end of the span / on failure)."""
        return self._h.integrator_step(max(int(n), 1))

    def solve(self):
        self._h.integrator_step(0)
        return self

    @property
    def t(self):
        return self._h.integrator_state(with_u=False)[0]

    @property
    def u(self):
        return self._h.integrator_state()[1]

    @property
    def retcode(self):
        return capi.RETCODE_NAMES[self._h.integrator_state(with_u=False)[2]]

    @property
    def stats(self):
        return self._h.integrator_state(with_u=False)[3]

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def solve_network(method, sd, rd, copy_network=True, return_integrator=False):
    """Solve a network with static or variable kinetics on the MI355X.

    Pipeline order as in the reference: deepcopy -> [variable: solve condition
    profiles] -> filter -> splice -> setup_network! -> low-k cutoff (mutates the copies AND the
    caller's calculator) -> rates -> u0 -> solve -> ODESolveOutput with the REDUCED network."""
    pars, conditions, calc = method.pars, method.conditions, method.calculator
    sd_a, rd_a = (copy.deepcopy(sd), copy.deepcopy(rd)) if copy_network else (sd, rd)
    variable = isinstance(method, VariableODESolve)
    if variable:
        solve_variable_conditions(conditions, pars)
    mask = get_filter_mask(method.filter, sd_a, rd_a)
    rd_a.splice(np.nonzero(mask)[0])
    setup_network(sd_a, rd_a, calc)
    apply_low_k_cutoff(rd_a, calc, pars, conditions)
    u0 = make_u0(sd_a, pars)

    h = capi.HipNetwork(*rd_a.flat(sd_a.n), index_base=1)
    try:
        arr = isinstance(calc, PrecalculatedArrheniusCalculator)
        if arr:
            h.set_arrhenius(calc.Ea, calc.A, calc.k_max, calc.t_mult)
        sol_k = None
        sol_vcs = None
        continuous = variable and not conditions.discrete_updates and arr
        if continuous:
            # continuous rate updates (methods.jl:363-653): T(t) = the profile solution, linearly interpolated
            # (src/utils.jl:135-139); a static T in the set is a constant trace
            prof = conditions.profiles[conditions.symbols.index("T")]
            if isstatic(prof):
                nodes_t, nodes_T = np.array([pars.tspan[0], pars.tspan[1]]), np.array([prof.value, prof.value], dtype=float)
            else:
                nodes_t, nodes_T = prof.sol.t, prof.sol.u
        if pars.explicit and (return_integrator or continuous):
            # the explicit pair exists behind kin_solve_explicit only (static / discrete-update solves)
            raise ValueError("solver='RK45' is not available with return_integrator=true or continuous rate updates; "
                             "use the default BDF there")
        if return_integrator:
            # methods.jl:175-178, 242-246, 445-449, 706-709: hand back the initialised integrator instead of solving
            if continuous:
                h.integrator_init_continuous(pars.to_kin_params(), u0, nodes_t, nodes_T)
            elif not variable or not conditions.discrete_updates:
                h.set_rates(get_initial_rates(conditions, calc))
                h.integrator_init(pars.to_kin_params(), u0)
            else:
                if arr:
                    tstops, T = discrete_stop_temperatures(conditions)
                    h.integrator_init(pars.to_kin_params(), u0, tstops=tstops, T_stops=T)
                else:
                    tstops, T, table = calculate_discrete_rates(conditions, calc, rd_a.nr, handle=None)
                    h.integrator_init(pars.to_kin_params(), u0, tstops=tstops, k_table=table)
            integ, h = HipIntegrator(h, sd_a, rd_a, pars), None      # the integrator owns the handle now
            return integ
        if not variable or (not conditions.discrete_updates and not arr):
            # static rates (a Dummy calculator is constant under continuous updates as well)
            k0 = get_initial_rates(conditions, calc)
            h.set_rates(k0)
            t, u, rc, st, status = h.solve(pars.to_kin_params(), u0, explicit=pars.explicit)
        elif not conditions.discrete_updates:
            # k(t) = calculator(T(t)) evaluated on the device at every step attempt
            t, u, rc, st, status = h.solve_continuous(pars.to_kin_params(), u0, nodes_t, nodes_T)
        else:
            if arr:
                # Arrhenius + discrete updates: only T(tstop) is evaluated here (calculate_discrete_rates' interpolation of
                # the profile solutions, solve_utils.jl:91-104); the rate constants of a stop are formed on the device when
                # the solve reaches it, and sol_k hands out rows on demand - no S x R table on either side
                tstops, T = discrete_stop_temperatures(conditions)
                sol_k = ArrheniusRates(tstops, T, calc.Ea, calc.A, calc.k_max, calc.t_mult)
                t, u, rc, st, status = h.solve(pars.to_kin_params(), u0, tstops=tstops, T_stops=T, explicit=pars.explicit)
            else:
                tstops, T, table = calculate_discrete_rates(conditions, calc, rd_a.nr, handle=None)
                sol_k = DiscreteRates(tstops, table)
                t, u, rc, st, status = h.solve(pars.to_kin_params(), u0, tstops=tstops, k_table=table, explicit=pars.explicit)
        if status == capi.KIN_ERR_SOLVE_FAILED:
            raise RuntimeError("ODE solution failed.")      # ErrorException (solve_utils.jl:405-411)
        if variable and not conditions.discrete_updates:
            # ODESolutionVC's condition traces (solutions.jl:1-21, rebuild_vc_solution): one per VARIABLE condition of
            # the set, whatever the calculator reads (a Dummy calculator accepts :T and :V, calculator.jl:154-156)
            sol_vcs = {}
            for sym, prof in zip(conditions.symbols, conditions.profiles):
                if not isstatic(prof):
                    sol_vcs[sym] = np.interp(t, prof.sol.t, prof.sol.u)
        if pars.update_tols and st["final_abstol"] != pars.abstol:
            pars.abstol, pars.reltol = st["final_abstol"], st["final_reltol"]   # solve_utils.jl:397-401
        umax = h.solution_max()      # what identify_next_seeds reads, reduced where the trajectory lives
    finally:
        if h is not None:
            h.close()
    sol = ODESolution(t, u, capi.RETCODE_NAMES[rc], k=sol_k, stats=st, umax=umax)
    return ODESolveOutput(sd_a, rd_a, sol, sol_k, sol_vcs, pars, conditions)


# ---- the consumer of a level's solve (src/exploration/explore_utils.jl:338-406) ---------------------------
def identify_next_seeds(sol, sd: SpeciesData, seed_conc: Optional[float] = None, *, elim_small_na: int = 0,
                        ignore: Optional[Sequence[str]] = (), saveto: Optional[str] = None) -> List[str]:
    """identify_next_seeds(sol, sd, seed_conc; elim_small_na, ignore, saveto) and the three-argument-less method
    without `seed_conc` (explore_utils.jl:338-374, 376-406): species whose maximum concentration over the saved
    trajectory reaches `seed_conc` (all species when it is None), minus `ignore`d SMILES and, with
    `elim_small_na > 0`, species of fewer atoms (`sd.xyz[i]["N_atoms"]`). `saveto` writes the reference's seeds.out
    table. The per-species maxima are `sol.umax` - reduced on the device by `kin_solution_max` when `sol` comes from
    `solve_network`, so the trajectory is never scanned on the host; a solution read back from disk (`load_output`)
    has no device side and its saved `u` is reduced here."""
    umax = getattr(sol, "umax", None)
    if umax is None:
        umax = np.max(np.asarray(sol.u, dtype=float), axis=0)
    if len(umax) != sd.n:
        raise ValueError("solution and SpeciesData disagree on the number of species")
    if elim_small_na > 0 and sd.xyz is None:
        raise ValueError("elim_small_na needs SpeciesData.xyz (N_atoms per species)")
    ignore = set(ignore or ())
    seeds, concs = [], []
    for sid in range(1, sd.n + 1):
        smi = sd.toStr[sid]
        if smi in ignore:
            continue
        c = float(umax[sid - 1])
        if seed_conc is not None and not c >= seed_conc:
            continue
        if elim_small_na > 0 and sd.xyz[sid]["N_atoms"] < elim_small_na:
            continue
        seeds.append(smi); concs.append(c)
    if saveto is not None:
        w = max(len(x) for x in seeds)      # an empty selection raises here as `maximum` of an empty list does there
        with open(saveto, "w") as f:
            f.write(f"{len(seeds)}\n")
            f.write(f"SID   {'SMILES'.ljust(w)}   Max. Conc.\n")
            for i, (smi, c) in enumerate(zip(seeds, concs), start=1):
                f.write(f"{str(i).ljust(5)} {smi.ljust(w)}   {_julia_float(c)}\n")
    return seeds


def _julia_float(x: float) -> str:
    """Julia's `print(::Float64)`: shortest round-trip digits, fixed notation for 1e-4 <= |x| < 1e6, else `d.ddde±x`."""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    if x == 0:
        return "-0.0" if str(x).startswith("-") else "0.0"
    r = repr(float(x))
    mant, _, ex = r.partition("e")
    sign = "-" if mant.startswith("-") else ""
    mant = mant.lstrip("-")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0") or "0"
    # decimal exponent of the first significant digit
    if ex:
        e10 = int(ex) + len(ip) - 1 if ip.strip("0") else int(ex) - (len(fp) - len(fp.lstrip("0"))) - 1
    else:
        e10 = len(ip) - 1 if ip.strip("0") else -(len(fp) - len(fp.lstrip("0"))) - 1
    digits = digits.rstrip("0") or "0"
    if -5 < e10 < 6:
        if e10 >= 0:
            whole = digits[: e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
        else:
            whole, frac = "0", "0" * (-e10 - 1) + digits
        return f"{sign}{whole}.{frac}"
    m = digits[0] + "." + (digits[1:] or "0")
    return f"{sign}{m}e{e10}"
