"""Condition profiles and ConditionSet - host-side mirror of the reference's src/conditions/*.jl
(only the numeric semantics the solve path consumes: X(t), tstops, minima/maxima; the
Symbolics registration machinery of condition_set.jl:204-232 exists solely to feed
ModelingToolkit and has no counterpart here).

Names, arguments and error behaviour follow the reference so that its tests
(test/Main/conditions.jl) read the same against this module.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

__all__ = ["StaticConditionProfile", "NullDirectProfile", "LinearDirectProfile", "NullGradientProfile",
           "LinearGradientProfile", "DoubleRampGradientProfile", "ConditionSet", "isstatic", "isvariable",
           "get_profile", "get_tstops", "get_t_final", "get_initial_conditions", "get_static_conditions",
           "solve_variable_conditions", "create_savepoints", "float_range"]


# ---- Julia float ranges -----------------------------------------------------------------
def _lift(x):
    """Rational lift of a 'nice' decimal float (what Base's range code does with rat())."""
    if x == 0:
        return Fraction(0)
    for den in (1, 10, 100, 1000, 10 ** 4, 10 ** 5, 10 ** 6, 10 ** 7, 10 ** 8, 10 ** 9):
        g = Fraction(round(x * den), den)
        if float(g) == x:
            return g
    f = Fraction(x).limit_denominator(10 ** 9)
    return f if float(f) == x else None


def float_range(start, step, stop):
    """collect(start:step:stop) for Float64 with Julia's correctly-rounded elements."""
    if step <= 0:
        raise ValueError("step must be positive")
    if stop < start:
        return np.empty(0)
    fa, fs, fb = _lift(start), _lift(step), _lift(stop)
    if fa is not None and fs is not None and fb is not None:
        n = int((fb - fa) // fs) + 1
        return np.array([float(fa + i * fs) for i in range(n)])
    n = int(math.floor((stop - start) / step + 1e-12)) + 1
    return start + step * np.arange(n)


def create_savepoints(start, stop, step):
    """src/utils.jl:108-115: range of savepoints that always contains the final time."""
    cstep = float(f"{step:.9g}") if (step > 1e-9 and abs(step - math.floor(step)) < 1e-9) else step
    r = float_range(start, cstep, stop)
    if r[-1] < stop:
        r = np.append(r, stop)
    return r


# ---- stored profile solutions (DiffEqArray / ODESolution stand-in) ------------------------
class ProfileSolution:
    """(t, u) samples with the linear interpolation of src/utils.jl:135-139."""

    def __init__(self, t, u):
        self.t = np.asarray(t, dtype=float)
        self.u = np.asarray(u, dtype=float)

    def __call__(self, tq):
        tq = np.atleast_1d(np.asarray(tq, dtype=float))
        i = np.clip(np.searchsorted(self.t, tq, side="left"), 1, len(self.t) - 1)
        dt = self.t[i] - self.t[i - 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            th = np.where(dt > 0, (tq - self.t[i - 1]) / dt, 1.0)
        th = np.clip(th, 0.0, 1.0)
        return (1 - th) * self.u[i - 1] + th * self.u[i]


# ---- profiles --------------------------------------------------------------------------------
class AbstractConditionProfile:
    pass


class StaticConditionProfile(AbstractConditionProfile):
    """static.jl:7-9"""

    def __init__(self, value):
        self.value = value


class AbstractVariableProfile(AbstractConditionProfile):
    sol = None

    def minimum(self):
        """abstract_profiles.jl:113-118"""
        if self.sol is None:
            raise RuntimeError("Condition profile is missing a solution.")
        return float(np.min(self.sol.u))

    def maximum(self):
        """abstract_profiles.jl:134-139"""
        if self.sol is None:
            raise RuntimeError("Condition profile is missing a solution.")
        return float(np.max(self.sol.u))


class AbstractDirectProfile(AbstractVariableProfile):
    def solve(self, pars, reset=False):
        """solve_variable_condition! (direct_variable.jl:34-43)"""
        if self.sol is None or reset:
            si = pars.tspan[1] / 1000 if pars.save_interval is None else pars.save_interval
            t = create_savepoints(pars.tspan[0], pars.tspan[1], si)
            self.sol = ProfileSolution(t, [self.f(tp, self) for tp in t])


class AbstractGradientProfile(AbstractVariableProfile):
    def solve(self, pars, reset=False):
        """solve_variable_condition! (gradient_variable.jl:35-64). The reference integrates
        D(X) ~ grad(t) with OwrenZen5 (abstol 1e-6, reltol 1e-4, condition_set.jl:260-268) and
        tstops at every kink; every shipped gradient is piecewise linear between those stops, which
        a 5th-order Runge-Kutta pair integrates exactly, so each piece is integrated in closed
        form (Simpson, exact to degree 3) on the same save grid (savepoints + tstops)."""
        if self.sol is None or reset:
            si = pars.tspan[1] / 1000 if pars.save_interval is None else pars.save_interval
            grid = create_savepoints(pars.tspan[0], pars.tspan[1], si)
            ts = np.sort(np.concatenate([grid, np.asarray(self.tstops, dtype=float)]))
            ts = ts[(ts >= pars.tspan[0]) & (ts <= pars.tspan[1])]
            kinks = np.unique(np.asarray(self.tstops, dtype=float))
            X = np.empty(len(ts))
            x = self.X_start
            X[0] = x
            for i in range(1, len(ts)):
                a0, b0 = ts[i - 1], ts[i]
                # kinks strictly inside (a0, b0): none when every stop is a grid point (the usual case, `ts` contains the
                # stops); found by bisection on the sorted kinks, not by a scan (C4 has 14 001 stops x 15 000 intervals)
                lo, hi = np.searchsorted(kinks, a0, side="right"), np.searchsorted(kinks, b0, side="left")
                cuts = [a0] + list(kinks[lo:hi]) + [b0]
                for a, b in zip(cuts[:-1], cuts[1:]):
                    if b > a:
                        e = (b - a) * 1e-9
                        x = x + (b - a) * (self.grad(a + e, self) + 4.0 * self.grad(0.5 * (a + b), self) + self.grad(b - e, self)) / 6.0
                X[i] = x
            self.sol = ProfileSolution(ts, X)


class NullDirectProfile(AbstractDirectProfile):
    """direct_variable.jl:49-95"""

    def __init__(self, *, X_start, t_end):
        self.X_start, self.t_end = float(X_start), float(t_end)
        self.tstops = np.array([self.t_end])
        self.f = lambda t, p: p.X_start

    def create_discrete_tstops(self, ts_update):
        if ts_update > self.t_end:
            raise ValueError("Error defining tstops, `ts_update` is too large.")
        self.tstops = float_range(0.0, ts_update, self.t_end)


class LinearDirectProfile(AbstractDirectProfile):
    """direct_variable.jl:101-155"""

    def __init__(self, *, rate, X_start, X_end):
        if (X_end < X_start and rate > 0) or (X_end > X_start and rate < 0):
            raise RuntimeError("Impossible temperature ramp defined. Check heating rates have the correct signs.")
        self.rate, self.X_start, self.X_end = float(rate), float(X_start), float(X_end)
        self.t_end = (self.X_end - self.X_start) / self.rate
        self.tstops = np.array([self.t_end])
        self.f = _f_linear_direct

    def create_discrete_tstops(self, ts_update):
        if ts_update > self.t_end:
            raise ValueError("Error defining tstops, `ts_update` is too large.")
        self.tstops = create_savepoints(0.0, self.t_end, ts_update)


def _f_linear_direct(t, p):
    """direct_variable.jl:139-145"""
    return ((t <= 0.0) * p.X_start) + ((0.0 < t <= p.t_end) * (p.X_start + (p.rate * t))) + ((t > p.t_end) * p.X_end)


class NullGradientProfile(AbstractGradientProfile):
    """gradient_variable.jl:70-122"""

    def __init__(self, *, X_start, t_end):
        self.X_start, self.t_end = float(X_start), float(t_end)
        self.tstops = np.array([self.t_end])
        self.grad = lambda t, p: 0.0

    def create_discrete_tstops(self, ts_update):
        if ts_update > self.t_end:
            raise ValueError("Error defining tstops, `ts_update` is too large.")
        self.tstops = float_range(0.0, ts_update, self.t_end)


class LinearGradientProfile(AbstractGradientProfile):
    """gradient_variable.jl:128-175"""

    def __init__(self, *, rate, X_start, X_end):
        if (X_end < X_start and rate > 0) or (X_end > X_start and rate < 0):
            raise RuntimeError("Impossible condition ramp defined. Check heating rates have the correct signs.")
        self.rate, self.X_start, self.X_end = float(rate), float(X_start), float(X_end)
        self.t_end = (self.X_end - self.X_start) / self.rate
        self.tstops = np.array([self.t_end])
        self.grad = lambda t, p: ((t <= p.t_end) * p.rate) + ((t > p.t_end) * 0.0)

    def create_discrete_tstops(self, ts_update):
        if ts_update > self.t_end:
            raise ValueError("Error defining tstops, `ts_update` is too large.")
        self.tstops = create_savepoints(0.0, self.t_end, ts_update)


class DoubleRampGradientProfile(AbstractGradientProfile):
    """gradient_variable.jl:181-310"""

    def __init__(self, *, X_start, t_start_plateau, rate1, X_mid, t_mid_plateau, rate2, X_end, t_end_plateau,
                 t_blend=None):
        if (X_mid > X_start and rate1 < 0) or (X_mid < X_start and rate1 > 0) or \
                (X_end > X_mid and rate2 < 0) or (X_end < X_mid and rate2 > 0):
            raise RuntimeError("Impossible condition ramp defined. Check heating rates have the correct signs.")
        self.rate1, self.rate2 = float(rate1), float(rate2)
        self.X_start, self.X_mid, self.X_end = float(X_start), float(X_mid), float(X_end)
        self.t_start_plateau, self.t_mid_plateau, self.t_end_plateau = float(t_start_plateau), float(t_mid_plateau), float(t_end_plateau)
        self.t_startr1 = self.t_start_plateau
        self.t_endr1 = self.t_startr1 + ((self.X_mid - self.X_start) / self.rate1)
        self.t_startr2 = self.t_endr1 + self.t_mid_plateau
        self.t_endr2 = self.t_startr2 + ((self.X_end - self.X_mid) / self.rate2)
        self.t_end = self.t_endr2 + self.t_end_plateau
        if t_blend is None:
            self.t_blend = 0.0
            self.tstops = np.array([self.t_startr1, self.t_endr1, self.t_startr2, self.t_endr2, self.t_end])
            self.grad = _grad_double_ramp
        else:
            b = self.t_blend = float(t_blend)
            self.tstops = np.array([self.t_startr1 - b, self.t_startr1 + b, self.t_endr1 - b, self.t_endr1 + b,
                                    self.t_startr2 - b, self.t_startr2 + b, self.t_endr2 - b, self.t_endr2 + b, self.t_end])
            self.grad = _grad_double_ramp_blended

    def create_discrete_tstops(self, ts_update):
        if ts_update > self.t_end:
            raise ValueError("Error defining tstops, `ts_update` is too large.")
        b = self.t_blend
        self.tstops = np.concatenate([[0.0], create_savepoints(self.t_startr1 - b, self.t_endr1 + b, ts_update),
                                      create_savepoints(self.t_startr2 - b, self.t_endr2 + b, ts_update), [self.t_end]])


def _grad_double_ramp(t, p):
    """gradient_variable.jl:275-283"""
    return float(((p.t_startr1 <= t < p.t_endr1) * p.rate1) + ((p.t_startr2 <= t < p.t_endr2) * p.rate2))


def _grad_double_ramp_blended(t, p):
    """gradient_variable.jl:285-299"""
    b, r1, r2 = p.t_blend, p.rate1, p.rate2
    s1, e1, s2, e2 = p.t_startr1, p.t_endr1, p.t_startr2, p.t_endr2
    return float(((s1 - b <= t < s1 + b) * (r1 * (t - s1 - b) / (2 * b) + r1)) + ((s1 + b <= t < e1 - b) * r1) +
                 ((e1 - b <= t < e1 + b) * (-r1 * (t - e1 - b) / (2 * b))) +
                 ((s2 - b <= t < s2 + b) * (r2 * (t - s2 - b) / (2 * b) + r2)) + ((s2 + b <= t < e2 - b) * r2) +
                 ((e2 - b <= t < e2 + b) * (-r2 * (t - e2 - b) / (2 * b))))


# ---- ConditionSet ------------------------------------------------------------------------------
def isstatic(x, sym=None):
    """abstract_profiles.jl:23-31 / condition_set.jl:61-72"""
    if isinstance(x, ConditionSet):
        if sym is not None:
            return isstatic(get_profile(x, sym))
        return all(isstatic(p) for p in x.profiles)
    return isinstance(x, StaticConditionProfile)


def isvariable(x, sym=None):
    """abstract_profiles.jl:47-55 / condition_set.jl:74-85 (all-variable, as in the reference)"""
    if isinstance(x, ConditionSet):
        if sym is not None:
            return isvariable(get_profile(x, sym))
        return all(isvariable(p) for p in x.profiles)
    return isinstance(x, AbstractVariableProfile)


class ConditionSet:
    """condition_set.jl:1-58: symbols, profiles, discrete_updates, ts_update."""

    def __init__(self, d, ts_update=None):
        self.symbols = list(d.keys())
        self.profiles = []
        for sym in self.symbols:
            v = d[sym]
            if isinstance(v, (int, float)) and not isinstance(v, bool):
                self.profiles.append(StaticConditionProfile(v))
            elif isinstance(v, AbstractConditionProfile):
                if ts_update is not None and isinstance(v, AbstractVariableProfile):
                    v.create_discrete_tstops(ts_update)
                self.profiles.append(v)
            else:
                raise ValueError(f"Condition {sym} does not have a valid profile.")   # ArgumentError
        self.discrete_updates = ts_update is not None
        self.ts_update = ts_update


def get_profile(cs, sym):
    """condition_set.jl:93-99"""
    if sym not in cs.symbols:
        raise KeyError(f"Condition {sym} does not exist in this ConditionSet")
    return cs.profiles[cs.symbols.index(sym)]


def get_initial_conditions(cs):
    """condition_set.jl:113-123"""
    return {s: (p.value if isstatic(p) else p.X_start) for s, p in zip(cs.symbols, cs.profiles)}


def get_static_conditions(cs):
    """condition_set.jl:133-141"""
    return {s: p.value for s, p in zip(cs.symbols, cs.profiles) if isstatic(p)}


def get_tstops(cs):
    """condition_set.jl:172-176: sorted unique union of the variable profiles' tstops."""
    if isstatic(cs):
        raise RuntimeError("No tstops available, all conditions in ConditionSet are static.")
    return np.unique(np.concatenate([np.asarray(p.tstops, dtype=float) for p in cs.profiles if isvariable(p)]))


def get_t_final(cs):
    """condition_set.jl:187-191"""
    if isstatic(cs):
        raise RuntimeError("No t_end available, all conditions in ConditionSet are static.")
    return max(p.t_end for p in cs.profiles if isvariable(p))


def solve_variable_conditions(cs, pars, reset=False):
    """solve_variable_conditions! (condition_set.jl:260-268)"""
    for p in cs.profiles:
        if isvariable(p):
            p.solve(pars, reset=reset)
