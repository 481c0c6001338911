"""CPU oracle for Kinetica.jl's solve path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product (kinetica_jl_amd/) never does. PARITY UNPINNED for trajectories: see
the header of kin_oracle.c and DESIGN.md. Every function cites the reference lines it
restates (paths relative to the reference repository root).

The arithmetic kernels live in kin_oracle.c (plain C, built by oracle/Makefile); the
host-side bookkeeping of the path (time grids, condition profiles, cutoff, u0) is restated
here in numpy.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from fractions import Fraction

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

R_GAS = 8.314462618      # src/constants.jl:4
N_A = 6.02214076e23      # src/constants.jl:5

_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build():
    """Compile kin_oracle.c with gcc (idempotent)."""
    so = os.path.join(_HERE, "libkin_oracle.so")
    src = os.path.join(_HERE, "kin_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libkin_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_jac_coo.restype = ctypes.c_int64
    return _LIB


def _pi(a):
    return a.ctypes.data_as(_i64p)


def _pf(a):
    return a.ctypes.data_as(_f64p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    for quota_f, period_f in (("/sys/fs/cgroup/cpu.max", None),
                              ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_f is None:
                quota, period = open(quota_f).read().split()
            else:
                quota, period = open(quota_f).read().strip(), open(period_f).read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(-(-int(quota) // int(period)))))
            break
        except (OSError, ValueError):
            continue
    return n


class OracleNetwork:
    """Flat CRN topology (RxData's four ragged vectors, src/exploration/network.jl:193-203)."""

    def __init__(self, n_species, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto):
        self.n = int(n_species)
        self.rp, self.ri, self.rs = _c(reac_ptr, np.int64), _c(reac_idx, np.int64), _c(reac_sto, np.int64)
        self.pp, self.pi, self.ps = _c(prod_ptr, np.int64), _c(prod_idx, np.int64), _c(prod_sto, np.int64)
        self.nr = len(self.rp) - 1

    @classmethod
    def from_flat(cls, net):
        return cls(net.n_species, net.reac_ptr, net.reac_idx, net.reac_sto,
                   net.prod_ptr, net.prod_idx, net.prod_sto)

    def _topo(self):
        return (_pi(self.rp), _pi(self.ri), _pi(self.rs), _pi(self.pp), _pi(self.pi), _pi(self.ps))

    def rhs(self, k, u):
        """make_rs mass-action ODEs (src/solving/solve_utils.jl:318-334)."""
        k, u = _c(k, np.float64), _c(u, np.float64)
        du = np.empty(self.n)
        _lib().orc_rhs(ctypes.c_int64(self.n), ctypes.c_int64(self.nr), *self._topo(), _pf(k), _pf(u), _pf(du))
        return du

    def rhs_many(self, k, U, n_threads=None):
        """B states at once, one per OpenMP thread (k: [nr] shared or [B][nr]); the all-core CPU baseline."""
        n_threads = int(n_threads or usable_cores())
        U = _c(U, np.float64)
        k = _c(k, np.float64)
        B = U.shape[0]
        DU = np.empty_like(U)
        _lib().orc_rhs_many(ctypes.c_int64(self.n), ctypes.c_int64(self.nr), *self._topo(), ctypes.c_int64(B), _pf(k),
                            ctypes.c_int64(self.nr if k.ndim == 2 else 0), _pf(U), _pf(DU), ctypes.c_int(n_threads))
        return DU

    def rates(self, k, u):
        k, u = _c(k, np.float64), _c(u, np.float64)
        out = np.empty(self.nr)
        _lib().orc_rates(ctypes.c_int64(self.nr), _pi(self.rp), _pi(self.ri), _pi(self.rs), _pf(k), _pf(u), _pf(out))
        return out

    def abs_rhs(self, k, u):
        """sum_r |nu_ir| |rate_r| - the natural error scale of du_i (tests only)."""
        rate = np.abs(self.rates(k, u))
        out = np.zeros(self.n)
        rr = np.repeat(np.arange(self.nr), np.diff(self.rp))
        np.add.at(out, self.ri, self.rs * rate[rr])
        pr = np.repeat(np.arange(self.nr), np.diff(self.pp))
        np.add.at(out, self.pi, self.ps * rate[pr])
        return out

    def jac(self, k, u):
        """Analytic sparse Jacobian (what jac=true, sparse=true yields, methods.jl:157-158) as CSR."""
        k, u = _c(k, np.float64), _c(u, np.float64)
        cap = 4 * (len(self.ri) + len(self.pi)) * 2 + 16
        row = np.empty(cap, np.int64); col = np.empty(cap, np.int64); val = np.empty(cap)
        m = _lib().orc_jac_coo(ctypes.c_int64(self.n), ctypes.c_int64(self.nr), *self._topo(), _pf(k), _pf(u),
                               ctypes.c_int64(cap), _pi(row), _pi(col), _pf(val))
        if m < 0:
            raise RuntimeError("jac capacity")
        J = sp.coo_matrix((val[:m], (row[:m], col[:m])), shape=(self.n, self.n)).tocsr()
        J.sum_duplicates()
        J.sort_indices()
        return J

    def jac_pattern(self):
        """Structural pattern incl. the diagonal (rows x reactant columns per reaction)."""
        k = np.ones(self.nr); u = np.ones(self.n)
        J = self.jac(k, u)
        P = (abs(J) + sp.eye(self.n)).tocsr()  # abs: cancellations must not drop structure
        # entries that cancel to exactly 0 with k=u=1 are still structural: rebuild from COO
        cap = 4 * (len(self.ri) + len(self.pi)) * 2 + 16
        row = np.empty(cap, np.int64); col = np.empty(cap, np.int64); val = np.empty(cap)
        m = _lib().orc_jac_coo(ctypes.c_int64(self.n), ctypes.c_int64(self.nr), *self._topo(), _pf(k), _pf(u),
                               ctypes.c_int64(cap), _pi(row), _pi(col), _pf(val))
        S = sp.coo_matrix((np.ones(m), (row[:m], col[:m])), shape=(self.n, self.n)).tocsr()
        P = (S + sp.eye(self.n)).tocsr()
        P.sort_indices()
        P.data[:] = 1.0
        return P


# ----------------------------------------------------------------------------------------
# calculators (src/solving/calculator.jl)
# ----------------------------------------------------------------------------------------
_T_UNIT = {  # src/utils.jl:77-97
    "picoseconds": 1.0e-12, "ps": 1.0e-12, "nanoseconds": 1.0e-9, "ns": 1.0e-9,
    "microseconds": 1.0e-6, "us": 1.0e-6, "milliseconds": 1.0e-3, "ms": 1.0e-3,
    "seconds": 1.0, "s": 1.0, "minutes": 60.0, "mins": 60.0, "hours": 3600.0, "hrs": 3600.0,
    "days": 86400.0, "months": 2.6297368e06, "mts": 2.6297368e06, "years": 3.15576e07, "yrs": 3.15576e07,
}


def tconvert(t, from_unit, to_unit=None):
    """src/utils.jl:21-30; tconvert(from, to) == tconvert(1.0, from, to) (utils.jl:41-43)."""
    if to_unit is None:
        t, from_unit, to_unit = 1.0, t, from_unit
    if from_unit not in _T_UNIT or to_unit not in _T_UNIT:
        raise ValueError("Unknown unit specified in time conversion!")
    return float(t) * _T_UNIT[from_unit] / _T_UNIT[to_unit]


def arrhenius(Ea, A, T, k_max=None, t_mult=1.0):
    """PrecalculatedArrheniusCalculator functor (src/solving/calculator.jl:223-232)."""
    Ea, A = _c(Ea, np.float64), _c(A, np.float64)
    out = np.empty(len(Ea))
    _lib().orc_arrhenius(ctypes.c_int64(len(Ea)), _pf(Ea), _pf(A), ctypes.c_int(k_max is not None),
                         ctypes.c_double(0.0 if k_max is None else k_max), ctypes.c_double(t_mult),
                         ctypes.c_double(T), _pf(out))
    return out


def dummy_rates(rates, k_max=None, t_mult=1.0):
    """DummyKineticCalculator functor (src/solving/calculator.jl:127-152)."""
    rates = _c(rates, np.float64)
    out = np.empty(len(rates))
    _lib().orc_dummy(ctypes.c_int64(len(rates)), _pf(rates), ctypes.c_int(k_max is not None),
                     ctypes.c_double(0.0 if k_max is None else k_max), ctypes.c_double(t_mult), _pf(out))
    return out


def rate_table(Ea, A, T_stops, k_max=None, t_mult=1.0):
    """calculate_discrete_rates (src/solving/solve_utils.jl:91-109) for the Arrhenius calculator."""
    Ea, A, T_stops = _c(Ea, np.float64), _c(A, np.float64), _c(T_stops, np.float64)
    out = np.empty((len(T_stops), len(Ea)))
    _lib().orc_rate_table(ctypes.c_int64(len(Ea)), _pf(Ea), _pf(A), ctypes.c_int(k_max is not None),
                          ctypes.c_double(0.0 if k_max is None else k_max), ctypes.c_double(t_mult),
                          _pf(T_stops), ctypes.c_int64(len(T_stops)), _pf(out))
    return out


# ----------------------------------------------------------------------------------------
# time grids (src/utils.jl:108-115 on top of Julia's float ranges)
# ----------------------------------------------------------------------------------------
def _rat(x, tol_scale=1.0):
    """Simplest rational close to x (continued fractions), as Base.rat does for float ranges."""
    if x == 0:
        return Fraction(0, 1)
    y = Fraction(x)
    f = y.limit_denominator(10 ** 9)
    if abs(float(f) - x) <= 4 * np.spacing(abs(x)) * tol_scale:
        # look for an even simpler one
        for den in (1, 10, 100, 1000, 10 ** 4, 10 ** 5, 10 ** 6, 10 ** 7, 10 ** 8, 10 ** 9):
            g = Fraction(round(x * den), den)
            if float(g) == x:
                return g
        return f
    return None


def julia_range(start, step, stop):
    """collect(start:step:stop) for Float64: Julia lifts the endpoints to rationals when they
    are 'nice' decimals, so element i is the correctly rounded start + i*step (Base
    twiceprecision.jl). Falls back to start + i*step when no rational lift exists."""
    if step <= 0:
        raise ValueError("step must be positive")
    if stop < start:
        return np.empty(0)
    fa, fs, fb = _rat(start), _rat(step), _rat(stop)
    if fa is not None and fs is not None and fb is not None:
        n = int((fb - fa) // fs) + 1
        den = math.lcm(fa.denominator, fs.denominator)
        a_n = fa.numerator * (den // fa.denominator)
        s_n = fs.numerator * (den // fs.denominator)
        if den < 2 ** 53 and abs(a_n) + n * s_n < 2 ** 62:
            return (a_n + s_n * np.arange(n, dtype=np.int64)).astype(np.float64) / float(den) \
                if abs(a_n) + n * s_n < 2 ** 53 else \
                np.array([float(Fraction(a_n + s_n * i, den)) for i in range(n)])
    n = int(math.floor((stop - start) / step + 1e-12)) + 1
    return start + step * np.arange(n)


def create_savepoints(start, stop, step):
    """src/utils.jl:108-115."""
    cstep = float(f"{step:.9g}") if (step > 1e-9 and abs(step - math.floor(step)) < 1e-9) else step
    r = julia_range(start, cstep, stop)
    if r[-1] < stop:
        r = np.append(r, stop)
    return r


# ----------------------------------------------------------------------------------------
# condition profiles (src/conditions/*.jl); plain dict records, one constructor each
# ----------------------------------------------------------------------------------------
def static_profile(value):
    """StaticConditionProfile (static.jl:7-9)."""
    return {"kind": "static", "value": float(value)}


def null_direct(X_start, t_end):
    """NullDirectProfile (direct_variable.jl:73-86)."""
    return {"kind": "nulldirect", "X_start": X_start, "t_end": t_end, "tstops": np.array([t_end])}


def linear_direct(rate, X_start, X_end):
    """LinearDirectProfile (direct_variable.jl:123-137)."""
    if (X_end < X_start and rate > 0) or (X_end > X_start and rate < 0):
        raise ValueError("Impossible temperature ramp defined.")
    t_end = (X_end - X_start) / rate
    return {"kind": "lineardirect", "rate": rate, "X_start": X_start, "X_end": X_end, "t_end": t_end,
            "tstops": np.array([t_end])}


def null_gradient(X_start, t_end):
    """NullGradientProfile (gradient_variable.jl:100-113)."""
    return {"kind": "nullgradient", "X_start": X_start, "t_end": t_end, "tstops": np.array([t_end])}


def linear_gradient(rate, X_start, X_end):
    """LinearGradientProfile (gradient_variable.jl:150-170)."""
    if (X_end < X_start and rate > 0) or (X_end > X_start and rate < 0):
        raise ValueError("Impossible condition ramp defined.")
    t_end = (X_end - X_start) / rate
    return {"kind": "lineargradient", "rate": rate, "X_start": X_start, "X_end": X_end, "t_end": t_end,
            "tstops": np.array([t_end])}


def double_ramp_gradient(X_start, t_start_plateau, rate1, X_mid, t_mid_plateau, rate2, X_end,
                         t_end_plateau, t_blend=None):
    """DoubleRampGradientProfile (gradient_variable.jl:229-273)."""
    if (X_mid > X_start and rate1 < 0) or (X_mid < X_start and rate1 > 0) or \
            (X_end > X_mid and rate2 < 0) or (X_end < X_mid and rate2 > 0):
        raise ValueError("Impossible condition ramp defined.")
    t_startr1 = t_start_plateau
    t_endr1 = t_startr1 + ((X_mid - X_start) / rate1)
    t_startr2 = t_endr1 + t_mid_plateau
    t_endr2 = t_startr2 + ((X_end - X_mid) / rate2)
    t_end = t_endr2 + t_end_plateau
    p = {"kind": "doubleramp", "rate1": rate1, "rate2": rate2, "X_start": X_start, "X_mid": X_mid,
         "X_end": X_end, "t_startr1": t_startr1, "t_endr1": t_endr1, "t_startr2": t_startr2,
         "t_endr2": t_endr2, "t_end": t_end}
    if t_blend is None:
        p["t_blend"] = 0.0
        p["tstops"] = np.array([t_startr1, t_endr1, t_startr2, t_endr2, t_end])
    else:
        b = t_blend
        p["t_blend"] = b
        p["kind"] = "doubleramp_blended"
        p["tstops"] = np.array([t_startr1 - b, t_startr1 + b, t_endr1 - b, t_endr1 + b,
                                t_startr2 - b, t_startr2 + b, t_endr2 - b, t_endr2 + b, t_end])
    return p


def profile_f(p, t):
    """Direct-profile condition function f(t, profile) (direct_variable.jl:88-90, 139-145)."""
    if p["kind"] == "nulldirect":
        return p["X_start"]
    if p["kind"] == "lineardirect":
        return ((t <= 0.0) * p["X_start"]) + ((t > 0.0 and t <= p["t_end"]) * (p["X_start"] + (p["rate"] * t))) + \
            ((t > p["t_end"]) * p["X_end"])
    raise ValueError("not a direct profile")


def profile_grad(p, t):
    """Gradient-profile function grad(t, profile) (gradient_variable.jl:115-117, 165-170, 275-299)."""
    kd = p["kind"]
    if kd == "nullgradient":
        return 0.0
    if kd == "lineargradient":
        return ((t <= p["t_end"]) * p["rate"]) + ((t > p["t_end"]) * 0.0)
    if kd == "doubleramp":
        return (((t >= p["t_startr1"] and t < p["t_endr1"]) * p["rate1"]) +
                ((t >= p["t_startr2"] and t < p["t_endr2"]) * p["rate2"]))
    if kd == "doubleramp_blended":
        b = p["t_blend"]; r1 = p["rate1"]; r2 = p["rate2"]
        s1, e1, s2, e2 = p["t_startr1"], p["t_endr1"], p["t_startr2"], p["t_endr2"]
        return (((t >= s1 - b and t < s1 + b) * (r1 * (t - s1 - b) / (2 * b) + r1)) +
                ((t >= s1 + b and t < e1 - b) * r1) +
                ((t >= e1 - b and t < e1 + b) * (-r1 * (t - e1 - b) / (2 * b))) +
                ((t >= s2 - b and t < s2 + b) * (r2 * (t - s2 - b) / (2 * b) + r2)) +
                ((t >= s2 + b and t < e2 - b) * r2) +
                ((t >= e2 - b and t < e2 + b) * (-r2 * (t - e2 - b) / (2 * b))))
    raise ValueError("not a gradient profile")


def create_discrete_tstops(p, ts_update):
    """create_discrete_tstops! (direct_variable.jl:92-95, 152-155; gradient_variable.jl:119-122,
    172-175, 301-310)."""
    if ts_update > p["t_end"]:
        raise ValueError("Error defining tstops, `ts_update` is too large.")
    kd = p["kind"]
    if kd in ("nulldirect", "nullgradient"):
        p["tstops"] = julia_range(0.0, ts_update, p["t_end"])
    elif kd in ("lineardirect", "lineargradient"):
        p["tstops"] = create_savepoints(0.0, p["t_end"], ts_update)
    else:
        b = p["t_blend"]
        p["tstops"] = np.concatenate([[0.0],
                                      create_savepoints(p["t_startr1"] - b, p["t_endr1"] + b, ts_update),
                                      create_savepoints(p["t_startr2"] - b, p["t_endr2"] + b, ts_update),
                                      [p["t_end"]]])
    return p


def _integrate_grad(p, ts):
    """Stored solution of a gradient profile on the time grid ts.

    The reference integrates D(X) ~ grad(t) with OwrenZen5 (abstol 1e-6, reltol 1e-4) and
    tstops at every kink (gradient_variable.jl:35-64; condition_set.jl:260-268). All shipped
    gradients are piecewise polynomials of degree <= 1 between those tstops, which a 5th-order
    Runge-Kutta method integrates exactly up to round-off, so the restatement integrates each
    piece in closed form (Simpson's rule is exact for degree <= 3) between consecutive grid
    points, accumulating like the time stepper does."""
    kinks = np.unique(np.asarray(p["tstops"], dtype=float))
    X = np.empty(len(ts))
    x = p["X_start"]
    t_prev = ts[0]
    X[0] = x
    for i in range(1, len(ts)):
        t = ts[i]
        cuts = [t_prev] + [c for c in kinks if t_prev < c < t] + [t]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                eps_ = (b - a) * 1e-9
                ga, gm, gb = profile_grad(p, a + eps_), profile_grad(p, 0.5 * (a + b)), profile_grad(p, b - eps_)
                x = x + (b - a) * (ga + 4.0 * gm + gb) / 6.0
        X[i] = x
        t_prev = t
    return X


def solve_variable_condition(p, tspan, save_interval=None):
    """solve_variable_condition! (direct_variable.jl:34-43; gradient_variable.jl:35-64):
    stores the profile's solution (t, u) on the save grid (plus tstops for gradient profiles)."""
    si = tspan[1] / 1000 if save_interval is None else save_interval
    grid = create_savepoints(tspan[0], tspan[1], si)
    if p["kind"] in ("nulldirect", "lineardirect"):
        p["sol_t"] = grid
        p["sol_u"] = np.array([profile_f(p, t) for t in grid], dtype=float)
    else:
        # saveat = sort(vcat(savepoints, tstops)); the integrator cannot save beyond tspan
        ts = np.sort(np.concatenate([grid, np.asarray(p["tstops"], dtype=float)]))
        ts = ts[(ts >= tspan[0]) & (ts <= tspan[1])]
        p["sol_t"] = ts
        p["sol_u"] = _integrate_grad(p, ts)
    return p


def interp_linear(ts, us, t):
    """DiffEqArray functor -> SciMLBase.LinearInterpolation (src/utils.jl:135-139):
    (1-theta)*u[i-1] + theta*u[i]; clamps to the end values outside the grid."""
    ts = np.asarray(ts); us = np.asarray(us)
    t = np.atleast_1d(np.asarray(t, dtype=float))
    i = np.clip(np.searchsorted(ts, t, side="left"), 1, len(ts) - 1)
    dt = ts[i] - ts[i - 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        th = np.where(dt > 0, (t - ts[i - 1]) / dt, 1.0)
    th = np.clip(th, 0.0, 1.0)
    if us.ndim == 1:
        return (1 - th) * us[i - 1] + th * us[i]
    return (1 - th)[:, None] * us[i - 1] + th[:, None] * us[i]


def get_tstops(profiles):
    """get_tstops (condition_set.jl:172-176): sorted unique union over variable profiles."""
    allt = [np.asarray(p["tstops"], dtype=float) for p in profiles if p["kind"] != "static"]
    if not allt:
        raise ValueError("No tstops available, all conditions in ConditionSet are static.")
    return np.unique(np.concatenate(allt))


def profile_minmax(p):
    """Base.minimum / maximum of a solved profile (abstract_profiles.jl:113-139)."""
    return float(np.min(p["sol_u"])), float(np.max(p["sol_u"]))


# ----------------------------------------------------------------------------------------
# pre-solve pipeline pieces (src/solving/solve_utils.jl)
# ----------------------------------------------------------------------------------------
def low_k_cutoff_value(low_k_cutoff, reltol, tspan_end):
    """apply_low_k_cutoff! threshold (solve_utils.jl:217-227): 'auto' -> reltol/tspan[end],
    'none' -> None, number -> itself (the reference's uType(...) there is an undefined name;
    a plain Float64 is what was meant)."""
    if low_k_cutoff == "none":
        return None
    if low_k_cutoff == "auto":
        return reltol / tspan_end
    return float(low_k_cutoff)


def low_k_keep_mask(k_max_rates, cutoff, low_k_maxconc):
    """Reactions kept by apply_low_k_cutoff! (solve_utils.jl:229-238): removed iff
    k_max * low_k_maxconc^2 < cutoff."""
    if cutoff is None:
        return np.ones(len(k_max_rates), bool)
    return ~((np.asarray(k_max_rates) * low_k_maxconc ** 2) < cutoff)


def get_max_rates_arrhenius(Ea, A, k_max, t_mult, T_min, T_max):
    """get_max_rates (solve_utils.jl:19-54) for one variable condition (T): evaluate the
    calculator at both corners, keep the corner with the greater mean rate. Enumeration order
    is min first ('0') then max ('1'); findmax returns the first maximum."""
    lo = arrhenius(Ea, A, T_min, k_max, t_mult)
    hi = arrhenius(Ea, A, T_max, k_max, t_mult)
    return lo if np.mean(lo) >= np.mean(hi) else hi


def make_u0(n_species, u0, species_index=None, allow_short_u0=False):
    """make_u0 (solve_utils.jl:262-297). u0: dict name->conc (needs species_index: name->0-based
    id) or a vector."""
    if isinstance(u0, dict):
        out = np.zeros(n_species)
        for name, conc in u0.items():
            if species_index is None or name not in species_index:
                raise KeyError(f"Species {name} not in SpeciesData. Check pars.u0 is correct.")
            out[species_index[name]] = conc
        return out
    u0 = np.asarray(u0, dtype=float)
    if len(u0) != n_species:
        if allow_short_u0 and len(u0) < n_species:
            out = np.zeros(n_species)
            out[:len(u0)] = u0
            return out
        raise ValueError("Length of supplied initial concentration vector does not match with number of species in system.")
    return u0.copy()


def chunk_grids(tspan1, chunkstep, save_interval):
    """Chunk count and per-chunk save grid (methods.jl:756-763): n = Int(tspan[2]/chunkstep)
    must be exact (params.jl:89-99)."""
    q = tspan1 / chunkstep
    if q != math.floor(q):
        raise ValueError("Simulation timespan is not divisible by requested chunkwise simulation step size")
    n_chunks = int(q)
    si = chunkstep if save_interval is None else save_interval
    saveat_local = julia_range(0.0, si, chunkstep)
    size_final = (len(saveat_local) - 1) * n_chunks + 1
    return n_chunks, saveat_local, size_final


# --- deviation statistics of a trajectory against a truth (the parity gates of tests/test_gpu_configs.py, bench.py) ---------
def deviation_stats(u, ref, abstol=1e-10, reltol=1e-8, n_major=50):
    """u, ref: [saves][N]. e = |u - ref| / (abstol + reltol |ref|) in units of the solve's tolerances. Returns
    rms      largest over the save points of the rms of e over the species (what the integrator's error test controls),
    p999     99.9th percentile of e over all entries,
    major    largest e over the `n_major` species with the largest concentrations in `ref` (where a unit is a RELATIVE error
             of reltol: the species a user of the trajectory reads),
    max      largest e over everything - REPORTED, not gated: over 10 000 species it moves by 30 % and more under
             perturbations that leave the algorithm alone (profiles/r04_truth_maxima_noise.txt)."""
    import numpy as np
    u, ref = np.atleast_2d(np.asarray(u, float)), np.atleast_2d(np.asarray(ref, float))
    e = np.abs(u - ref) / (abstol + reltol * np.abs(ref))
    major = np.argsort(np.abs(ref).max(axis=0))[-n_major:]
    return {"rms": float(np.sqrt((e ** 2).mean(axis=1)).max()), "p999": float(np.percentile(e, 99.9)),
            "major": float(e[:, major].max()), "max": float(e.max())}
