"""CPU oracle of the implicit solve - TEST INFRASTRUCTURE ONLY (see oracle.py's header).

The reference hands integration to a user-supplied SciML solver; its documented choice is
Sundials CVODE_BDF with the KLU sparse LU (docs/src/getting-started.md:69), neither of which
is vendored or runnable here (Project.toml:36-60 pins only compat bounds; docs/Project.toml:9
leaves Sundials unpinned). The stand-in restated here is the published quasi-constant-step,
variable-order (1-5) BDF/NDF of Shampine & Reichelt, "The MATLAB ODE Suite", SIAM J. Sci.
Comput. 18 (1997): backward differences D, Newton corrector on  d - c f(y_pred + d) + psi = 0
with c = h / alpha_k, local error kappa-corrected, order chosen from the error estimates at
k-1, k, k+1. SciPy's solve_ivp(method="BDF") implements the same paper, and the integrator core below
(KAPPA / GAMMA / ALPHA / ERROR_CONST, _R, change_D, the step and order-selection logic) is ADAPTED FROM
SciPy's scipy/integrate/_ivp/bdf.py (BSD-3-Clause, (c) SciPy developers) - restructured for restarts,
segment-local time, the ode15s/CVODE corrector acceptance, history resets and the reference's
orchestration. tests/test_oracle_bdf.py therefore shows a faithful adaptation (same step sequence as
SciPy with scipy_newton=True), not an independent pin. Sparse LU: SuperLU (scipy.sparse.linalg.splu) with a
minimum-degree ordering - the same class of CPU solver as KLU.

PARITY UNPINNED against the reference itself: trajectories are checked against closed forms and
a high-accuracy Radau integration instead (tests/golden/make_truth.py).

The orchestration around the integrator restates src/solving/methods.jl (chunk loop, save
grid stitching, discrete rate updates) and adaptive_solve! (solve_utils.jl:376-424).
"""
from __future__ import annotations

import math
import time

import os

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

MAX_ORDER = 5
NEWTON_MAXITER = 4
MIN_FACTOR = 0.2
MAX_FACTOR = 10.0
FIRST_MAX_FACTOR = 1e4     # growth cap of the first step-size selection after a (re)initialisation (CVODE: ETAMX1), MAX_FACTOR afterwards
EPS = np.finfo(float).eps

KAPPA = np.array([0, -0.1850, -1 / 9, -0.0823, -0.0415, 0])
GAMMA = np.hstack((0, np.cumsum(1 / np.arange(1, MAX_ORDER + 1))))
ALPHA = (1 - KAPPA) * GAMMA
ERROR_CONST = KAPPA * GAMMA + 1 / np.arange(1, MAX_ORDER + 2)

RET_SUCCESS, RET_MAXITERS, RET_DTMIN, RET_UNSTABLE = 0, 1, 2, 3
NEG_DEEP = 1e3


def rms(x):
    return np.linalg.norm(x) / math.sqrt(x.size)


def _R(order, factor):
    I = np.arange(1, order + 1)[:, None]
    J = np.arange(1, order + 1)
    M = np.zeros((order + 1, order + 1))
    M[1:, 1:] = (I - 1 - factor * J) / I
    M[0] = 1
    return np.cumprod(M, axis=0)


def change_D(D, order, factor):
    RU = _R(order, factor).dot(_R(order, 1))
    D[:order + 1] = RU.T.dot(D[:order + 1])


class OracleBDF:
    """fun(y) -> f, jac(y) -> scipy CSR (autonomous system: the rate constants are frozen
    between restarts, exactly as in the discrete-rate solves of methods.jl:655-865)."""

    def __init__(self, fun, jac, n, atol, rtol, dtmin=0.0, ban_negatives=False, scipy_newton=False, lu_band=0.35, lu_slots=128):
        # scipy_newton=True reproduces SciPy's corrector acceptance (tolerance from Hairer's RADAU5
        # heuristic, no acceptance on the first iteration) and exists only for the step-for-step
        # pin test; the production rule below is the one of the BDF codes themselves.
        self.scipy_newton = scipy_newton
        # LU cache, mirroring Solver (solver.cpp): factorisations are kept in slots and reused - across step-size changes
        # and across restarts - whenever a slot's c_fact is within lu_band of the current c = h / alpha_k, the Newton
        # update scaled by 2 / (1 + c / c_fact) (CVODE's gamrat correction); a slot is refreshed only when a corrector
        # that used it fails. lu_band = 0: a new factorisation at every change of c (SciPy's behaviour).
        self.lu_band = 0.0 if scipy_newton else lu_band
        self.lu_slots = lu_slots
        # guards of the cache (solver.cpp): a reused factorisation must contract at least 10-fold per iteration
        # (lu_rate_max); after an error-test rejection the retry gets a factorisation of its own (force_fresh_lu, with a
        # new Jacobian if the old one is more than 20 steps old); a reused slot that needed every allowed iteration is dropped
        self.lu_rate_max = 0.15
        self.lu_max_age = 50             # restarts a slot stays on offer after its Jacobian was evaluated
        self.lu_drift_max = 1.0          # drift guard: see Solver::restart (solver.cpp); 0.25 until round 4
        self.jac_stamp_now = 0
        self.cache_suspended = False     # a tolerance retry runs its chunk without the cache
        self.force_fresh_lu = False
        self.steps_since_jac = 0
        self.slot_is_fresh = True
        self.slots = []          # dicts: c_fact, LU, last_use, crate, ...
        self.cur_crate = 1.0
        self.crate_fresh = False
        self.nc_crate, self.nc_crate_step, self.nc_crate_restart = 1.0, 0, -1     # the single factorisation of the cache-less mode
        self.crate_max_age = 10  # accepted steps a measured rate is trusted for (and never across a restart)
        self.crate_dy_max = 0.2  # a first correction larger than this always gets a second iteration (solver.cpp: crate_dy_max)
        self.use_clock = 0
        self.c_fact = 0.0
        self.fun, self.jac, self.n = fun, jac, n
        self.dtmin = dtmin
        self.ban_negatives = ban_negatives
        self.set_tols(atol, rtol)
        self.stats = dict(n_steps=0, n_rejected=0, n_rhs=0, n_jac=0, n_factor=0, n_linsolve=0, n_newton_fail=0,
                          n_restarts=0, n_lu_reused=0)
        self.I = sp.identity(n, format="csc")
        self.iters_left = 0
        self.first_selection = False
        self.pre_attempt = None   # continuous rates: called with the local time of every step attempt
        # EXPERIMENT HOOK (tools/err_cap_experiment.py; None = the algorithm as built, everywhere else): a cap on the error
        # estimate of the WORST species, in error weights - a step whose rms error test passes but whose worst species is above
        # the cap is treated as an error-test failure of size worst / cap (docs/DESIGN_HISTORY.md R5.13)
        self.err_cap = None

    def set_tols(self, atol, rtol):
        self.atol, self.rtol = atol, rtol
        if self.scipy_newton:
            self.newton_tol = max(10 * EPS / rtol, min(0.03, rtol ** 0.5))
        else:
            # ode15s accepts the corrector when the estimated iteration error is below 0.05*rtol in its
            # relative norm, i.e. 0.05 in units of the error weights atol + rtol*|y| used here (Shampine &
            # Reichelt 1997, sec. 2.3); CVODE uses 0.1 of its error-test constant. Both accept on the first
            # iteration when the correction is already that small.
            # 0.03 at rtol >= 3.3e-9 (rounds 2-4: 0.05 collapsed the step size in two of 140 sweep solves), rising to CVODE's
            # 0.1 at rtol <= 1e-9, where 0.03 asks for less than the rounding of the right-hand side leaves
            # (kinetica_jl_amd/csrc/solver_kernels.hpp bdf_newton_frac has the rule and its measurements)
            self.newton_tol = max(10 * EPS / rtol, min(0.1, max(0.03, 1e-10 / rtol)))

    def _f(self, y):
        self.stats["n_rhs"] += 1
        return self.fun(y)

    def invalidate_lu(self):
        """The cache lives within one solve (solve_entry does the same)."""
        self.slots = []
        self.use_clock = 0
        self.LU = None

    def _nearest_slot(self, c):
        best, bd = None, 1e300
        for sl in self.slots:
            if self.stats["n_restarts"] - sl["jac_stamp"] > self.lu_max_age:
                continue
            if self.pre_attempt is not None and self.stats["n_steps"] - sl["step_stamp"] > 50:
                continue          # continuous rate updates: the Jacobian behind a slot is at most 50 accepted steps old
            r = abs(math.log(c / sl["c_fact"]))
            if r < bd and abs(c / sl["c_fact"] - 1.0) <= self.lu_band:
                best, bd = sl, r
        return best

    def _factor_into(self, sl, c):
        if sl is None:
            expired = [q for q in self.slots if self.stats["n_restarts"] - q["jac_stamp"] > self.lu_max_age]
            if expired:
                sl = expired[0]
            elif len(self.slots) < self.lu_slots:
                sl = {}
                self.slots.append(sl)
            else:
                sl = min(self.slots, key=lambda q: q["last_use"])
        sl["LU"] = self._factor(c)
        sl["c_fact"] = c
        sl["crate"] = 1.0               # contraction rate this factorisation has shown (CVODE's crate; 1 = unknown)
        sl["crate_step"], sl["crate_restart"] = 0, -1     # when that rate was last measured
        sl["jd"] = self.J.diagonal().copy()
        sl["jac_stamp"] = self.jac_stamp_now
        sl["step_stamp"] = self.stats["n_steps"] - self.steps_since_jac
        self.use_clock += 1
        sl["last_use"] = self.use_clock
        return sl

    def _corrector_cached(self, c, y_pred, psi, scale):
        """The cached factorisation closest to this c, else a new one; a failure with a matrix that was not made in this
        attempt from a current Jacobian refreshes the slot (Jacobian at the predictor, factorisation at this c), one retry."""
        sl = self._nearest_slot(c)
        fresh = False
        if sl is not None and not self.force_fresh_lu:
            self.use_clock += 1
            sl["last_use"] = self.use_clock
            self.stats["n_lu_reused"] += 1
        else:
            if self.force_fresh_lu and not self.jac_current and self.steps_since_jac > 20:
                self.J = self.jac(y_pred); self.stats["n_jac"] += 1
                self.jac_current = True
                self.steps_since_jac = 0
                self.jac_stamp_now = self.stats["n_restarts"]
            sl = self._factor_into(sl, c)
            fresh = self.jac_current
        self.force_fresh_lu = False
        while True:
            self.LU, self.c_fact = sl["LU"], sl["c_fact"]
            self.slot_is_fresh = fresh or sl["c_fact"] == c
            self.cur_crate = sl["crate"]
            self.crate_fresh = self._crate_fresh(sl["crate"], sl["crate_step"], sl["crate_restart"])
            converged, n_iter, y_new, d = self._newton(y_pred, c, psi, scale)
            if n_iter > 1:                  # a rate was measured
                sl["crate"], sl["crate_step"], sl["crate_restart"] = self.cur_crate, self.stats["n_steps"], self.stats["n_restarts"]
            if converged:
                if not fresh and n_iter >= NEWTON_MAXITER:
                    self.slots = [q for q in self.slots if q is not sl]          # too stale to be offered again
                return converged, n_iter, y_new, d
            self.stats["n_newton_fail"] += 1
            if fresh:
                return converged, n_iter, y_new, d
            if not self.jac_current:
                self.J = self.jac(y_pred); self.stats["n_jac"] += 1
                self.jac_current = True
                self.steps_since_jac = 0
                self.jac_stamp_now = self.stats["n_restarts"]
            self._factor_into(sl, c)
            fresh = True

    def _crate_fresh(self, crate, step, restart):
        """May the remembered rate decide a step after ONE iteration? Only when it was measured in this restart segment and
        at most crate_max_age accepted steps ago (Solver::crate_fresh, solver.cpp)."""
        return (not self.scipy_newton) and crate < 1.0 and restart == self.stats["n_restarts"] and \
            self.stats["n_steps"] - step <= self.crate_max_age

    def _factor(self, c):
        self.stats["n_factor"] += 1
        if self.n <= 64:
            return ("dense", sla.lu_factor(np.eye(self.n) - c * self.J.toarray()))
        return ("sparse", spla.splu((self.I - c * self.J).tocsc(), permc_spec="MMD_AT_PLUS_A"))

    def _lusolve(self, LU, b):
        self.stats["n_linsolve"] += 1
        return sla.lu_solve(LU[1], b) if LU[0] == "dense" else LU[1].solve(b)

    def restart(self, t0, y0, t_bound):
        """reinit!-like restart: order 1, initial step from the standard two-evaluation
        heuristic (Hairer, Norsett & Wanner, Solving ODEs I, II.4), fresh Jacobian."""
        self.stats["n_restarts"] += 1
        self.t = t0
        y0 = np.array(y0, dtype=float)
        f0 = self._f(y0)
        if not np.all(np.isfinite(f0)):
            return False
        interval = abs(t_bound - t0)
        scale = self.atol + np.abs(y0) * self.rtol
        if self.scipy_newton:
            # SciPy mode (tests/test_oracle_bdf.py pins it to scipy.integrate.BDF's own step sequence): SciPy's select_initial_step
            d0, d1 = rms(y0 / scale), rms(f0 / scale)
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            h0 = min(h0, interval)
            f1 = self._f(y0 + h0 * f0)
            if not np.all(np.isfinite(f1)):
                return False
            d2 = rms((f1 - f0) / scale) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** 0.5
            self.h_abs = min(100 * h0, h1, interval)
        else:
            # CVODE's initial step (cvode.c: cvHin / cvUpperBoundH0 / cvYddNorm; oracle/cpu_bdf.cpp `cvhin` has the description),
            # rounded DOWN to a power of ten by exact IEEE operations only (kinetica_jl_amd/csrc/solver.cpp: decade_floor)
            hlb = 100.0 * EPS * max(abs(t0), abs(t_bound))
            hub_inv = float(np.max(np.abs(f0) / (0.1 * np.abs(y0) + scale)))
            hub = 0.1 * interval
            if hub * hub_inv > 1.0:
                hub = 1.0 / hub_inv
            hg = hnew = math.sqrt(hlb * hub)
            if hub >= hlb:
                for count in range(1, 5):
                    f1 = self._f(y0 + hg * f0)
                    if not np.all(np.isfinite(f1)):
                        return False
                    ydd = rms((f1 - f0) / scale) / hg
                    hnew = math.sqrt(2.0 / ydd) if ydd * hub * hub > 2.0 else math.sqrt(hg * hub)
                    if count == 4:
                        break
                    hrat = hnew / hg
                    if 0.5 < hrat < 2.0:
                        break
                    if count > 1 and hrat > 2.0:
                        hnew = hg
                        break
                    hg = hnew
            h0 = min(max(0.5 * hnew, hlb), hub)
            h0 = min(h0, interval)
            p = 1.0
            while p > h0:
                p /= 10.0
            while p * 10.0 <= h0:
                p *= 10.0
            self.h_abs = p
        self.first_selection = not self.scipy_newton   # growth cap 1e4 at the first selection after a (re)initialisation (CVODE's ETAMX1)
        self.D = np.zeros((MAX_ORDER + 3, self.n))
        self.D[0] = y0
        self.D[1] = f0 * self.h_abs
        self.order = 1
        self.n_equal = 0
        self.J = self.jac(y0); self.stats["n_jac"] += 1
        self.steps_since_jac = 0
        self.jac_stamp_now = self.stats["n_restarts"]
        if self.lu_band > 0 and self.lu_drift_max > 0 and self.slots:
            # drift of diag(I - c_s J) of every slot against today's Jacobian at the same c_s
            jd_now = self.J.diagonal()
            keep = []
            for sl in self.slots:
                with np.errstate(all="ignore"):
                    q = (1.0 - sl["c_fact"] * sl["jd"]) / (1.0 - sl["c_fact"] * jd_now)
                    dev = np.where(q > 0.0, np.maximum(q, 1.0 / q), 1e300)
                    dev = np.where(np.isnan(dev), 1e300, dev)
                if dev.max() - 1.0 <= self.lu_drift_max:
                    keep.append(sl)
            self.slots = keep
        self.LU = None
        self.jac_current = True
        self.pending = None
        self.fail_score = 0.0
        return True

    def _reset_history(self):
        """After repeated step failures the interpolated difference history is not trusted any more:
        drop to order 1 and rebuild it from f at the current state, keeping the (already reduced)
        step size - CVODE's strategy after MXNEF1 error-test failures (Hindmarsh et al., SUNDIALS,
        ACM TOMS 31, 2005, sec. 2.1)."""
        y0 = self.D[0].copy()
        f0 = self._f(y0)
        self.D[:] = 0.0
        self.D[0] = y0
        self.D[1] = f0 * self.h_abs
        self.order = 1
        self.n_equal = 0
        self.LU = None
        self.fail_score = 0.0
        self.slots = []                   # three failed attempts in a row: nothing cached is trusted any more either
        self.stats["n_resets"] = self.stats.get("n_resets", 0) + 1

    def resume(self, rates_changed):
        """Warm continuation at a segment boundary (see Solver::resume in solver.cpp): history, order
        and step size are kept, the Jacobian is refreshed when the rates changed."""
        self.stats["n_restarts"] += 1
        self.t = 0.0
        if rates_changed:
            self.J = self.jac(self.D[0]); self.stats["n_jac"] += 1
            self.LU = None
            self.jac_current = True

    def step(self, t_bound):
        """One accepted step towards t_bound. Returns 'ok' | 'dtmin' | 'maxiters'."""
        t, D, order = self.t, self.D, self.order
        accepted = False
        first_attempt = True
        while not accepted:
            self.iters_left -= 1
            if self.iters_left < 0:
                return "maxiters"
            min_step = max(self.dtmin, 10 * (np.nextafter(t, np.inf) - t))
            if self.h_abs < min_step:
                if not first_attempt:
                    return "dtmin"
                # a step that merely starts below the resolution of t is raised to it
                change_D(D, order, min_step / self.h_abs)
                self.h_abs = min_step
                self.n_equal = 0
                self.LU = None
            first_attempt = False
            t_new = t + self.h_abs
            if t_new - t_bound > 0:
                t_new = t_bound
                change_D(D, order, abs(t_new - t) / self.h_abs)
                self.n_equal = 0
                self.LU = None
            h = t_new - t
            self.h_abs = abs(h)
            y_pred = np.sum(D[:order + 1], axis=0)
            scale = self.atol + self.rtol * np.abs(y_pred)
            psi = np.dot(D[1:order + 1].T, GAMMA[1:order + 1]) / ALPHA[order]
            c = h / ALPHA[order]
            if self.pre_attempt is not None:
                self.pre_attempt(t_new)
            converged = False
            use_cache = self.lu_band > 0 and not self.cache_suspended
            while use_cache:
                converged, n_iter, y_new, d = self._corrector_cached(c, y_pred, psi, scale)
                break
            while not use_cache:
                if self.LU is None:
                    self.LU = self._factor(c)
                    self.c_fact = c
                    self.nc_crate, self.nc_crate_step, self.nc_crate_restart = 1.0, 0, -1
                self.cur_crate = self.nc_crate
                self.crate_fresh = self._crate_fresh(self.nc_crate, self.nc_crate_step, self.nc_crate_restart)
                converged, n_iter, y_new, d = self._newton(y_pred, c, psi, scale)
                if n_iter > 1:
                    self.nc_crate, self.nc_crate_step, self.nc_crate_restart = self.cur_crate, self.stats["n_steps"], self.stats["n_restarts"]
                if converged:
                    break
                self.stats["n_newton_fail"] += 1
                if self.jac_current:
                    break
                self.J = self.jac(y_pred); self.stats["n_jac"] += 1
                self.LU = None
                self.jac_current = True
            if not converged or (self.ban_negatives and np.any(y_new < 0)):
                # a failed corrector cuts the step to a quarter and does NOT count towards the history reset (CVODE: ETACF = 0.25;
                # its history is rebuilt only after repeated error-test failures); a banned negative state halves it and counts
                eta = 0.25 if not converged else 0.5
                self.h_abs *= eta
                change_D(D, order, eta)
                self.n_equal = 0
                self.LU = None
                self.stats["n_rejected"] += 1
                self.first_selection = False   # (CVODE: any failed attempt sets etamax = 1, the first step's 1e4 is gone)
                if converged:
                    self.fail_score += 1.0
                if self.fail_score >= 3.0 and order > 1:
                    self._reset_history()
                    order = self.order
                continue
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter)
            scale = self.atol + self.rtol * np.abs(y_new)
            err_norm = rms(ERROR_CONST[order] * d / scale)
            if self.err_cap is not None and err_norm <= 1:
                worst = float(np.max(np.abs(ERROR_CONST[order] * d / scale)))
                if worst > self.err_cap:
                    err_norm = max(1.0001, worst / self.err_cap)
            if err_norm > 1:
                factor = max(MIN_FACTOR, safety * err_norm ** (-1 / (order + 1)))
                self.h_abs *= factor
                change_D(D, order, factor)
                self.n_equal = 0
                # without the cache the matrix is kept for the retry; with it the retry gets a factorisation of its own
                self.force_fresh_lu = self.lu_band > 0 and not self.cache_suspended
                self.stats["n_rejected"] += 1
                self.first_selection = False
                self.fail_score += 1.0
                if self.fail_score >= 3.0 and order > 1:
                    self._reset_history()
                    order = self.order
            else:
                # an accepted step that leaves a species below -NEG_DEEP error weights ends the segment as Unstable: the negative
                # excursion, given up early (kinetica_jl_amd/csrc/solver_kernels.hpp BDF_NEG_DEEP has the reasoning)
                if np.any(y_new < -NEG_DEEP * scale):
                    return "unstable"
                accepted = True
        self.stats["n_steps"] += 1
        self.steps_since_jac += 1
        self.fail_score = max(0.0, self.fail_score - 0.2)
        self.n_equal += 1
        self.t = t_new
        D[order + 2] = d - D[order + 1]
        D[order + 1] = d
        for i in reversed(range(order + 1)):
            D[i] += D[i + 1]
        self.jac_current = False
        self.pending = None
        if self.n_equal >= order + 1:
            err_m = rms(ERROR_CONST[order - 1] * D[order] / scale) if order > 1 else np.inf
            err_p = rms(ERROR_CONST[order + 1] * D[order + 2] / scale) if order < MAX_ORDER else np.inf
            self.pending = (err_m, err_norm, err_p, safety)
        return "ok"

    def _newton(self, y_pred, c, psi, scale):
        d = np.zeros_like(y_pred)
        y = y_pred.copy()
        dy_norm_old = None
        converged = False
        k = 0
        # a factorisation made for another c (cache hit, or kept across an error-test rejection): update scaled by
        # 2 / (1 + c / c_fact)
        upd = 2.0 / (1.0 + c / self.c_fact) if (self.c_fact != c and not self.scipy_newton) else 1.0
        # CVODE's carried convergence rate (cvNlsConvTest: crate <- max(CRDOWN crate, del / delp), CRDOWN = 0.3, reset to 1
        # by every linear-solver setup): each factorisation keeps the contraction its iterations have shown, and the FIRST
        # iteration of a step is judged with it, like the later ones are with the rate measured inside the step
        crate0 = crate = self.cur_crate
        for k in range(NEWTON_MAXITER):
            f = self._f(y)
            dy = self._lusolve(self.LU, c * f - psi - d)
            if upd != 1.0:
                dy = dy * upd
            if not np.all(np.isfinite(dy)):
                break
            dy_norm = rms(dy / scale)
            rate = None if dy_norm_old is None else dy_norm / dy_norm_old
            if rate is not None and np.isfinite(dy_norm):
                crate = max(0.3 * crate, rate)
            rate_max = self.lu_rate_max if (self.lu_band > 0 and not self.cache_suspended and not self.slot_is_fresh) else 1.0
            if rate is not None and (rate >= rate_max or rate ** (NEWTON_MAXITER - k) / (1 - rate) * dy_norm > self.newton_tol):
                break
            y += dy
            d += dy
            if dy_norm == 0 or (rate is not None and rate / (1 - rate) * dy_norm < self.newton_tol) or \
                    (rate is None and not self.scipy_newton and
                     (dy_norm < self.newton_tol or (self.crate_fresh and crate0 < 1.0 and dy_norm <= self.crate_dy_max and
                                                    crate0 / (1.0 - crate0) * dy_norm < self.newton_tol))):
                converged = True
                break
            dy_norm_old = dy_norm
        self.cur_crate = crate
        return converged, k + 1, y, d

    def select_order(self):
        """Order / step-size selection; called after the dense-output saves of the step."""
        if self.pending is None:
            return
        err_m, err_o, err_p, safety = self.pending
        self.pending = None
        norms = np.array([err_m, err_o, err_p])
        with np.errstate(divide="ignore"):
            factors = norms ** (-1 / np.arange(self.order, self.order + 3))
        delta = int(np.argmax(factors)) - 1
        self.order += delta
        factor = min(FIRST_MAX_FACTOR if self.first_selection else MAX_FACTOR, safety * np.max(factors))
        self.first_selection = False
        self.h_abs *= factor
        change_D(self.D, self.order, factor)
        self.n_equal = 0
        self.LU = None

    def interpolate(self, ts):
        """Dense output over the step that ended at self.t (Newton form on the backward differences)."""
        order, h = self.order, self.h_abs
        j = np.arange(order)
        x = (ts - (self.t - h * j)) / (h * (1 + j))
        p = np.cumprod(x)
        return self.D[0] + np.dot(self.D[1:order + 1].T, p)


class OracleRK45:
    """The explicit counterpart (kin_solve_explicit): SciPy's own RK45 (Dormand-Prince 5(4), `scipy.integrate.RK45`)
    behind the interface the driver below expects from an integrator - an independent implementation, so the
    device path is compared with it step for step. One RK45 object per segment (restart)."""

    def __init__(self, fun, n, atol, rtol):
        self.fun, self.n = fun, n
        self.set_tols(atol, rtol)
        self.stats = dict(n_steps=0, n_rejected=0, n_rhs=0, n_jac=0, n_factor=0, n_linsolve=0, n_newton_fail=0, n_restarts=0)
        self.iters_left = 0
        self.pre_attempt = None
        self.rk = None

    def set_tols(self, atol, rtol):
        self.atol, self.rtol = atol, rtol

    def _f(self, t, y):
        self.stats["n_rhs"] += 1
        return self.fun(y)

    def restart(self, t0, y0, t_bound):
        from scipy.integrate import RK45
        self.stats["n_restarts"] += 1
        y0 = np.array(y0, dtype=float)
        if not np.all(np.isfinite(self.fun(y0))):
            return False
        self.rk = RK45(self._f, t0, y0, t_bound, rtol=self.rtol, atol=self.atol)
        self.t, self.h_abs = t0, self.rk.h_abs
        self.D = [y0.copy()]
        return True

    def step(self, t_bound):
        self.iters_left -= 1          # (SciPy retries rejected attempts inside step(): counted per accepted step)
        if self.iters_left < 0:
            return "maxiters"
        msg = self.rk.step()
        if self.rk.status == "failed":
            return "dtmin"
        self.stats["n_steps"] += 1
        self.t = self.rk.t
        self.D = [self.rk.y.copy()]
        self._dense = self.rk.dense_output()
        return "ok"

    def select_order(self):
        pass

    def interpolate(self, ts):
        return self._dense(ts)


def solve_network_oracle(fun_of_k, jac_of_k, n, params, u0, k0=None, tstops=None, k_of_stop=None, k_of_time=None):
    """CPU restatement of the solve orchestration. `fun_of_k(k)(y)`, `jac_of_k(k)(y)`;
    `k_of_stop(i)` gives the rate vector in force from tstops[i] on (zero-order hold,
    solve_utils.jl:435-509). params: dict with the kin_params fields. Returns
    (t[M], u[M][n], retcode, stats)."""
    wall0 = time.time()
    tspan0, tspan1 = params["tspan"]
    abstol, reltol = params.get("abstol", 1e-10), params.get("reltol", 1e-8)
    chunks = params.get("solve_chunks", True)
    chunkstep = params.get("solve_chunkstep", 1e-3)
    save_interval = params.get("save_interval", None)
    maxiters = params.get("maxiters", 100000)
    adaptive_tols = params.get("adaptive_tols", True)
    variable = tstops is not None and len(tstops) > 0
    if not tspan0 < tspan1:
        raise ValueError("Invalid time span")
    n_chunks = 1
    if chunks:
        q = tspan1 / chunkstep
        if q != math.floor(q):
            raise ValueError("Simulation timespan is not divisible by requested chunkwise simulation step size")
        n_chunks = int(q)
        if save_interval is not None and save_interval > chunkstep:
            raise ValueError("Solution save interval must be less than chunkwise simulation step size")
    span_len = chunkstep if chunks else tspan1 - tspan0
    save_local = None
    if chunks or save_interval is not None:
        si = save_interval if save_interval is not None else chunkstep
        base, last = (0.0, chunkstep) if chunks else (tspan0, tspan1)
        cnt = int(math.floor(span_len / si + 1e-9)) + 1
        save_local = [min(base + i * si, last) for i in range(cnt)]
        if not chunks and save_local[-1] < last:
            save_local.append(last)
        # chunkwise: collect(0:save_interval:chunkstep) holds the chunk end only when it is a grid point
        # (methods.jl:756-758); see solver.cpp for the off-grid case
        if chunks and abs(save_local[-1] - last) <= 1e-9 * last:
            save_local[-1] = last
    L = len(save_local) if save_local is not None else 0
    save_hits_end = chunks and L > 0 and save_local[-1] == chunkstep

    state = {"k": None if k0 is None else np.array(k0, dtype=float)}
    if params.get("explicit", False):
        bdf = OracleRK45(lambda y: fun_of_k(state["k"])(y), n, abstol, reltol)
    else:
        # dtmin as the reference passes it: eps(solve_chunkstep) chunkwise (methods.jl:232, 770), eps(tspan[end]) otherwise
        # (methods.jl:164, 694)
        dtmin = params.get("dtmin", 0.0)
        if not dtmin > 0.0:
            dtmin = float(np.spacing(abs(chunkstep if chunks else tspan1)))
        bdf = OracleBDF(lambda y: fun_of_k(state["k"])(y), lambda y: jac_of_k(state["k"])(y), n, abstol, reltol,
                        dtmin=dtmin, ban_negatives=params.get("ban_negatives", False),
                        lu_band=params.get("lu_band", 0.35), lu_slots=params.get("lu_slots", 128))
    # continuous rate updates (methods.jl:363-653): k re-evaluated at the global time of every step attempt
    seg_origin = [0.0]
    if k_of_time is not None:
        def _hook(tau):
            state["k"] = k_of_time(seg_origin[0] + tau)
        bdf.pre_attempt = _hook
    out_t, out_u = [], []
    y = np.array(u0, dtype=float)
    next_stop = 0
    retcode = RET_SUCCESS
    n_retries = 0
    have_history, rates_changed, rates_in_force = False, False, -1
    # default: re-initialise at every segment start like the reference (reinit!, methods.jl:260, 819);
    # warm continuation is an opt-in experiment (see Solver::resume in solver.cpp)
    cold_restarts = bool(params.get("cold_restarts", True))
    for nc in range(n_chunks):
        t_start_g = chunkstep * nc if chunks else tspan0
        t_end_g = t_start_g + chunkstep if chunks else tspan1
        shift = nc * chunkstep if chunks else 0.0
        t_loc0, t_loc1 = (0.0, chunkstep) if chunks else (tspan0, tspan1)
        y_start = y.copy()
        n_out_start = len(out_t)
        attempts = 0
        if hasattr(bdf, "cache_suspended"):
            bdf.cache_suspended = False
        while True:
            attempts += 1
            if attempts > 1:
                have_history = False
            retcode = RET_SUCCESS
            bdf.iters_left = maxiters
            stop_i = next_stop
            while variable and stop_i < len(tstops) and tstops[stop_i] <= t_start_g:
                stop_i += 1
            if variable and max(stop_i - 1, 0) != rates_in_force:
                rates_in_force = max(stop_i - 1, 0)
                state["k"] = k_of_stop(rates_in_force)
                rates_changed = True
            failed = False
            save_i = 0
            if L > 0:
                out_t.append(save_local[0] + shift); out_u.append(y.copy()); save_i = 1
            else:
                out_t.append(t_loc0 + shift); out_u.append(y.copy())
            t_seg = t_loc0
            while t_seg < t_loc1 and not failed:
                seg_end, ends_at_stop = t_loc1, False
                if variable and stop_i < len(tstops) and tstops[stop_i] < t_end_g:
                    loc = tstops[stop_i] - shift
                    if loc < t_loc1:
                        seg_end, ends_at_stop = loc, True
                if seg_end > t_seg:
                    # segment-local time (see solver.cpp): restart at tau = 0, integrate to seg_len
                    seg_len = seg_end - t_seg
                    seg_origin[0] = t_seg + shift
                    if bdf.pre_attempt is not None:
                        bdf.pre_attempt(0.0)
                    if have_history and not cold_restarts:
                        bdf.resume(rates_changed)
                    elif not bdf.restart(0.0, y, seg_len):
                        retcode, failed = RET_UNSTABLE, True
                        break
                    have_history, rates_changed = True, False
                    while bdf.t < seg_len:
                        status = bdf.step(seg_len)
                        if status == "maxiters":
                            retcode, failed = RET_MAXITERS, True
                            break
                        if status == "dtmin":
                            retcode, failed = RET_DTMIN, True
                            break
                        if status == "unstable":
                            retcode, failed = RET_UNSTABLE, True
                            break
                        t_abs = seg_end if bdf.t >= seg_len else t_seg + bdf.t
                        if L > 0:
                            last_i = L - 1 if (chunks and not (nc == n_chunks - 1 and not save_hits_end)) else L
                            while save_i < last_i and save_local[save_i] <= t_abs:
                                out_t.append(save_local[save_i] + shift)
                                out_u.append(bdf.interpolate(min(save_local[save_i] - t_seg, bdf.t)))
                                save_i += 1
                        else:
                            out_t.append(t_abs + shift); out_u.append(bdf.D[0].copy())
                        bdf.select_order()
                    if failed:
                        break
                    y = bdf.D[0].copy()
                t_seg = seg_end
                if ends_at_stop:
                    state["k"] = k_of_stop(stop_i)
                    rates_in_force = stop_i
                    rates_changed = True
                    stop_i += 1
            if not failed:
                if chunks and nc == n_chunks - 1 and L > 1 and save_hits_end:
                    out_t.append(save_local[L - 1] + shift); out_u.append(y.copy())
                next_stop = stop_i
                break
            if (not adaptive_tols) or attempts >= 5 or abstol / 10 <= EPS or reltol / 10 <= EPS:
                break
            abstol /= 10; reltol /= 10
            rates_in_force = -1
            bdf.set_tols(abstol, reltol)
            n_retries += 1
            if hasattr(bdf, "cache_suspended"):
                bdf.invalidate_lu()
                bdf.cache_suspended = True
            y = np.maximum(y_start, 0.0)      # the rescue path clips inherited negative concentrations (solver.cpp, solve_entry)
            del out_t[n_out_start:]; del out_u[n_out_start:]
        if retcode != RET_SUCCESS:
            break
    stats = dict(bdf.stats)
    stats.update(n_retries=n_retries, final_abstol=abstol, final_reltol=reltol, wall_seconds=time.time() - wall0)
    return np.array(out_t), np.array(out_u).reshape(len(out_t), n), retcode, stats
