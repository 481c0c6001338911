"""Independent cross-check of the committed C3 / C4 truths (tests/golden/truth_c3.npz, truth_c4.npz).

Those truths come from oracle/cpu_bdf.cpp - the same BDF family as the device integrator, at 1000x tighter tolerances.
A semantic error shared by both (restart rule, zero-order hold of the rate constants, chunk stitching) would be invisible
in a comparison between them. This script integrates the same problems with an integrator that shares nothing with
them - SciPy's Radau IIA (order 5, own step control, SuperLU) on the oracle's RHS and analytic sparse Jacobian, straight
through [0, t_end] for the static case and segment by segment between rate updates for the ramp (no chunking, no BDF
history, no LU cache) - and stores its deviation from the committed truth, in units of the default tolerances
(abstol 1e-10 + reltol 1e-8 |u|), as `self_check_independent` inside the truth files. The tests assert it.

    python tests/golden/make_truth_independent.py c3      (35 minutes on one core; INDEP_TIGHT=1e-2: 5 minutes, 0.16 units)
    python tests/golden/make_truth_independent.py c4      (25 minutes)
    python tests/golden/make_truth_independent.py c3_mid c4_long c5     (round 5: the longer truths; an hour or more each)
"""
import os
import sys
import time

import numpy as np
from scipy.integrate import Radau, solve_ivp
from scipy.sparse.linalg import splu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ABSTOL, RELTOL = 1e-10, 1e-8
# tolerances of the independent run relative to the defaults: 1e-3 = what the stored truths were made with (C4); the static
# C3 problem at 1000 K runs into the rounding floor of the right-hand side at that setting (step sizes of 1e-7 s late in the
# run: hours) and uses 1e-2 (INDEP_TIGHT in the environment overrides)
TIGHT = float(os.environ.get("INDEP_TIGHT", "1e-3"))


def units(u, ref):
    return np.abs(u - ref) / (ABSTOL + RELTOL * np.abs(ref))


class RadauMMD(Radau):
    """SciPy's Radau with SuperLU's minimum-degree ordering on A' + A instead of the default COLAMD: on these Jacobians
    (a few hub species with thousands of neighbours) COLAMD fills 32 M entries and takes 22 + 51 s per real + complex
    factorisation at 10k species, MMD_AT_PLUS_A 1.6 M entries and 0.8 + 0.9 s. Integrator and step control untouched."""

    def __init__(self, *a, newton_tol_override=None, **kw):
        super().__init__(*a, **kw)
        if newton_tol_override is not None:
            self.newton_tol = newton_tol_override

        def lu(A):
            self.nlu += 1
            return splu(A, permc_spec="MMD_AT_PLUS_A")
        self.lu = lu


def radau(on, k, u0, t0, t1, t_eval, newton_tol=None):
    """newton_tol: SciPy's Radau stops its simplified Newton iteration at max(10 eps / rtol, min(0.03, sqrt(rtol))) of the
    error weight = 3e-6 at rtol 1e-11. Near this problem's fast equilibria the right-hand side is a difference of fluxes
    ~1e10 times larger than itself, and h x (its rounding) exceeds that tolerance once h > ~1e-7 s: the static C3 run
    then crawls at h ~ 1e-8 (measured: t = 7.1e-5 -> 7.3e-5 s in 135 s and 170 factorisations). 0.03 - the value ode15s
    and CVODE use, and SciPy's own cap - lets it through (2 ms in 4 minutes); what it leaves unconverged is < 0.03 of a
    weight that is itself 100-1000x below a default tolerance unit."""
    n = [0]

    def f(t, u):
        n[0] += 1
        return on.rhs(k, u)

    sol = solve_ivp(f, (t0, t1), u0, method=RadauMMD, jac=lambda t, u: on.jac(k, u).tocsc(), rtol=RELTOL * TIGHT, atol=ABSTOL * TIGHT,
                    t_eval=t_eval, first_step=1e-22, **({} if newton_tol is None else {"newton_tol_override": newton_tol}))
    assert sol.success, sol.message
    return sol.y.T, sol.nfev, sol.njev, sol.nlu


def update(path, **fields):
    d = dict(np.load(path))
    d.update(fields)
    np.savez_compressed(path, **d)


def c3():
    path = os.path.join(HERE, "truth_c3.npz")
    tr = np.load(path)
    net, Ea, A = synthetic_crn(10000, 50000)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, float(tr["T"]), k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    t0 = time.time()
    t_eval = tr["t"][1:]
    u, nfev, njev, nlu = radau(on, k, u0, 0.0, float(tr["t"][-1]), t_eval, newton_tol=0.03)
    dev = units(u, tr["u"][1:])
    print(f"c3: Radau {nfev} rhs, {njev} jac, {nlu} lu, {time.time() - t0:.0f} s; max deviation from the stored truth {dev.max():.3f} units, "
          f"rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           independent_method=f"scipy Radau (SuperLU, MMD ordering), oracle rhs + analytic sparse Jacobian, rtol {RELTOL * TIGHT:g}, "
                              f"atol {ABSTOL * TIGHT:g}, corrector tolerance 0.03, no chunking")


def c4():
    """First chunk of the ramp (10 ms, 10 rate updates, saves at 0 / 5 / 10 ms): piecewise-constant k, one Radau
    integration per interval between rate updates."""
    path = os.path.join(HERE, "truth_c4.npz")
    tr = np.load(path)
    net, Ea, A = synthetic_crn(10000, 50000)
    on = orc.OracleNetwork.from_flat(net)
    tst, T = tr["tstops"], tr["T_stops"]
    u = np.zeros(10000); u[0] = 1.0
    saves = {}
    t0 = time.time()
    for i in range(10):                                     # [tst[i], tst[i+1]) with the rates of stop i
        k = orc.arrhenius(Ea, A, float(T[i]), k_max=1e12)
        # segment-local time: the first steps after a rate switch are ~1e-20 s, below the resolution of t = 5e-3
        y, nfev, njev, nlu = radau(on, k, u, 0.0, float(tst[i + 1] - tst[i]), [float(tst[i + 1] - tst[i])])
        u = y[-1]
        print(f"   segment {i}: {nfev} rhs, {nlu} lu, {time.time() - t0:.0f} s", flush=True)
        for ts in (5e-3, 1e-2):
            if abs(tst[i + 1] - ts) < 1e-12:
                saves[ts] = u.copy()
    idx = {5e-3: int(np.argmin(np.abs(tr["t"] - 5e-3))), 1e-2: int(np.argmin(np.abs(tr["t"] - 1e-2)))}
    dev = np.stack([units(saves[ts], tr["u"][idx[ts]]) for ts in (5e-3, 1e-2)])
    print(f"c4: max deviation from the stored truth {dev.max():.3f} units, rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           independent_method=f"scipy Radau (SuperLU, MMD ordering) per rate interval (zero-order hold), first chunk (saves at 5 and "
                              f"10 ms), rtol {RELTOL * TIGHT:g}, atol {ABSTOL * TIGHT:g}")


def c3_mid():
    """truth_c3_mid.npz (30 default chunks, every 5th chunk end stored): the save points at 5 and 10 ms by ONE Radau integration
    from t = 0 - no chunking, no BDF history, no LU cache. Tolerances x1e-2 (INDEP_TIGHT overrides): see c3()."""
    path = os.path.join(HERE, "truth_c3_mid.npz")
    tr = np.load(path)
    tight = float(os.environ.get("INDEP_TIGHT", "1e-2"))
    global TIGHT
    TIGHT = tight
    net, Ea, A = synthetic_crn(10000, 50000)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, float(tr["T"]), k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    t0 = time.time()
    n_pts = int(os.environ.get("INDEP_POINTS", "2"))
    t_eval = tr["t"][1:1 + n_pts]
    u, nfev, njev, nlu = radau(on, k, u0, 0.0, float(t_eval[-1]), t_eval, newton_tol=0.03)
    dev = units(u, tr["u"][1:1 + n_pts])
    print(f"c3_mid: Radau {nfev} rhs, {njev} jac, {nlu} lu, {time.time() - t0:.0f} s; deviation from the stored truth at t = {list(t_eval)}: "
          f"max {dev.max():.3f} units, rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           independent_points=np.asarray(t_eval),
           independent_method=f"scipy Radau (SuperLU, MMD ordering), oracle rhs + analytic sparse Jacobian, rtol {RELTOL * TIGHT:g}, "
                              f"atol {ABSTOL * TIGHT:g}, corrector tolerance 0.03, no chunking")


def ramp_segments(path, n, r, n_seg, check_times, label):
    """Zero-order hold of the rate constants, one Radau integration per interval between rate updates, from t = 0 through
    `n_seg` intervals; deviation from the stored truth at `check_times`."""
    tr = np.load(path)
    net, Ea, A = synthetic_crn(n, r)
    on = orc.OracleNetwork.from_flat(net)
    tst, T = tr["tstops"], tr["T_stops"]
    u = np.zeros(n); u[0] = 1.0
    saves = {}
    t0 = time.time()
    for i in range(n_seg):
        k = orc.arrhenius(Ea, A, float(T[i]), k_max=1e12)
        y, nfev, njev, nlu = radau(on, k, u, 0.0, float(tst[i + 1] - tst[i]), [float(tst[i + 1] - tst[i])])
        u = y[-1]
        print(f"   {label} segment {i}: {nfev} rhs, {nlu} lu, {time.time() - t0:.0f} s", flush=True)
        for ts in check_times:
            if abs(tst[i + 1] - ts) < 1e-12:
                saves[ts] = u.copy()
    if os.environ.get("INDEP_DUMP"):      # the Radau states themselves (to be compared with a truth that is regenerated meanwhile)
        np.savez_compressed(os.environ["INDEP_DUMP"], t=np.asarray(check_times), u=np.stack([saves[ts] for ts in check_times]))
    dev = np.stack([units(saves[ts], tr["u"][int(np.argmin(np.abs(tr["t"] - ts)))]) for ts in check_times])
    print(f"{label}: max deviation from the stored truth at t = {list(check_times)}: {dev.max():.3f} units, rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           independent_points=np.asarray(check_times),
           independent_method=f"scipy Radau (SuperLU, MMD ordering) per rate interval (zero-order hold), {n_seg} intervals from t = 0, "
                              f"rtol {RELTOL * TIGHT:g}, atol {ABSTOL * TIGHT:g}")


def c4_long():
    """truth_c4_long.npz (20 chunks of the ramp, chunk ends stored): the first two chunk ends (10 and 20 ms, 20 rate intervals)."""
    ramp_segments(os.path.join(HERE, "truth_c4_long.npz"), 10000, 50000, 20, (1e-2, 2e-2), "c4_long")


def c5():
    """truth_c5.npz (50k species, 2 chunks of the ramp): the state at the FIRST rate update (1 ms: the interval that starts from the
    pure initial state with steps of 1e-22 s and holds most of the work; `u_early` of the truth file). A SuperLU factorisation
    pair of Radau's takes 40 s at this size: one interval is ~2.5 hours, the ten of the first chunk would be a day."""
    path = os.path.join(HERE, "truth_c5.npz")
    tr = np.load(path)
    net, Ea, A = synthetic_crn(50000, 250000)
    on = orc.OracleNetwork.from_flat(net)
    u0 = np.zeros(50000); u0[0] = 1.0
    k = orc.arrhenius(Ea, A, float(tr["T_stops"][0]), k_max=1e12)
    t0 = time.time()
    y, nfev, njev, nlu = radau(on, k, u0, 0.0, float(tr["tstops"][1]), [float(tr["tstops"][1])])
    print(f"c5: Radau over the first rate interval: {nfev} rhs, {nlu} lu, {time.time() - t0:.0f} s", flush=True)
    if os.environ.get("INDEP_DUMP"):
        np.savez_compressed(os.environ["INDEP_DUMP"], t=np.asarray([float(tr["tstops"][1])]), u=y[-1:])
    if "u_early" not in tr.files:
        print("truth_c5.npz has no u_early yet (regenerate it with make_truth_configs.py c5); the Radau state is in INDEP_DUMP", flush=True)
        return
    dev = units(y[-1], tr["u_early"][0])
    print(f"c5: deviation from the stored truth at t = 1 ms: max {dev.max():.3f} units, rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           independent_points=np.asarray([float(tr["tstops"][1])]),
           independent_method=f"scipy Radau (SuperLU, MMD ordering) over the first rate interval, rtol {RELTOL * TIGHT:g}, atol {ABSTOL * TIGHT:g}")


def c3_long():
    """truth_c3_long.npz (100 chunks at x1e-2 tolerances) is the same trajectory as truth_c3_mid.npz (30 chunks at x1e-3), which
    c3_mid() checks against Radau at 5 and 10 ms: stored here are its deviation from the mid truth at their common save points
    (10, 20, 30 ms) and, at 10 ms, the bound on its deviation from the Radau state that the triangle inequality gives
    (|long - mid| + |mid - Radau|, maxima over the species). No integration is run."""
    path = os.path.join(HERE, "truth_c3_long.npz")
    lg, md = np.load(path), np.load(os.path.join(HERE, "truth_c3_mid.npz"))
    assert "self_check_independent" in md.files, "run c3_mid first"
    common = [t for t in lg["t"] if t > 0 and np.abs(md["t"] - t).min() < 1e-12]
    dev = [float(units(lg["u"][int(np.argmin(np.abs(lg["t"] - t)))], md["u"][int(np.argmin(np.abs(md["t"] - t)))]).max()) for t in common]
    at10 = dev[0] + float(md["self_check_independent"])
    print(f"c3_long: against truth_c3_mid at t = {common}: {dev} units; at 10 ms at most {at10:.2f} units from the Radau state", flush=True)
    update(path, self_check_vs_c3_mid=np.asarray(dev), self_check_vs_c3_mid_points=np.asarray(common), self_check_independent=at10,
           independent_method="bound at t = 10 ms: |truth_c3_long - truth_c3_mid| + |truth_c3_mid - scipy Radau| (make_truth_independent.py c3_mid), maxima over the species")


def c3_long_direct():
    """truth_c3_long.npz against ONE Radau integration from t = 0 to 0.1 s (tolerances x1e-2, corrector tolerance 0.03: see c3()),
    at every stored chunk end - the direct form of what c3_long() bounds through truth_c3_mid. Hours on one core."""
    path = os.path.join(HERE, "truth_c3_long.npz")
    tr = np.load(path)
    global TIGHT
    TIGHT = float(os.environ.get("INDEP_TIGHT", "1e-2"))
    net, Ea, A = synthetic_crn(10000, 50000)
    on = orc.OracleNetwork.from_flat(net)
    k = orc.arrhenius(Ea, A, float(tr["T"]), k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    t0 = time.time()
    t_eval = tr["t"][1:]
    u, nfev, njev, nlu = radau(on, k, u0, 0.0, float(t_eval[-1]), t_eval, newton_tol=0.03)
    dev = units(u, tr["u"][1:])
    per = [float(x) for x in dev.max(axis=1)]
    print(f"c3_long: Radau {nfev} rhs, {njev} jac, {nlu} lu, {time.time() - t0:.0f} s; deviation from the stored truth per chunk end {per}, rms {np.sqrt((dev ** 2).mean()):.4f}", flush=True)
    if os.environ.get("INDEP_DUMP"):
        np.savez_compressed(os.environ["INDEP_DUMP"], t=t_eval, u=u)
    update(path, self_check_independent=float(dev.max()), self_check_independent_rms=float(np.sqrt((dev ** 2).mean())),
           self_check_independent_per_point=np.asarray(per), independent_points=np.asarray(t_eval),
           independent_method=f"scipy Radau (SuperLU, MMD ordering), oracle rhs + analytic sparse Jacobian, rtol {RELTOL * TIGHT:g}, "
                              f"atol {ABSTOL * TIGHT:g}, corrector tolerance 0.03, one integration from t = 0 over all 100 ms")


if __name__ == "__main__":
    for name in sys.argv[1:] or ["c3", "c4"]:
        {"c3": c3, "c4": c4, "c3_mid": c3_mid, "c4_long": c4_long, "c5": c5, "c3_long": c3_long, "c3_long_direct": c3_long_direct}[name]()
