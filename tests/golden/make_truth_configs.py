"""Tight-tolerance truths for the BASELINE configurations' SOLVES (C3 static, C4 ramp, C5 variable conditions at 50k
species), generated in the build container by the compiled CPU baseline oracle/cpu_bdf.cpp (BDF + KLU-style sparse
LU) at tolerances 1000x tighter than the defaults (C3 long: 100x) the device path runs with. Outputs
truth_c3.npz, truth_c4.npz, truth_c5.npz are committed; the reference itself cannot produce them (Julia, no toolchain
in the image; its tests hold no trajectories - SURVEY 8(c)).

Each file also stores a second, even tighter (or looser) integration's deviation from the stored one in tolerance
units (`self_check`), so the reader knows how far the truth itself can be trusted.

    python tests/golden/make_truth_configs.py c3 c4 c5        (minutes to tens of minutes on 1 core each)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from kinetica_jl_amd.synth import synthetic_crn  # noqa: E402
from oracle import cpu_bdf  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ABSTOL, RELTOL = 1e-10, 1e-8      # the defaults of ODESimulationParams (params.jl:61-62): the unit of every deviation


def units(u, ref):
    return np.abs(u - ref) / (ABSTOL + RELTOL * np.abs(ref))


def solve(cs, pars, u0, tight, **kw):
    t0 = time.time()
    p = dict(pars, abstol=ABSTOL * tight, reltol=RELTOL * tight, dtmin=1e-300, adaptive_tols=False)
    t, u, rc, st = cs.solve(p, u0, **kw)
    assert rc == 0, (rc, st)
    print(f"   tolerance x{tight:g}: {st['n_steps']} steps, {st['n_factor']} factorisations, {time.time() - t0:.1f} s", flush=True)
    return t, u


def c3():
    """C3: 10k species / 50k reactions, static 1000 K, chunkwise defaults: the first 2 chunks, chunk ends saved."""
    net, Ea, A = synthetic_crn(10000, 50000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    pars = dict(tspan=(0.0, 2e-3), solve_chunks=True, solve_chunkstep=1e-3)
    t, u = solve(cs, pars, u0, 1e-3, k0=k)
    t2, u2 = solve(cs, pars, u0, 1e-2, k0=k)
    sc = float(units(u2, u).max())
    np.savez_compressed(os.path.join(HERE, "truth_c3.npz"), t=t, u=u, self_check=sc, T=1000.0)
    print("wrote truth_c3.npz", u.shape, "x1e-2 vs x1e-3:", sc)


def ramp_inputs(n, r, n_chunks):
    """C4 / C5: LinearGradientProfile(rate=50, 500 -> 1200 K), ts_update 1 ms, chunk 10 ms, save 5 ms: the first chunks."""
    net, Ea, A = synthetic_crn(n, r)
    tst = np.arange(0, 10 * n_chunks + 1) * 1e-3
    T = 500.0 + 50.0 * tst
    ks = orc.rate_table(Ea, A, T, k_max=1e12)
    u0 = np.zeros(n); u0[0] = 1.0
    pars = dict(tspan=(0.0, 1e-2 * n_chunks), solve_chunks=True, solve_chunkstep=1e-2, save_interval=5e-3)
    return net, tst, T, ks, u0, pars


def c4():
    net, tst, T, ks, u0, pars = ramp_inputs(10000, 50000, 3)
    cs = cpu_bdf.CpuSolver(net)
    t, u = solve(cs, pars, u0, 1e-3, tstops=tst, k_table=ks)
    t2, u2 = solve(cs, pars, u0, 1e-2, tstops=tst, k_table=ks)
    sc = float(units(u2, u).max())
    np.savez_compressed(os.path.join(HERE, "truth_c4.npz"), t=t, u=u, self_check=sc, tstops=tst, T_stops=T)
    print("wrote truth_c4.npz", u.shape, "x1e-2 vs x1e-3:", sc)


def c5():
    """C5: 50k species / 250k reactions under the ramp, 2 chunks (20 rate updates). Round 5: stored at x1e-3 tolerances and
    checked against x1e-2 (rounds 1-4: x1e-2 against x1e-1, self-check 23.7 units - the weakest pin of the suite); the x1e-2
    integration is kept next to it (`u_x1e2`) so that the self-check can be recomputed without re-running either. Saved every
    millisecond: the chunk ends (10, 20 ms) are what the device test compares (`t`, `u`), the state at the FIRST rate update
    (`t_early` = 1 ms, `u_early`) is what an integrator outside the BDF family can reach at this size in hours rather than days
    (make_truth_independent.py c5: 40 s per pair of SuperLU factorisations at 50k species)."""
    net, tst, T, ks, u0, pars = ramp_inputs(50000, 250000, 2)
    pars = dict(pars, save_interval=1e-3)
    cs = cpu_bdf.CpuSolver(net)
    t2, u2 = solve(cs, pars, u0, 1e-2, tstops=tst, k_table=ks)
    keep = [0, 10, 20]   # chunk ends only (3 x 50k doubles)
    assert np.allclose(t2[keep], [0.0, 1e-2, 2e-2], rtol=0, atol=1e-15) and abs(t2[1] - 1e-3) < 1e-15
    np.savez_compressed(os.path.join(HERE, "truth_c5_x1e2.partial.npz"), t=t2[keep], u=u2[keep], t_early=t2[1:2], u_early=u2[1:2])
    t, u = solve(cs, pars, u0, 1e-3, tstops=tst, k_table=ks)
    sc = float(units(u2, u).max())
    np.savez_compressed(os.path.join(HERE, "truth_c5.npz"), t=t[keep], u=u[keep], u_x1e2=u2[keep], t_early=t[1:2], u_early=u[1:2],
                        self_check=sc, tstops=tst, T_stops=T)
    os.remove(os.path.join(HERE, "truth_c5_x1e2.partial.npz"))
    print("wrote truth_c5.npz", u[keep].shape, "x1e-2 vs x1e-3:", sc)


def c3_long():
    """C3 over (0, 0.1) s = 100 default chunks: every 10th chunk end is stored. Tolerances x1e-2 (checked
    against x1e-1): at x1e-3 = rtol 1e-11 the integration sits on the rounding floor of the right-hand side and did not finish
    in four hours on one core of the build container (30 chunks at x1e-3 took 39 minutes, c3_mid; DESIGN 4.0 has the floor)."""
    net, Ea, A = synthetic_crn(10000, 50000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    pars = dict(tspan=(0.0, 0.1), solve_chunks=True, solve_chunkstep=1e-3)
    t, u = solve(cs, pars, u0, 1e-2, k0=k)
    t2, u2 = solve(cs, pars, u0, 1e-1, k0=k)
    keep = list(range(0, 101, 10))
    sc = float(units(u2[keep], u[keep]).max())
    np.savez_compressed(os.path.join(HERE, "truth_c3_long.npz"), t=t[keep], u=u[keep], self_check=sc, T=1000.0, keep=np.array(keep))
    print("wrote truth_c3_long.npz", u[keep].shape, "x1e-1 vs x1e-2:", sc)


def c3_mid():
    """C3 over (0, 0.03) s = 30 default chunks (a shorter stand-in for c3_long on slow hosts): every 5th chunk end is stored."""
    net, Ea, A = synthetic_crn(10000, 50000)
    k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
    u0 = np.zeros(10000); u0[0] = 1.0
    cs = cpu_bdf.CpuSolver(net)
    pars = dict(tspan=(0.0, 0.03), solve_chunks=True, solve_chunkstep=1e-3)
    t, u = solve(cs, pars, u0, 1e-3, k0=k)
    t2, u2 = solve(cs, pars, u0, 1e-2, k0=k)
    keep = list(range(0, 31, 5))
    sc = float(units(u2[keep], u[keep]).max())
    np.savez_compressed(os.path.join(HERE, "truth_c3_mid.npz"), t=t[keep], u=u[keep], self_check=sc, T=1000.0, keep=np.array(keep))
    print("wrote truth_c3_mid.npz", u[keep].shape, "x1e-2 vs x1e-3:", sc)


def c4_long():
    """C4 ramp, the first 20 chunks (0.2 s, 200 rate updates / restarts): chunk ends are stored."""
    net, tst, T, ks, u0, pars = ramp_inputs(10000, 50000, 20)
    cs = cpu_bdf.CpuSolver(net)
    t, u = solve(cs, pars, u0, 1e-3, tstops=tst, k_table=ks)
    t2, u2 = solve(cs, pars, u0, 1e-2, tstops=tst, k_table=ks)
    keep = list(range(0, len(t), 2))          # save_interval 5 ms: every second saved point is a chunk end
    sc = float(units(u2[keep], u[keep]).max())
    np.savez_compressed(os.path.join(HERE, "truth_c4_long.npz"), t=t[keep], u=u[keep], self_check=sc, tstops=tst, T_stops=T)
    print("wrote truth_c4_long.npz", u[keep].shape, "x1e-2 vs x1e-3:", sc)


if __name__ == "__main__":
    for name in sys.argv[1:] or ["c3", "c4", "c5"]:
        print(name, flush=True)
        {"c3": c3, "c4": c4, "c5": c5, "c3_long": c3_long, "c4_long": c4_long, "c3_mid": c3_mid}[name]()
