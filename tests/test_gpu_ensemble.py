"""Several handles driven concurrently by several host threads on ONE device (ensembles are the path's
natural parallelism, docs/src/tutorials/ode-solution.md:190): every replica's result equals what the same handle
computes alone, bit for bit, and matches the oracle's integration of the same problem."""
import threading

import numpy as np
import pytest

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _pars(t1, chunk=1e-3):
    return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                          ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=-1.0, dtmin=0.0)


def test_four_replicas_on_one_device_match_their_solo_runs_and_the_oracle():
    from oracle import cpu_bdf
    n, r, K = 1000, 5000, 4
    net, Ea, A = synthetic_crn(n, r)
    temps = [900.0 + 100.0 * i for i in range(K)]
    u0 = np.zeros(n); u0[0] = 1.0
    hs = [capi.HipNetwork.from_flat(net) for _ in range(K)]
    ks = [orc.arrhenius(Ea, A, T, k_max=1e12) for T in temps]
    for h, k in zip(hs, ks):
        h.set_rates(k)
    solo = [h.solve(_pars(3e-3), u0) for h in hs]                       # one after the other
    out = [None] * K
    start = threading.Barrier(K)

    def work(i):
        start.wait()
        for _ in range(3):                                              # several solves per thread: the calls interleave
            out[i] = hs[i].solve(_pars(3e-3), u0)

    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(K):
        t, u, rc, st, status = out[i]
        assert rc == 0 and status == 0
        assert np.array_equal(t, solo[i][0]) and np.array_equal(u, solo[i][1])     # concurrency changes nothing
        assert st["n_steps"] == solo[i][3]["n_steps"]
    # against the oracle's own integration (same algorithm on the CPU): the documented trajectory bound
    cs = cpu_bdf.CpuSolver(net)
    for i in (0, K - 1):
        tc, uc, rcc, stc = cs.solve(dict(tspan=(0.0, 3e-3), solve_chunks=True, solve_chunkstep=1e-3), u0, k0=ks[i])
        e = np.abs(out[i][1] - uc) / (1e-10 + 1e-8 * np.abs(uc))
        assert rcc == 0 and e.max() <= 100.0, e.max()
    # the batched sweep of one handle next to a solve of another (different kernels sharing the device)
    big = threading.Thread(target=lambda: hs[0].solve(_pars(3e-3), u0))
    big.start()
    U = 10.0 ** np.random.default_rng(0).uniform(-12, 0, (64, n))
    dU = hs[1].rhs_batched(U)
    big.join()
    on = orc.OracleNetwork.from_flat(net)
    for b in (0, 63):
        assert (np.abs(dU[b] - on.rhs(ks[1], U[b])) / (on.abs_rhs(ks[1], U[b]) + 1e-300)).max() < 1e-13
    [h.close() for h in hs]
