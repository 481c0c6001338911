"""CPU tests of the N > 1 path: world-size-2 gloo processes. Every rank loads libkinetica_hip.so through the C ABI (on
a box without a GPU the library must answer KIN_ERR_DEVICE - there is no CPU fallback to hide behind) and runs the
collectives of kinetica_jl_amd.distributed on tensors: uneven row all-gather (rate-table slices), tensor-based
ensemble gather, max-over-ranks. The same helpers run on device tensors over RCCL on a multi-GPU node and with two
ranks on one card in tests/test_gpu_distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kinetica_jl_amd import distributed as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kinetica_jl_amd import capi
        # (0) the product library is what each rank loads; without a device it refuses to compute
        L = capi.lib()
        assert all(hasattr(L, s) for s in ("kin_solution_max_dev", "kin_rate_table_dev", "kin_rhs_block_dev"))
        status = None
        if capi.device_count() == 0:
            try:
                capi.HipNetwork(2, [0, 1], [0], [1], [0, 1], [1], [1])
            except capi.KineticaHipError as e:
                status = e.code
        # (1) bench contract: step time = MAX over ranks
        t = D.max_over_ranks(1.0 + rank, dist)
        # (2) rate-table slices of uneven length (11 stops over 2 ranks: 6 + 5), gathered in stop order as tensors
        T = np.linspace(500.0, 1200.0, 11)
        lo, hi = D.shard_range(len(T), rank, world)
        mine = torch.tensor(np.outer(T[lo:hi], [1.0, 2.0, 3.0]))
        table = D.all_gather_rows(mine, dist).numpy()
        # (3) ensemble of 5 independent "solves", round-robin over ranks, results as float64 vectors
        res = D.solve_ensemble(list(range(5)), lambda m: np.array([m, dist.get_rank(), m * 10.0]), dist)
        # (4) an ensemble of ONE network sharded by members (7 members over 2 ranks: 4 + 3): every rank runs ONE ensemble call on
        # its block (here a stand-in with the C ABI's return shape: the product call needs a device), rows come back in member order
        def fake_ensemble(h, pars, u0s, Tm, km):
            Kb, Nn = u0s.shape
            u = np.stack([np.stack([u0s[i] * (1.0 + j) + Tm[i] for j in range(3)]) for i in range(Kb)])      # [Kb][3 rows][N]
            return np.arange(3.0), u, np.full(Kb, 3), (Tm > 1250.0).astype(np.int32), None
        u0s = np.arange(7.0)[:, None] + np.zeros((7, 4))
        Tm = 1000.0 + 50.0 * np.arange(7)
        rows_max, rcs = D.solve_ensemble_sharded(None, None, u0s, T=Tm, dist=dist, solve_fn=fake_ensemble)
        rows_fin, _ = D.solve_ensemble_sharded(None, None, u0s, T=Tm, dist=dist, reduce="final", solve_fn=fake_ensemble)
        q.put((rank, t, table, np.array(res), status, rows_max, rows_fin, rcs))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = np.linspace(500.0, 1200.0, 11)
    from kinetica_jl_amd import capi
    for rank, t, table, res, status, rows_max, rows_fin, rcs in out:
        u0s = np.arange(7.0)[:, None] + np.zeros((7, 4))
        Tm = 1000.0 + 50.0 * np.arange(7)
        np.testing.assert_array_equal(rows_max, u0s * 3.0 + Tm[:, None])          # the largest of the three rows, member order
        np.testing.assert_array_equal(rows_fin, u0s * 3.0 + Tm[:, None])
        np.testing.assert_array_equal(rcs, [0, 0, 0, 0, 0, 0, 1])
        assert t == 2.0                                               # max(1.0, 2.0)
        np.testing.assert_allclose(table, np.outer(T, [1.0, 2.0, 3.0]))
        np.testing.assert_array_equal(res[:, 0], [0, 1, 2, 3, 4])
        np.testing.assert_array_equal(res[:, 1], [0, 1, 0, 1, 0])     # i % world
        np.testing.assert_array_equal(res[:, 2], [0, 10, 20, 30, 40])
        if capi.device_count() == 0:
            assert status == capi.KIN_ERR_DEVICE


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 14001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.max_over_ranks(3.5) == 3.5
    r = D.solve_ensemble([1, 2, 3], lambda m: np.array([m * m]))
    assert [float(x[0]) for x in r] == [1.0, 4.0, 9.0]


def test_bench_self_launch_is_wired_before_any_gpu_call():
    """bench.py --gpus N without a launcher must start its own ranks before touching the GPU: the launch branch sits in
    front of `import torch` and of every capi call."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("self_launch(args)") < body.index("import torch") < body.index("torch.cuda.set_device")
    assert body.index("measure_traffic(args)") < body.index("torch.cuda.set_device")
