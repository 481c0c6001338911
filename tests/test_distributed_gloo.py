"""CPU tests of the N>1 path: world-size-2 gloo processes exercising the partitioning,
gathering and max-over-ranks logic bench.py and the ensemble driver use (no GPU needed)."""
import os
import socket

import numpy as np
import pytest
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
        # (1) bench contract: step time = MAX over ranks
        t = D.max_over_ranks(1.0 + rank, dist)
        # (2) rate table rows sharded over time stops, gathered in stop order
        T = np.linspace(500.0, 1200.0, 11)
        table = D.rate_table_sharded(lambda Ts: np.outer(Ts, [1.0, 2.0, 3.0]), T, dist)
        # (3) ensemble of 5 independent solves, round-robin over ranks
        res = D.solve_ensemble(list(range(5)), lambda m: {"member": m, "rank": dist.get_rank(), "umax": m * 10.0}, dist)
        q.put((rank, t, table, res))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    T = np.linspace(500.0, 1200.0, 11)
    for rank, t, table, res in out:
        assert t == 2.0                                               # max(1.0, 2.0)
        np.testing.assert_allclose(table, np.outer(T, [1.0, 2.0, 3.0]))
        assert [r["member"] for r in res] == [0, 1, 2, 3, 4]
        assert [r["rank"] for r in res] == [0, 1, 0, 1, 0]            # i % world


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 14001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert D.max_over_ranks(3.5) == 3.5
    assert D.solve_ensemble([1, 2, 3], lambda m: m * m) == [1, 4, 9]
