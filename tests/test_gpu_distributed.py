"""GPU test of the N > 1 path with real HIP calls under two ranks: two processes share the one card of the test box
(gloo backend: RCCL refuses two ranks on one device; the collectives then stage through the host, the HIP side is the
same code that runs over RCCL on a multi-GPU node). Covers the three partitions of kinetica_jl_amd.distributed:
rate-table slices, replicas + gathered per-species maxima, reaction blocks of one right-hand side."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from kinetica_jl_amd import capi
    from kinetica_jl_amd import distributed as D
    from kinetica_jl_amd.synth import synthetic_crn
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        net, Ea, A = synthetic_crn(300, 1500)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        # (1) rate table: 11 stops -> slices of 6 and 5 rows generated on the device, gathered
        T = np.linspace(500.0, 1200.0, 11)
        table = D.rate_table_sharded(h, T, dist, dev).cpu().numpy()
        lo, hi, mine = D.rate_table_sharded(h, T, dist, dev, gather=False)
        # (2) replicas: rank r solves at 1000 + 50 r K; per-species maxima gathered from device buffers
        h.rates_at(1000.0 + 50.0 * rank)
        u0 = np.zeros(300); u0[0] = 1.0
        p = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                           ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
        t, u, rc, st, _ = h.solve(p, u0)
        umax = D.gather_solution_max(h, dist, dev)
        ens = D.solve_ensemble([1000.0, 1050.0, 1100.0],
                               lambda Tm: (h.rates_at(Tm), h.solve(p, u0), h.solution_max())[2], dist)
        # (2b) an ensemble of the ONE network sharded by members: 5 members -> blocks of 3 and 2, one kin_solve_ensemble call per rank
        Tm = 1000.0 + 50.0 * np.arange(5)
        U0 = np.tile(u0, (5, 1))
        ens_rows, ens_rcs = D.solve_ensemble_sharded(h, p, U0, T=Tm, dist=dist, device=dev)
        # (3) one right-hand side, reactions split over the ranks, partial du summed
        k = h.rates_at(1000.0)
        uu = 10.0 ** np.random.default_rng(0).uniform(-12, 0, 300)
        d_u = torch.tensor(uu, dtype=torch.float64, device=dev)
        d_du = torch.empty_like(d_u)
        D.rhs_reaction_blocks(h, d_u, d_du, dist)
        torch.cuda.synchronize()
        timing = D.time_rhs_reaction_blocks(h, d_u, dist, reps=20)
        q.put((rank, table, (lo, hi, tuple(mine.shape)), rc, u.max(axis=0), umax, np.array(ens), d_du.cpu().numpy(), k, uu, timing, ens_rows, ens_rcs))
        h.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_card():
    from kinetica_jl_amd.synth import synthetic_crn
    from oracle import oracle as orc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    net, Ea, A = synthetic_crn(300, 1500)
    on = orc.OracleNetwork.from_flat(net)
    T = np.linspace(500.0, 1200.0, 11)
    ref_table = orc.rate_table(Ea, A, T, k_max=1e12)
    own_max = [o[4] for o in out]
    for rank, table, (lo, hi, shp), rc, umax_own, umax_all, ens, du, k, uu, timing, ens_rows, ens_rcs in out:
        assert rc == 0
        assert ens_rows.shape == (5, 300) and (ens_rcs == 0).all()
        np.testing.assert_array_equal(ens_rows, out[0][11])          # the same table on every rank
        np.testing.assert_array_equal(ens_rows[0], own_max[0])       # member 0 (1000 K) = rank 0's solo replica, bit for bit
        np.testing.assert_array_equal(ens_rows[1], own_max[1])       # member 1 (1050 K) = rank 1's
        assert (lo, hi) == ((0, 6), (6, 11))[rank] and shp == (hi - lo, 1500)
        np.testing.assert_allclose(table, ref_table, rtol=1e-11)
        assert umax_all.shape == (2, 300)
        np.testing.assert_array_equal(umax_all[0], own_max[0])
        np.testing.assert_array_equal(umax_all[1], own_max[1])
        assert ens.shape == (3, 300)
        np.testing.assert_array_equal(ens[0], own_max[0])            # member 0 = rank 0's replica at 1000 K
        np.testing.assert_array_equal(ens[1], own_max[1])            # member 1 = rank 1's at 1050 K
        ref = on.rhs(k, uu)
        assert np.max(np.abs(du - ref) / (on.abs_rhs(k, uu) + 1e-300)) < 1e-13
        assert timing["ranks"] == 2 and timing["split_rhs_plus_allreduce_us"] > 0


def test_bench_two_ranks_on_one_card_end_to_end():
    """`python bench.py --gpus 2` as the driver would start it (self-launch, one process per rank), rehearsed on the one
    card of the test box over gloo: the JSON line is the last line of stdout and carries the contract's fields for N = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--species", "1000", "--reactions", "5000",
                        "--batch", "256", "--steps", "3", "--warmup", "1", "--solve-chunks", "2", "--no-cpu", "--no-pmc",
                        "--sustain-seconds", "0", "--replicas", ""], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["unit"] == "RHS evals/s" and rec["dtype"] == "f64"
    assert rec["solve_network"]["replicas"] == 2 and rec["solve_network"]["retcode"] == 0
    assert rec["single_trajectory_rhs_allreduce"]["ranks"] == 2
    ens = rec["ensemble_sharded_by_members"]       # 64 members per rank, one kin_solve_ensemble launch each, rows all-gathered
    assert ens["members"] == 128 and ens["all_ok"] and ens["rows_gathered"] == [128, 300] and ens["solves_per_s"] > 0
    # a world size that contradicts --gpus is refused before any GPU work
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-pmc"],
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_bench_four_ranks_rehearsal_reports_who_took_part():
    """The N > 2 case of the driver's scaling run, rehearsed with FOUR ranks on the one card (gloo; the box allows at most
    six GPU processes, so the N = 8 case itself is the driver's): the JSON line names backend, world size and the device of
    every rank, value is the whole-job aggregate, every rank solved its own replica."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--species", "1000", "--reactions", "5000",
                        "--batch", "128", "--steps", "3", "--warmup", "1", "--solve-chunks", "2", "--no-cpu", "--no-pmc",
                        "--sustain-seconds", "0", "--spinup-seconds", "0", "--replicas", ""], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 4 and rec["scaling"] == "weak" and rec["value"] > 0
    who = rec["ranks"]
    assert who["backend"] == "gloo" and who["world_size"] == 4 and sorted(m["rank"] for m in who["members"]) == [0, 1, 2, 3]
    assert rec["solve_network"]["replicas"] == 4 and rec["solve_network"]["retcode"] == 0
    assert rec["single_trajectory_rhs_allreduce"]["ranks"] == 4
    assert rec["ensemble_sharded_by_members"]["members"] == 256 and rec["ensemble_sharded_by_members"]["all_ok"]
    assert list(rec)[-1] == "roofline" and list(rec)[-2] == "solve_network_summary"      # what the driver keeps ends the line
    assert rec["solve_network_summary"]["gpu_wall_s"] > 0


def test_bench_five_ranks_rehearsal():
    """As many ranks as the test box allows next to the test process itself (six GPU processes per card: this process + five
    ranks; the driver's N = 8 run needs a node): the same checks at world size 5."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "5", "--species", "1000", "--reactions", "5000",
                        "--batch", "64", "--steps", "3", "--warmup", "1", "--solve-chunks", "2", "--no-cpu", "--no-pmc", "--no-tiled",
                        "--sustain-seconds", "0", "--spinup-seconds", "0", "--replicas", ""], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 5 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["ranks"]["world_size"] == 5 and sorted(m["rank"] for m in rec["ranks"]["members"]) == list(range(5))
    assert rec["solve_network"]["replicas"] == 5 and rec["solve_network"]["retcode"] == 0
    assert rec["ensemble_sharded_by_members"]["members"] == 320 and rec["ensemble_sharded_by_members"]["all_ok"]
