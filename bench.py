#!/usr/bin/env python
"""Benchmark of the kinetic-ODE solve path on MI355X (BASELINE.json metric:
"RHS evals/sec + wall-clock per solve_network, 10k-species CRN").

A "step" is one batched RHS sweep: B states of the 10k-species / 50k-reaction synthetic CRN
(SURVEY.md 8(d), C3/C4), each state with its own rate-constant vector (ensemble of
temperatures), all inputs resident in HBM before the timed region. `value` = RHS evaluations
per second over all ranks. The same JSON line carries
  roofline      - algorithmic bytes of one sweep (M2: 20R + B(8R + 16N)) / HIP-event time of the sweep kernels,
  cpu_baseline  - the CPU oracle's RHS (plain C, 1 core) on the same CRN, bounded sample,
  solve_network - wall-clock of kin_solve (implicit BDF, on-device sparse LU) on the same CRN
                  next to the CPU oracle's BDF + SuperLU on a bounded number of chunks.
Multi-GPU (weak scaling): every rank sweeps its own B states (ensemble replicas);
no data-path collective, only the barrier / max-over-ranks timing of the contract.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device memory, streams, torch.distributed: plumbing only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--species", type=int, default=10000)
    ap.add_argument("--reactions", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=4096, help="states per sweep per GPU")
    ap.add_argument("--solve-chunks", type=int, default=20, help="chunks of the timed kin_solve (0 = skip)")
    ap.add_argument("--cpu-solve-chunks", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # rehearsal knobs (a one-GPU box can run 2 ranks on the same card over gloo): never set by the driver
    if os.environ.get("BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from kinetica_jl_amd import capi
    from kinetica_jl_amd.synth import synthetic_crn
    assert capi.lib().kin_set_device(local_rank) == 0

    N, R, B = args.species, args.reactions, args.batch
    net, Ea, A = synthetic_crn(N, R)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    k1000 = h.rates_at(1000.0)

    # ---- resident inputs: u[B][N] LogUniform(1e-12, 1), k[B][R] = Arrhenius at B temperatures 500..1200 K
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev); g.manual_seed(12345 + rank)
    d_u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12.0 - 12.0)
    T = torch.linspace(500.0, 1200.0, B, dtype=torch.float64, device=dev)
    dEa = torch.tensor(Ea, dtype=torch.float64, device=dev)[None, :]
    dA = torch.tensor(A, dtype=torch.float64, device=dev)[None, :]
    kr = dA * torch.exp(-dEa / (8.314462618 * T[:, None])) * 6.02214076e23
    d_k = (1.0 / ((1.0 / 1e12) + (1.0 / kr))).contiguous()
    del kr
    d_du = torch.empty((B, N), dtype=torch.float64, device=dev)
    # a dedicated (non-null) stream: the kernels are launched on it and timed on it with events
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    def sweep():
        h.rhs_batched_dev(B, d_u.data_ptr(), d_k.data_ptr(), d_du.data_ptr(), stream)

    for _ in range(args.warmup):
        sweep()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        sweep()
        b.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    from kinetica_jl_amd.distributed import max_over_ranks
    elapsed = max_over_ranks(elapsed, dist, dev if (dist is None or dist.get_backend() == "nccl") else "cpu")
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    out = None
    if rank == 0:
        alg_bytes = 20 * R + B * (8 * R + 16 * N)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (separate rocprofv3 runs of this same
        # command, tools/collect_profiles.sh), only when they were taken on this configuration
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_sweep_pmc.json")))
            if pm["algorithmic_bytes_per_launch"] == alg_bytes:
                traffic = pm["hbm_bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "RHS evals/sec (batched sweep) + wall-clock per solve_network, 10k-species CRN",
            "value": world * B * args.steps / elapsed, "unit": "RHS evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic CRN {N} species / {R} reactions (seed 12345), batched RHS sweep, "
                                   f"B={B} states per GPU with per-state Arrhenius k (500-1200 K)",
                       "states_per_gpu": B, "parallelism": f"replicas x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "kernel": "kin::sweep_reg_kernel<8, 4, BLK> (state fits LDS, reactions paired with their reverses; else kin::sweep_lds_kernel / sweep_big_kernel)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernel_ms},
        }

    # ---- single-state RHS latency (what the integrator sees), rank 0 only
    if rank == 0:
        u1 = 10.0 ** np.random.default_rng(0).uniform(-12, 0, N)
        h.set_rates(k1000)
        h.rhs(u1)
        t1 = time.perf_counter()
        for _ in range(200):
            h.rhs(u1)
        out["single_state_rhs_us_host_roundtrip"] = (time.perf_counter() - t1) / 200 * 1e6

    # ---- wall-clock per solve_network (C3: static 1000 K, chunkwise, defaults of params.jl:55-75). Every rank solves
    # its own replica (rank r at 1000 + 10 r K: independent trajectories); the reported wall-clock is
    # the maximum over ranks, rank 0's statistics are printed
    if args.solve_chunks > 0:
        u0 = np.zeros(N); u0[0] = 1.0
        chunk = 1e-3
        p = capi.KinParams(tspan0=0.0, tspan1=chunk * args.solve_chunks, abstol=1e-10, reltol=1e-8, adaptive_tols=1,
                           update_tols=0, solve_chunks=1, ban_negatives=0, solve_chunkstep=chunk, maxiters=100000,
                           save_interval=-1.0)
        if rank > 0:
            h.set_rates(h.rates_at(1000.0 + 10.0 * rank))
        h.solve(p, u0)     # warm-up: symbolic analysis + allocations
        if dist:
            dist.barrier()
        t1 = time.perf_counter()
        ts, us, rc, st, status = h.solve(p, u0)
        gpu_wall_local = time.perf_counter() - t1
        gpu_wall = max_over_ranks(gpu_wall_local, dist, dev if (dist is None or dist.get_backend() == "nccl") else "cpu")
    if rank == 0 and args.solve_chunks > 0:
        out["solve_network"] = {"workload": f"StaticODESolve, T=1000 K (+10 K per rank), tspan (0, {p.tspan1:g}) s, solve_chunkstep 1e-3 "
                                            f"({args.solve_chunks} chunks), abstol 1e-10, reltol 1e-8, one replica per GPU",
                                "gpu_wall_s": gpu_wall, "gpu_s_per_chunk": gpu_wall / args.solve_chunks,
                                "replicas": world, "solves_per_s": world / gpu_wall, "retcode": rc, "stats": st}
        if not args.no_cpu and args.cpu_solve_chunks > 0 and world == 1:   # CPU legs: rank 0 at N=1 only
            from oracle import bdf as obdf
            from oracle import oracle as orc
            on = orc.OracleNetwork.from_flat(net)
            pars = dict(tspan=(0.0, chunk * args.cpu_solve_chunks), solve_chunks=True, solve_chunkstep=chunk)
            t1 = time.perf_counter()
            to, uo, rco, sto = obdf.solve_network_oracle(lambda kk: (lambda y: on.rhs(kk, y)),
                                                         lambda kk: (lambda y: on.jac(kk, y)), N, pars, u0, k0=k1000)
            cpu_wall = time.perf_counter() - t1
            nck = args.cpu_solve_chunks
            dev = np.abs(us[:nck + 1] - uo) / (1e-10 + 1e-8 * np.abs(uo))
            dev_vs_cpu = float(dev.max())
            # the integrator controls the RMS norm over the N species (so a single species may sit sqrt(N) x
            # further out than the norm): report the controlled quantity next to the per-species maximum
            dev_rms = float(np.sqrt((dev ** 2).mean(axis=1)).max())
            out["solve_network"].update({"cpu_wall_s": cpu_wall, "cpu_chunks": nck, "cpu_s_per_chunk": cpu_wall / nck,
                                         "cpu_kind": "port (oracle BDF + SuperLU, 1 core)",
                                         "speedup_per_chunk": (cpu_wall / nck) / (gpu_wall / args.solve_chunks),
                                         "max_dev_vs_cpu_in_tol_units": dev_vs_cpu,
                                         "rms_dev_vs_cpu_in_tol_units": dev_rms, "cpu_stats": sto})

    # ---- CPU baseline for the headline metric: oracle RHS, 1 core, bounded sample
    if rank == 0 and not args.no_cpu and world == 1:
        from oracle import oracle as orc
        on = orc.OracleNetwork.from_flat(net)
        u1 = 10.0 ** np.random.default_rng(0).uniform(-12, 0, N)
        on.rhs(k1000, u1)
        n_eval, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            for _ in range(50):
                on.rhs(k1000, u1)
            n_eval += 50
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n_eval / dt, "unit": "RHS evals/s", "cores": 1, "kind": "port",
                               "sample": f"{n_eval} single-state evaluations of the same {N}/{R} CRN by oracle/kin_oracle.c "
                                         f"(scalar C, -O2) in {dt:.1f} s"}
        # the same evaluation over an ensemble of states, one state per OpenMP thread (every host core the
        # process may use): what the CPU can do for the batched sweep; the reference itself is single-threaded
        cores = orc.usable_cores()
        Ub = 10.0 ** np.random.default_rng(1).uniform(-12, 0, (8 * cores, N))
        on.rhs_many(k1000, Ub, cores)
        n_eval, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 5.0:
            on.rhs_many(k1000, Ub, cores)
            n_eval += Ub.shape[0]
        dt = time.perf_counter() - t1
        out["cpu_baseline_all_cores"] = {"value": n_eval / dt, "unit": "RHS evals/s", "cores": cores, "kind": "port",
                                         "sample": f"{n_eval} evaluations, {Ub.shape[0]} states per call, OpenMP over states, "
                                                   f"{dt:.1f} s"}
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    h.close()


if __name__ == "__main__":
    main()
