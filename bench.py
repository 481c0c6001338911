#!/usr/bin/env python
"""Benchmark of the kinetic-ODE solve path on MI355X (BASELINE.json metric:
"RHS evals/sec + wall-clock per solve_network, 10k-species CRN").

A "step" is one batched RHS sweep: B states of the 10k-species / 50k-reaction synthetic CRN
(SURVEY.md 8(d), C3/C4), each state with its own rate-constant vector (ensemble of
temperatures), all inputs resident in HBM before the timed region. `value` = RHS evaluations
per second over all ranks. The same JSON line carries
  roofline      - algorithmic bytes of one sweep (M2: 20R + B(8R + 16N)) / HIP-event time of the sweep kernel; `traffic`
                  is MEASURED IN THIS RUN: rank 0 (N = 1) starts two child processes under `rocprofv3 --pmc` (FETCH_SIZE
                  and WRITE_SIZE cannot share a pass on gfx950) that launch the same sweep, before this process touches
                  the GPU; null when the profiler is unavailable (never read from a committed file);
  cpu_baseline  - the CPU oracle's RHS (plain C, 1 core) on the same CRN, bounded sample,
  solve_network - wall-clock of kin_solve (implicit BDF, on-device sparse LU) on the same CRN, and the compiled CPU
                  baseline (oracle/cpu_bdf.cpp: the same BDF with a KLU-style sparse LU, -O3) on a bounded number of
                  chunks with the device timed on THE SAME chunks; 1 core (the reference's solve path is single-threaded)
                  and all cores (one replica per core, the CPU counterpart of one replica per GPU).
Multi-GPU (weak scaling): every rank sweeps its own B states and solves its own replica; no data-path
collective in the timed region, the contract's barrier / max-over-ranks timing only; the replicas' per-species maxima
(what identify_next_seeds reads) are gathered with an RCCL all-gather on device buffers after the solve.
`python bench.py --gpus N` without a launcher starts its own N ranks (before any GPU call).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--species", type=int, default=10000)
    ap.add_argument("--reactions", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=4096, help="states per sweep per GPU")
    ap.add_argument("--solve-chunks", type=int, default=100, help="chunks of the timed kin_solve (0 = skip)")
    ap.add_argument("--cpu-solve-chunks", type=int, default=2, help="chunks the CPU baseline solves (the device is timed on the same ones)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="back-to-back sweeps after the timed region (sustained clock)")
    ap.add_argument("--spinup-seconds", type=float, default=1.0,
                    help="back-to-back sweeps BEFORE the warm-up steps, untimed: the clocks settle (0 = cold-clock number)")
    ap.add_argument("--replicas", default="1,2,4,8", help="concurrent replicas on one GPU to time (comma list, '' = skip)")
    ap.add_argument("--no-tiled", dest="tiled", action="store_false", help="skip the library-order sweep legs (C3 + C5)")
    ap.add_argument("--no-configs", dest="configs", action="store_false",
                    help="skip the BASELINE configurations as SURVEY 8(d) states them (C3 whole span both ways, C4 20-chunk prefix, C5 5 chunks)")
    ap.add_argument("--no-crossover", dest="crossover", action="store_false", help="skip the network-size crossover table and the ensemble launch")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--replica-worker", type=int, nargs=3, default=None, help=argparse.SUPPRESS)   # threads, chunks, first replica index
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N without torchrun
# ---------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """Starts one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env) before this process has made any
    GPU call, relays rank 0's output and exits with the worst child status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    # every rank's stderr goes to a file of its own (printed when a rank fails): a rank that dies no longer takes its
    # reason with it, and the ranks' messages do not interleave on the launcher's stderr
    logdir = tempfile.mkdtemp(prefix="kin_bench_ranks_", dir="/tmp")
    logs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        logs.append(open(os.path.join(logdir, f"rank{r}.stderr"), "w"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=logs[-1]))
    # a rank that dies (e.g. fewer GPUs than ranks) leaves the others waiting at the rendezvous: watch the children and,
    # when one exits with an error, end the rest (these exact child processes) instead of hanging until an outer timeout
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        states = [p.poll() for p in procs]
        failed = [c for c in states if c not in (None, 0)]
        if failed:
            rc = failed[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            print(f"bench.py: a rank exited with status {rc}; the other ranks were stopped", file=sys.stderr)
            for r, f in enumerate(logs):
                f.flush()
                tail = open(f.name).read()[-2000:]
                if tail.strip():
                    print(f"---- rank {r} stderr (tail) ----\n{tail}", file=sys.stderr)
            break
        if all(c == 0 for c in states):
            break
        time.sleep(0.2)
    reader.join(timeout=5)
    for f in logs:
        f.close()
    if rc == 0:
        shutil.rmtree(logdir, ignore_errors=True)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    raise SystemExit(rc)


# ---------------------------------------------------------------------------------------------------------------
# HBM traffic of the sweep kernel from the PMC counters, measured by child processes of this run
# ---------------------------------------------------------------------------------------------------------------
def under_profiler():
    """True when this process was itself started under rocprofv3 / rocprof (its tool library is preloaded and has
    initialised the GPU): starting another profiler from here would exec through processes that hold the GPU."""
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_")) for k in os.environ)


def clean_child_env():
    """Environment of the profiler children: nothing of an outer profiler (LD_PRELOAD, ROCP* / ROCPROF*) is handed down."""
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_"))}
    env["TMPDIR"] = "/tmp"
    return env


def measure_traffic(args):
    """(bytes per launch, note). FETCH_SIZE and WRITE_SIZE in separate passes; units are KiB; FETCH_SIZE counts half of a
    wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM section) - doubled here."""
    if under_profiler():
        return None, "not measured: this process already runs under a profiler (no nested rocprofv3)"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="kin_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--species", str(args.species), "--reactions", str(args.reactions), "--batch", str(args.batch),
                   "--steps", "4", "--warmup", "1"]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=clean_child_env(), stdout=subprocess.PIPE,
                                   stderr=subprocess.STDOUT, timeout=300)
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f"rocprofv3 --pmc {ctr}: {type(e).__name__}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} gave no counters (rc {p.returncode})"
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                 if "kin::sweep_" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if not v:
                return None, f"no kin::sweep_ dispatch in the {ctr} pass"
            vals[ctr] = sum(v) / len(v)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, \
        "rocprofv3 --pmc child runs of this bench.py: (2 x FETCH_SIZE + WRITE_SIZE) KiB, averaged over the sweep launches"


def replica_worker(args):
    """One worker process of the ensemble leg: T threads, one handle each. Protocol on stdin / stdout: READY after the
    untimed first solves, GO from the parent, then one JSON line with the worker's own wall-clock and a checksum."""
    import threading
    from kinetica_jl_amd import capi
    from kinetica_jl_amd.synth import synthetic_crn
    T, nck, base = args.replica_worker
    N, R = args.species, args.reactions
    net, Ea, A = synthetic_crn(N, R)
    u0 = np.zeros(N); u0[0] = 1.0
    pars = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nck, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                          ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
    hs = [capi.HipNetwork.from_flat(net) for _ in range(T)]
    for i, h in enumerate(hs):
        h.set_arrhenius(Ea, A, k_max=1e12)
        h.rates_at(1000.0 + 10.0 * (base + i))
    outs = [None] * T
    gate = threading.Barrier(T + 1)

    def work(i):
        hs[i].solve(pars, u0)             # untimed: symbolic analysis + allocations
        gate.wait()
        gate.wait()
        outs[i] = hs[i].solve(pars, u0)
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for x in th:
        x.start()
    gate.wait()
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    gate.wait()
    for x in th:
        x.join()
    wall = time.perf_counter() - t0
    print(json.dumps({"wall_s": wall, "ok": all(o[2] == 0 for o in outs), "steps": [o[3]["n_steps"] for o in outs],
                      "sum0": float(outs[0][1][-1].sum()), "max0": float(outs[0][1][-1].max())}), flush=True)
    for h in hs:
        h.close()


def run_replica_workers(args, P, T, nck):
    """Starts P worker processes with T threads each, releases them together, returns the ensemble's throughput."""
    cmd = [sys.executable, os.path.abspath(__file__), "--species", str(args.species), "--reactions", str(args.reactions)]
    logdir = tempfile.mkdtemp(prefix="kin_bench_workers_", dir="/tmp")
    logs = [open(os.path.join(logdir, f"worker{p}.stderr"), "w") for p in range(P)]
    procs = [subprocess.Popen(cmd + ["--replica-worker", str(T), str(nck), str(p * T)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              stderr=logs[p], text=True, env=clean_child_env()) for p in range(P)]

    def tail(p):
        logs[p].flush()
        return open(logs[p].name).read()[-1500:]
    try:
        for pr in procs:
            line = pr.stdout.readline()
            while line and line.strip() != "READY":
                line = pr.stdout.readline()
            if not line:
                raise RuntimeError("a replica worker ended before it was ready: " + tail(procs.index(pr)))
        t0 = time.perf_counter()
        for pr in procs:
            pr.stdin.write("GO\n"); pr.stdin.flush()
        reps = []
        for i, pr in enumerate(procs):
            line = pr.stdout.readline()
            if not line.strip():
                raise RuntimeError("a replica worker ended without a result: " + tail(i))
            reps.append(json.loads(line))
        wall = time.perf_counter() - t0
    finally:
        for pr in procs:
            try:
                pr.wait(timeout=60)
            except subprocess.TimeoutExpired:
                pr.kill()
        for f in logs:
            f.close()
        shutil.rmtree(logdir, ignore_errors=True)
    K = P * T
    return {"processes": P, "threads_per_process": T, "wall_s": wall, "solves_per_s": K / wall, "all_success": all(r["ok"] for r in reps),
            "replica0_steps": reps[0]["steps"][0], "replica0_checksum": [reps[0]["sum0"], reps[0]["max0"]]}


def guarded(out, key, fn):
    """Optional legs never take the headline numbers with them: an exception becomes {'error': ...} under `key`."""
    try:
        out[key] = fn()
    except Exception as e:        # noqa: BLE001 - whatever an optional leg throws is recorded, not raised
        import traceback
        out[key] = {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().splitlines()[-3:]}


def main():
    args = parse()
    if args.replica_worker:
        return replica_worker(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.pmc_child:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not args.pmc_child:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    import torch  # device memory, streams, torch.distributed: plumbing only (importing it makes no GPU call)

    traffic, traffic_note = None, "not measured (--no-pmc, the counter-collecting child itself, or N > 1)"
    if rank == 0 and world == 1 and not args.no_pmc and not args.pmc_child:
        traffic, traffic_note = measure_traffic(args)      # child processes; this process has not touched the GPU yet

    # rehearsal knobs (a one-GPU box can run 2 ranks on the same card over gloo): never set by the driver
    if os.environ.get("BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        backend = os.environ.get("BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        tmo = datetime.timedelta(seconds=int(os.environ.get("BENCH_RENDEZVOUS_TIMEOUT", "180")))   # a missing rank ends the run
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    from kinetica_jl_amd import capi
    from kinetica_jl_amd.distributed import gather_solution_max, max_over_ranks
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

    # Clock spin-up (untimed, before the W warm-up steps): the chip's power management raises the clocks only after tens
    # of milliseconds of load - the first 20 launches after an idle period run at 0.48-0.49 ms, the next 20 at 0.43, then
    # 0.42 (tools/sweep_time.py) - and W = 5 warm-up steps of 0.45 ms are over long before that. The timed region is
    # unchanged (exactly K steps between barriers); --spinup-seconds 0 gives the cold-clock number.
    if args.spinup_seconds > 0 and not args.pmc_child:
        t_end = time.perf_counter() + args.spinup_seconds
        while time.perf_counter() < t_end:
            for _ in range(50):
                sweep()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        sweep()
    torch.cuda.synchronize()
    if args.pmc_child:                       # under rocprofv3 --pmc: the sweep launches only
        for _ in range(args.steps):
            sweep()
        torch.cuda.synchronize()
        h.close()
        return
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
    coll_dev = dev if (dist is None or dist.get_backend() == "nccl") else "cpu"
    elapsed = max_over_ranks(elapsed, dist, coll_dev)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # ---- the same sweep back to back for a few seconds (outside the timed region): the rate the chip sustains once its
    # clock management has settled, and enough GPU-resident time for an outside observer to see the device busy
    sustained_ms = None
    if args.sustain_seconds > 0:
        n_sus = max(10, int(args.sustain_seconds / (kernel_ms * 1e-3)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_sus):
            sweep()
        e1.record()
        torch.cuda.synchronize()
        sustained_ms = e0.elapsed_time(e1) / n_sus

    # who took part (so that the driver can see that RCCL had N ranks on N devices): backend, world size, device of every rank
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
          "pci_bus_id": getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "pci_bus_id", None)}
    if dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        rank_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "members": gathered,
                     "scaling_note": "weak scaling: every rank sweeps its own B states; no scaling curve has been measured before round 4"}
    else:
        rank_info = {"backend": None, "world_size": 1, "members": [me]}
    out = None
    if rank == 0:
        alg_bytes = 20 * R + B * (8 * R + 16 * N)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "RHS evals/sec (batched sweep) + wall-clock per solve_network, 10k-species CRN",
            "value": world * B * args.steps / elapsed, "unit": "RHS evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "ranks": rank_info,
            "config": {"workload": f"synthetic CRN {N} species / {R} reactions (seed 12345), batched RHS sweep, "
                                   f"B={B} states per GPU with per-state Arrhenius k (500-1200 K)",
                       "states_per_gpu": B, "parallelism": f"replicas x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_over_algorithmic": None if traffic is None else traffic / alg_bytes,
                         "kernel": "kin::sweep_reg_kernel<8, 4, BLK> (state fits LDS, reactions paired with their reverses; else kin::sweep_gen_kernel / sweep_big_kernel)",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kernel_ms,
                         "spinup_s": args.spinup_seconds,
                         "sustained_launch_ms": sustained_ms,
                         "sustained_frac": None if not sustained_ms else alg_bytes / (sustained_ms * 1e-3) / 8e12},
        }

    # ---- the sweep in the library's own data layout (kin_rhs_tiled_dev): the same C3 states with k written in library
    # order by the rate-table kernel and with NO k at all (rate constants formed inside the sweep from T[b], SURVEY M1'),
    # and the C5 network (50k species / 250k reactions, state too large for LDS: hubs + windows), rank 0 at N = 1
    if rank == 0 and world == 1 and args.tiled:
        def tiled_leg(hh, Nn, Rr, Bb, uu, TT):
            lay = hh.lib_layout()
            kl = torch.empty((Bb, lay["k_len"]), dtype=torch.float64, device=dev)
            hh.rate_table_lib_dev(TT.cpu().numpy(), kl.data_ptr())
            dd = torch.empty_like(uu)

            def ev(fn, reps=10):
                fn(); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record(); torch.cuda.synchronize()
                return a.elapsed_time(b) / reps
            ms_k = ev(lambda: hh.rhs_tiled_dev(Bb, uu.data_ptr(), dd.data_ptr(), d_k_lib=kl.data_ptr(), stream=stream))
            ms_T = ev(lambda: hh.rhs_tiled_dev(Bb, uu.data_ptr(), dd.data_ptr(), d_T=TT.data_ptr(), stream=stream))
            alg_k = 20 * Rr + Bb * (8 * Rr + 16 * Nn)
            alg_T = 36 * Rr + Bb * (16 * Nn + 8)
            # the drop-in entry: states in the CALLER's species order (the reference's sol.u layout), rate constants in the slot order
            # the library's own rate table writes (kin_rhs_batched_klib_dev); and the plain kin_rhs_batched_dev with the rate
            # constants in the caller's REACTION order next to it (kin::sweep_big_kernel where the state does not fit LDS)
            ms_d = ev(lambda: hh.rhs_batched_klib_dev(Bb, uu.data_ptr(), kl.data_ptr(), dd.data_ptr(), stream))
            kc = torch.rand((Bb, Rr), dtype=torch.float64, device=dev, generator=g) + 0.5
            ms_c = ev(lambda: hh.rhs_batched_dev(Bb, uu.data_ptr(), kc.data_ptr(), dd.data_ptr(), stream))
            del kl, dd, kc
            return {"species": Nn, "reactions": Rr, "states": Bb, "hubs": lay["hubs"], "windows": lay["windows"],
                    "species_order_is_callers": lay["identity"],
                    "dropin_callers_species_order_k_in_slot_order": {
                        "entry": "kin_rhs_batched_klib_dev", "ms": ms_d, "evals_per_s": Bb / (ms_d * 1e-3), "frac_of_8TBps": alg_k / (ms_d * 1e-3) / 8e12,
                        "hbm_bytes_expected_over_algorithmic": 1.0 if lay["identity"] else (alg_k + Bb * 32 * Nn) / alg_k},
                    "callers_species_and_reaction_order": {"entry": "kin_rhs_batched_dev", "ms": ms_c, "frac_of_8TBps": alg_k / (ms_c * 1e-3) / 8e12},
                    "k_stream": {"ms": ms_k, "evals_per_s": Bb / (ms_k * 1e-3), "algorithmic_bytes_M2": alg_k,
                                 "GBps": alg_k / (ms_k * 1e-3) / 1e9, "frac_of_8TBps": alg_k / (ms_k * 1e-3) / 8e12, "bound": "hbm / LDS atomics"},
                    "temperature_form": {"ms": ms_T, "evals_per_s": Bb / (ms_T * 1e-3), "algorithmic_bytes_M1prime": alg_T,
                                         "GBps": alg_T / (ms_T * 1e-3) / 1e9, "bound": "FP64 VALU (2 exp per record) + LDS atomics"}}
        def tiled_all():
            res = {"kernel": "kin::tiled_sweep_kernel (library order: kin_lib_layout / kin_rate_table_lib_dev / kin_rhs_tiled_dev)",
                   "C3": tiled_leg(h, N, R, B, d_u, T)}
            net5, Ea5, A5 = synthetic_crn(50000, 250000)
            h5 = capi.HipNetwork.from_flat(net5)
            try:
                h5.set_arrhenius(Ea5, A5, k_max=1e12)
                B5 = 1024
                u5 = torch.pow(10.0, torch.rand((B5, 50000), dtype=torch.float64, device=dev, generator=g) * 12.0 - 12.0)
                res["C5"] = tiled_leg(h5, 50000, 250000, B5, u5, torch.linspace(500.0, 1200.0, B5, dtype=torch.float64, device=dev))
                res["C5"]["traffic_note"] = "PMC counters of this kernel: profiles/r03_c5_tiled_pmc.json (tools/pmc_tiled.sh)"
                del u5
            finally:
                h5.close()
            return res
        guarded(out, "tiled_sweep", tiled_all)

        def post_cutoff_leg():
            # what apply_low_k_cutoff! (solve_utils.jl:213-245) leaves behind: the C3 CRN without a random 30 % of its reactions
            # - pairs broken, the records without a reverse at one rate-constant slot in library order. Algorithmic bytes of the
            # network that is left; the caller's layouts (kin::sweep_gen_kernel) and the library order next to each other.
            keep = np.sort(np.random.default_rng(0).choice(R, int(0.7 * R), replace=False))
            net_p = net.subset(keep)
            Rp = net_p.n_reactions
            hp = capi.HipNetwork.from_flat(net_p)
            try:
                hp.set_arrhenius(Ea[keep], A[keep], k_max=1e12)
                lay = hp.lib_layout()
                k_p = torch.rand((B, Rp), dtype=torch.float64, device=dev, generator=g) + 0.5
                du_p = torch.empty_like(d_u)
                kl_p = torch.empty((B, lay["k_len"]), dtype=torch.float64, device=dev)
                torch.cuda.synchronize()
                hp.rate_table_lib_dev(T.cpu().numpy(), kl_p.data_ptr())
                alg = 20 * Rp + B * (8 * Rp + 16 * N)

                def ev(fn, reps=10):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(reps):
                        fn()
                    b.record(); torch.cuda.synchronize()
                    return a.elapsed_time(b) / reps
                ms_c = ev(lambda: hp.rhs_batched_dev(B, d_u.data_ptr(), k_p.data_ptr(), du_p.data_ptr(), stream))
                # (the states in the library's species order: the caller's at this size)
                ms_l = ev(lambda: hp.rhs_batched_klib_dev(B, d_u.data_ptr(), kl_p.data_ptr(), du_p.data_ptr(), stream))
                del k_p, du_p, kl_p
                return {"network": "C3 CRN minus a random 30 % of its reactions", "species": N, "reactions": Rp, "states": B,
                        "records": lay["records"], "k_len": lay["k_len"], "algorithmic_GB": alg / 1e9,
                        "callers_layouts": {"kernel": "kin::sweep_gen_kernel", "ms": ms_c, "frac_of_8TBps": alg / (ms_c * 1e-3) / 8e12},
                        "library_order": {"kernel": "kin::tiled_sweep_kernel, one-slot records, through kin_rhs_batched_klib_dev (states in the caller's order, k in slot order)", "ms": ms_l,
                                          "frac_of_8TBps": alg / (ms_l * 1e-3) / 8e12}}
            finally:
                hp.close()
        guarded(out, "post_cutoff_sweep", post_cutoff_leg)

    # ---- ONE trajectory's RHS with the reactions split over the ranks and an all-reduce of du (N doubles):
    # measured at N > 1 so that the cost of the single-trajectory decomposition is a number, not an argument
    if world > 1:
        from kinetica_jl_amd.distributed import time_rhs_reaction_blocks
        h.set_rates(k1000)
        split = time_rhs_reaction_blocks(h, d_u[0].clone(), dist, reps=200)
        if rank == 0:
            out["single_trajectory_rhs_allreduce"] = dict(split, note="reaction blocks of one RHS over the ranks + SUM all-reduce of "
                                                          "du (RCCL) against the whole RHS on one rank: microseconds per evaluation")

    # ---- an ensemble of ONE small network sharded by members over the ranks: 64 members per rank (weak scaling), each
    # rank ONE kin_solve_ensemble launch on its GPU, then one row per member (the per-species maxima identify_next_seeds reads) and
    # the return codes all-gathered in member order (RCCL on device tensors). No data-path collective during the solves.
    if world > 1 and args.solve_chunks > 0:
        def sharded_ensemble_leg():
            from kinetica_jl_amd.distributed import solve_ensemble_sharded
            net_e, Ea_e, A_e = synthetic_crn(300, 1500)
            he = capi.HipNetwork.from_flat(net_e)
            try:
                he.set_arrhenius(Ea_e, A_e, k_max=1e12)
                Ke = 64 * world
                u0e = np.zeros((Ke, 300)); u0e[:, 0] = 1.0
                Te = np.linspace(900.0, 1300.0, Ke)
                pe = capi.KinParams(tspan0=0.0, tspan1=2e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1,
                                    ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
                def local(h_, p_, u_, T_, k_):
                    # a rank whose launch throws still takes part in the all-gather (retcode -1 for its members): the others
                    # must not be left waiting in a collective
                    try:
                        return h_.solve_ensemble(p_, u_, T=T_, k=k_)
                    except Exception:      # noqa: BLE001
                        return None, np.zeros((len(u_), 1, 300)), np.ones(len(u_), np.int64), -np.ones(len(u_), np.int32), None
                solve_ensemble_sharded(he, pe, u0e, T=Te, dist=dist, device=dev, solve_fn=local)       # warm-up (symbolic analysis, allocations)
                dist.barrier()
                t1 = time.perf_counter()
                rows, rcs = solve_ensemble_sharded(he, pe, u0e, T=Te, dist=dist, device=dev, solve_fn=local)
                w = max_over_ranks(time.perf_counter() - t1, dist, coll_dev)
                return {"network": "300 species / 1500 reactions", "members": Ke, "members_per_rank": 64, "wall_s": w, "solves_per_s": Ke / w,
                        "all_ok": bool((rcs == 0).all()), "rows_gathered": list(rows.shape),
                        "note": "rank r integrates members [64 r, 64 r + 64) in one kin_solve_ensemble launch; maxima + return codes all-gathered"}
            finally:
                he.close()
        # (every rank must reach the same collectives: a member launch that throws is absorbed inside `local` above; what is caught
        # here is a failure every rank meets alike - a refused allocation, a shape the gather rejects - so that the line is still printed)
        try:
            leg = sharded_ensemble_leg()
        except Exception as e:      # noqa: BLE001
            leg = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["ensemble_sharded_by_members"] = leg

    # ---- single-state RHS latency (what the integrator sees), rank 0 only
    if rank == 0:
        u1 = 10.0 ** np.random.default_rng(0).uniform(-12, 0, N)
        h.set_rates(k1000)
        h.rhs(u1)
        t1 = time.perf_counter()
        for _ in range(200):
            h.rhs(u1)
        out["single_state_rhs_us_host_roundtrip"] = (time.perf_counter() - t1) / 200 * 1e6
        # the same evaluation on device-resident buffers, back to back (what the integrator pays per RHS: two dependent
        # launches, no copies, no synchronisation in between)
        d_u1 = torch.tensor(u1, dtype=torch.float64, device=dev)
        d_du1 = torch.empty_like(d_u1)
        for _ in range(10):
            h.rhs_block_dev(0, R, d_u1.data_ptr(), d_du1.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2000):
            h.rhs_block_dev(0, R, d_u1.data_ptr(), d_du1.data_ptr(), stream)
        torch.cuda.synchronize()
        out["single_state_rhs_us_device_resident"] = (time.perf_counter() - t1) / 2000 * 1e6

    # ---- wall-clock per solve_network (C3: static 1000 K, chunkwise, defaults of params.jl:55-75). Every rank solves
    # its own replica (rank r at 1000 + 10 r K: independent trajectories); the reported wall-clock is
    # the maximum over ranks, rank 0's statistics are printed; the replicas' per-species maxima (identify_next_seeds'
    # input) are gathered over RCCL from device buffers
    chunk = 1e-3
    u0 = np.zeros(N); u0[0] = 1.0

    def kparams(nch):
        return capi.KinParams(tspan0=0.0, tspan1=chunk * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0,
                              solve_chunks=1, ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=-1.0)

    if args.solve_chunks > 0:
        k_rank = h.rates_at(1000.0 + 10.0 * rank)
        # cold wall-clock of ONE solve_network call as an exploration level makes it (explore_network solves every
        # network exactly once, exploration/methods.jl:221): handle creation (network compilation, table upload) +
        # first solve (symbolic LU analysis, plan uploads, allocations) on a fresh handle; `gpu_wall_s` below is the
        # same solve on the warm handle
        t1 = time.perf_counter()
        hc = capi.HipNetwork.from_flat(net)
        t_create = time.perf_counter() - t1
        hc.set_rates(k_rank)
        t1 = time.perf_counter()
        _, _, rc_cold, st_cold, _ = hc.solve(kparams(args.solve_chunks), u0)
        t_first = time.perf_counter() - t1
        hc.close()
        cold = {"create_s": t_create, "first_solve_s": t_first, "cold_wall_s": t_create + t_first, "chunks": args.solve_chunks,
                "retcode": rc_cold}
        h.solve(kparams(2), u0)     # warm-up: symbolic analysis + allocations
        if dist:
            dist.barrier()
        t1 = time.perf_counter()
        ts, us, rc, st, status = h.solve(kparams(args.solve_chunks), u0)
        gpu_wall_local = time.perf_counter() - t1
        gpu_wall = max_over_ranks(gpu_wall_local, dist, coll_dev)
        t1 = time.perf_counter()
        umax_all = gather_solution_max(h, dist, dev)         # [world][N] on every rank (RCCL all-gather on device buffers)
        gather_s = time.perf_counter() - t1
        assert umax_all.shape == (world, N) and np.array_equal(umax_all[rank], us.max(axis=0))
        if rank == 0:
            out["solve_network"] = {
                "workload": f"StaticODESolve, T=1000 K (+10 K per rank), tspan (0, {chunk * args.solve_chunks:g}) s, solve_chunkstep 1e-3 "
                            f"({args.solve_chunks} chunks), abstol 1e-10, reltol 1e-8, one replica per GPU",
                "gpu_wall_s": gpu_wall, "cold_wall_s": cold["cold_wall_s"], "cold": cold,
                "gpu_s_per_chunk": gpu_wall / args.solve_chunks, "replicas": world,
                "solves_per_s": world / gpu_wall, "retcode": rc, "stats": st,
                "solution_max_allgather_s": gather_s, "solution_max_allgather": "RCCL all-gather of N doubles per rank from device buffers"
                                                                                 if world > 1 else "single rank"}
        # CPU legs: rank 0 at N = 1 only. The device is timed on the SAME chunks the CPU baseline solves.
        if rank == 0 and world == 1 and not args.no_cpu and args.cpu_solve_chunks > 0:
            from oracle import cpu_bdf
            from oracle import oracle as orc
            nck = args.cpu_solve_chunks
            h.set_rates(k_rank)
            h.solve(kparams(nck), u0)
            t1 = time.perf_counter()
            tg, ug, rcg, stg, _ = h.solve(kparams(nck), u0)
            gpu_same = time.perf_counter() - t1
            cs = cpu_bdf.CpuSolver(net)
            pars = dict(tspan=(0.0, chunk * nck), solve_chunks=True, solve_chunkstep=chunk)
            # both sides are timed on their SECOND solve: the first one pays for the symbolic analysis (ordering, fill
            # pattern, plans) on either side; its cost is reported separately (`setup_s`)
            t1 = time.perf_counter()
            cs.solve(pars, u0, k0=k_rank)
            cpu_first = time.perf_counter() - t1
            t1 = time.perf_counter()
            tc, uc, rcc, stc = cs.solve(pars, u0, k0=k_rank)
            cpu_wall = time.perf_counter() - t1
            dev_u = np.abs(ug - uc) / (1e-10 + 1e-8 * np.abs(uc))
            # all cores: one replica per core (the CPU counterpart of one replica per GPU; the LU itself is sequential,
            # as KLU's), each on its own network handle
            import threading
            cores = orc.usable_cores()
            solvers = [cpu_bdf.CpuSolver(net) for _ in range(cores)]
            ks = [orc.arrhenius(Ea, A, 1000.0 + 10.0 * i, k_max=1e12) for i in range(cores)]
            th = [threading.Thread(target=lambda i=i: solvers[i].solve(pars, u0, k0=ks[i])) for i in range(cores)]
            for x in th:          # untimed first solve of every replica (symbolic analysis), as on the device side
                x.start()
            for x in th:
                x.join()
            th = [threading.Thread(target=lambda i=i: solvers[i].solve(pars, u0, k0=ks[i])) for i in range(cores)]
            t1 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            cpu_all_wall = time.perf_counter() - t1
            out["solve_network"].update({
                "same_chunks": nck, "gpu_wall_same_chunks_s": gpu_same, "cpu_wall_same_chunks_s": cpu_wall,
                "setup_s": {"cpu_first_solve_minus_second": cpu_first - cpu_wall, "gpu_cold": cold},
                "cpu_kind": "port (oracle/cpu_bdf.cpp: the same BDF + LU cache, left-looking sparse LU with AMD ordering, partial "
                            "pivoting and KLU-style refactorisation, g++ -O3, 1 core)",
                "speedup_same_chunks_1core": cpu_wall / gpu_same,
                "cpu_all_cores": {"cores": cores, "replicas": cores, "wall_s": cpu_all_wall, "solves_per_s": cores / cpu_all_wall,
                                  "gpu_solves_per_s_same_chunks": 1.0 / gpu_same,
                                  "gpu_over_all_cores": (1.0 / gpu_same) / (cores / cpu_all_wall)},
                "max_dev_vs_cpu_in_tol_units": float(dev_u.max()),
                "rms_dev_vs_cpu_in_tol_units": float(np.sqrt((dev_u ** 2).mean(axis=1)).max()),
                "gpu_stats_same_chunks": {q: stg[q] for q in ("n_steps", "n_rejected", "n_factor", "n_linsolve", "n_newton_fail")},
                "cpu_stats": {q: stc[q] for q in ("n_steps", "n_rejected", "n_factor", "n_linsolve", "n_newton_fail", "lu_nnz",
                                                  "t_rhs", "t_jac", "t_factor", "t_solve")}})

    # ---- ensemble throughput on ONE GPU (rank 0, N = 1 only): K replicas = K host threads of one worker process, one
    # handle (own stream) per thread. One BDF trajectory occupies ~40 workgroups of 256 CUs in 5-15 us kernels; replicas
    # fill the rest. Measured limits (tools/ensemble_scaling.py): the throughput saturates at
    # ~4 replicas (the runtime's 4 hardware queues, each a serial chain of small dependent dispatches); more hardware
    # queues (GPU_MAX_HW_QUEUES = 8 ... 24) or several worker PROCESSES (2 x 4, 4 x 4 threads) were slower, not faster.
    if rank == 0 and world == 1 and args.solve_chunks > 0 and args.replicas and "solve_network" in out:
        def replicas_leg():
            nck = max(1, args.cpu_solve_chunks)
            res = {}
            for K in [int(x) for x in args.replicas.split(",")]:
                res[str(K)] = run_replica_workers(args, 1, K, nck)
            best = max(res.values(), key=lambda q: q["solves_per_s"])
            cac = out["solve_network"].get("cpu_all_cores")
            if cac:
                cac["gpu_over_all_cores_best_K"] = best["solves_per_s"] / cac["solves_per_s"]
            return dict(res, chunks=nck, note="K host threads of one worker process, one handle per thread, each solving the first "
                        f"{nck} chunks of its own replica (1000 + 10 i K); timed between a common start signal and the last worker's report")
        guarded(out["solve_network"], "concurrent_replicas", replicas_leg)

    # ---- the BASELINE configurations as SURVEY 8(d) states them (rank 0, N = 1): C3 over its whole span (0, 1) s chunkwise
    # (1 000 default chunks) AND as one integration (solve_chunks = false, methods.jl:132-183); the first 20 chunks of the C4
    # ramp (200 rate updates / restarts) against the committed tight-tolerance truth; a 5-chunk solve of the C5 network
    # (50k species). The ramp and the complete-timespan solve need dtmin below the reference's hard-coded eps(.) on this
    # synthetic CRN: kin_params.dtmin = 1e-30, what INTEGRATION.md's shim passes as HIPBDF(dtmin).
    if rank == 0 and world == 1 and args.configs and args.solve_chunks > 0:
        RAMP_DTMIN = 1e-30

        def kpc(t1, chunk, save=-1.0, chunks=1, dtmin=0.0):
            return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=chunks,
                                  ban_negatives=0, solve_chunkstep=chunk, maxiters=100000, save_interval=save, dtmin=dtmin)

        def brief(st):
            return {q: st[q] for q in ("n_steps", "n_rejected", "n_factor", "n_newton_fail", "n_restarts", "n_retries", "n_lu_reused")}

        def c3_full():
            h.rates_at(1000.0)
            res = {}
            t1 = time.perf_counter()
            tt, uu, rc1, st1, _ = h.solve(kpc(1.0, 1e-3), u0)
            res["chunkwise_1000_chunks"] = {"wall_s": time.perf_counter() - t1, "retcode": rc1, "n_saved": len(tt), "stats": brief(st1)}
            fin = uu[-1].copy()
            mass = h.solution_dot(net.mass.astype(float))
            res["chunkwise_1000_chunks"]["mass_invariant_max_rel_drift"] = float(np.max(np.abs(mass / mass[0] - 1.0)))
            t1 = time.perf_counter()
            tt, uu, rc2, st2, _ = h.solve(kpc(1.0, 1e-3, save=1e-3, chunks=0, dtmin=RAMP_DTMIN), u0)
            res["complete_timespan"] = {"wall_s": time.perf_counter() - t1, "retcode": rc2, "n_saved": len(tt), "stats": brief(st2),
                                        "dtmin": RAMP_DTMIN}
            e = np.abs(fin - uu[-1]) / (1e-10 + 1e-8 * np.abs(uu[-1]))
            res["final_states_apart_in_tolerance_units"] = {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean()))}
            # EXTENSION (kin_params.solve_chunks = 2): the same 1 000 chunks with history, order and step size carried across
            # the chunk starts (no rate update happens there) instead of the reference's re-initialisation
            fin_c = uu[-1].copy()
            t1 = time.perf_counter()
            tt, uu, rc3, st3, _ = h.solve(kpc(1.0, 1e-3, chunks=2), u0)
            ew = np.abs(uu[-1] - fin_c) / (1e-10 + 1e-8 * np.abs(fin_c))
            res["chunkwise_1000_chunks_warm_extension"] = {"wall_s": time.perf_counter() - t1, "retcode": rc3, "n_saved": len(tt), "stats": brief(st3),
                                                           "final_state_vs_complete_timespan_units": {"max": float(ew.max()), "rms": float(np.sqrt((ew ** 2).mean()))}}
            return res

        def c3_30_chunks():
            # chunkwise and as one integration against the committed tight-tolerance truth (tests/golden/truth_c3_mid.npz)
            tp = os.path.join(ROOT, "tests", "golden", "truth_c3_mid.npz")
            if not os.path.exists(tp):
                return {"error": "tests/golden/truth_c3_mid.npz is missing"}
            z = np.load(tp)
            h.rates_at(1000.0)
            res = {"truth": "tests/golden/truth_c3_mid.npz (CPU port at 1000x tighter tolerances, every 5th chunk end)", "truth_self_check": float(z["self_check"])}
            for name, pr in (("chunkwise", kpc(0.03, 1e-3)), ("complete_timespan", kpc(0.03, 1e-3, save=5e-3, chunks=0, dtmin=RAMP_DTMIN)),
                             ("chunkwise_warm_extension", kpc(0.03, 1e-3, chunks=2))):
                t1 = time.perf_counter()
                tt, uu, rcq, stq, _ = h.solve(pr, u0)
                w = time.perf_counter() - t1
                sel = [int(np.argmin(np.abs(tt - x))) for x in z["t"]]
                e = np.abs(uu[sel] - z["u"]) / (1e-10 + 1e-8 * np.abs(z["u"]))
                res[name] = {"wall_s": w, "retcode": rcq, "stats": brief(stq),
                             "vs_truth_in_tolerance_units": {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean(axis=1)).max()),
                                                             "max_per_saved_time": [round(float(x), 1) for x in e.max(axis=1)]}}
            return res

        def c4_prefix():
            tst = np.arange(201) * 1e-3
            Tst = 500.0 + 50.0 * tst
            pr = kpc(0.2, 1e-2, save=5e-3, dtmin=RAMP_DTMIN)
            h.solve(kpc(0.02, 1e-2, save=5e-3, dtmin=RAMP_DTMIN), u0, tstops=tst[:21], T_stops=Tst[:21])      # warm-up
            t1 = time.perf_counter()
            tt, uu, rc4, st4, _ = h.solve(pr, u0, tstops=tst, T_stops=Tst)
            res = {"workload": "ramp 500 -> 1200 K at 50 K/s, ts_update 1 ms, chunk 10 ms, save 5 ms: first 20 chunks (200 restarts)",
                   "wall_s": time.perf_counter() - t1, "retcode": rc4, "n_saved": len(tt), "dtmin": RAMP_DTMIN, "stats": brief(st4)}
            tp = os.path.join(ROOT, "tests", "golden", "truth_c4_long.npz")
            if os.path.exists(tp):
                z = np.load(tp)
                sel = np.searchsorted(tt, z["t"])
                e = np.abs(uu[sel] - z["u"]) / (1e-10 + 1e-8 * np.abs(z["u"]))
                res["vs_truth_in_tolerance_units"] = {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean(axis=1)).max()),
                                                      "p99.9": float(np.percentile(e, 99.9)), "truth_self_check": float(z["self_check"]),
                                                      "truth": "tests/golden/truth_c4_long.npz (CPU port at 1000x tighter tolerances)"}
            return res

        def c5_solve():
            net5, Ea5, A5 = synthetic_crn(50000, 250000)
            t1 = time.perf_counter()
            h5 = capi.HipNetwork.from_flat(net5)
            try:
                h5.set_arrhenius(Ea5, A5, k_max=1e12)
                h5.rates_at(1000.0)
                u05 = np.zeros(50000); u05[0] = 1.0
                tt, uu, rc5, st5, _ = h5.solve(kpc(5e-3, 1e-3, dtmin=RAMP_DTMIN), u05)
                cold = time.perf_counter() - t1
                t1 = time.perf_counter()
                tt, uu, rc5, st5, _ = h5.solve(kpc(5e-3, 1e-3, dtmin=RAMP_DTMIN), u05)
                warm = time.perf_counter() - t1
                m5 = h5.solution_dot(net5.mass.astype(float))
            finally:
                h5.close()
            return {"workload": "50k species / 250k reactions, static 1000 K, 5 default chunks", "cold_wall_s_incl_create_and_analysis": cold,
                    "wall_s": warm, "retcode": rc5, "dense_block": st5["lu_dense_dim"], "stats": brief(st5),
                    "mass_invariant_max_rel_drift": float(np.max(np.abs(m5 / m5[0] - 1.0)))}

        cfg = {}
        guarded(cfg, "C3_whole_span", c3_full)
        guarded(cfg, "C3_30_chunks_vs_truth", c3_30_chunks)

        def c3_100_chunks():
            # the headline solve (100 chunks) against tests/golden/truth_c3_long.npz, as the reference runs it and with warm chunk starts
            tp = os.path.join(ROOT, "tests", "golden", "truth_c3_long.npz")
            if not os.path.exists(tp):
                return {"error": "tests/golden/truth_c3_long.npz is missing"}
            z = np.load(tp)
            h.rates_at(1000.0)
            res = {"truth": "tests/golden/truth_c3_long.npz (CPU port at 100x tighter tolerances, every 10th chunk end)", "truth_self_check": float(z["self_check"])}
            for name, pr in (("chunkwise", kpc(0.1, 1e-3)), ("chunkwise_warm_extension", kpc(0.1, 1e-3, chunks=2))):
                t1 = time.perf_counter()
                tt, uu, rcq, stq, _ = h.solve(pr, u0)
                w = time.perf_counter() - t1
                sel = [int(np.argmin(np.abs(tt - x))) for x in z["t"]]
                e = np.abs(uu[sel] - z["u"]) / (1e-10 + 1e-8 * np.abs(z["u"]))
                res[name] = {"wall_s": w, "retcode": rcq, "stats": brief(stq),
                             "vs_truth_in_tolerance_units": {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean(axis=1)).max())}}
            return res
        guarded(cfg, "C3_100_chunks_vs_truth", c3_100_chunks)
        guarded(cfg, "C4_prefix", c4_prefix)
        guarded(cfg, "C5_static", c5_solve)
        out.setdefault("solve_network", {})["configs"] = cfg

    # ---- where a GPU pays: 20-chunk static solves over network sizes - GPU warm, GPU cold (handle creation + symbolic analysis
    # + solve, what an exploration level pays), the CPU port on one core; and one LAUNCH that integrates an ensemble of
    # trajectories (kin_solve_ensemble: one workgroup per member, resident on the GPU)
    if rank == 0 and world == 1 and args.crossover and args.solve_chunks > 0:
        def crossover():
            rows = []
            for n in (100, 300, 1000, 3000, 10000):
                netn, Ean, An = (net, Ea, A) if n == N else synthetic_crn(n, 5 * n)
                u0n = np.zeros(n); u0n[0] = 1.0
                kn = None
                t1 = time.perf_counter()
                hn = capi.HipNetwork.from_flat(netn)
                try:
                    hn.set_arrhenius(Ean, An, k_max=1e12)
                    kn = hn.rates_at(1000.0)
                    pr = kparams(20)
                    _, _, rcn, stn, _ = hn.solve(pr, u0n)
                    cold = time.perf_counter() - t1
                    best = 1e9
                    for _ in range(3):
                        t1 = time.perf_counter()
                        _, un, rcn, stn, _ = hn.solve(pr, u0n)
                        best = min(best, time.perf_counter() - t1)
                    row = {"species": n, "reactions": 5 * n, "gpu_warm_s": best, "gpu_cold_s": cold, "retcode": rcn, "steps": stn["n_steps"],
                           "factorisations": stn["n_factor"], "dense_block": stn["lu_dense_dim"],
                           "integrator": "resident (one workgroup owns the trajectory)" if stn["lu_slots"] <= 64 else "host-driven (one launch per stage)"}
                    if n <= 400:   # the other integrator on the same network
                        os.environ["KIN_RESIDENT"] = "0"
                        try:
                            hn.solve(pr, u0n)
                            t1 = time.perf_counter(); hn.solve(pr, u0n); row["gpu_warm_host_driven_s"] = time.perf_counter() - t1
                        finally:
                            os.environ.pop("KIN_RESIDENT")
                finally:
                    hn.close()
                prev = out.get("solve_network", {})
                if n == N and "cpu_wall_same_chunks_s" in prev:      # measured above on this very network: not repeated
                    row["cpu_port_1core_s"] = prev["cpu_wall_same_chunks_s"]
                    row["cpu_chunks"] = prev["same_chunks"]
                    row["gpu_same_chunks_s"] = prev["gpu_wall_same_chunks_s"]
                elif not args.no_cpu:
                    from oracle import cpu_bdf
                    nck = 20 if n <= 3000 else 2
                    cs = cpu_bdf.CpuSolver(netn)
                    cp = dict(tspan=(0.0, 1e-3 * nck), solve_chunks=True, solve_chunkstep=1e-3)
                    cs.solve(cp, u0n, k0=kn)
                    t1 = time.perf_counter(); cs.solve(cp, u0n, k0=kn); cw = time.perf_counter() - t1
                    row["cpu_port_1core_s"] = cw
                    row["cpu_chunks"] = nck
                    if nck == 20:
                        row["gpu_over_cpu_1core"] = cw / best
                rows.append(row)
            return {"workload": "static 1000 K, 20 default chunks of 1 ms from u0 = delta on species 0, default tolerances", "rows": rows,
                    "note": "cpu_port_1core_s at 10 000 species is for 2 chunks (cpu_chunks); the like-for-like 2-chunk pair is speedup_same_chunks_1core"}

        def ensemble_launch():
            res = {}
            for n, Ks in ((300, (16, 256, 1024)), (1000, (16, 256, 1024))):
                netn, Ean, An = synthetic_crn(n, 5 * n)
                hn = capi.HipNetwork.from_flat(netn)
                try:
                    hn.set_arrhenius(Ean, An, k_max=1e12)
                    U0 = np.zeros((max(Ks), n)); U0[:, 0] = 1.0
                    for K in Ks:
                        Tm = np.linspace(900.0, 1300.0, K)
                        hn.solve_ensemble(kparams(2), U0[:K], T=Tm)
                        t1 = time.perf_counter()
                        _, ue, nsv, rcs, sts = hn.solve_ensemble(kparams(2), U0[:K], T=Tm)
                        w = time.perf_counter() - t1
                        res[f"{n}_species_K{K}"] = {"wall_s": w, "solves_per_s": K / w, "members_ok": int((rcs == 0).sum()), "members": K,
                                                    "steps_per_member": float(np.mean([q["n_steps"] for q in sts])),
                                                    "steps_of_the_slowest_member": int(max(q["n_steps"] for q in sts)), "lu_slots_per_member": int(sts[0]["lu_slots"])}
                finally:
                    hn.close()
            # ... and the 10k-species network (beyond one compute unit's LDS): lockstep rounds of batched launches (ensemble.cpp)
            U0 = np.zeros((16, N)); U0[:, 0] = 1.0
            Tm = 1000.0 + 10.0 * np.arange(16)
            h.solve_ensemble(kparams(2), U0, T=Tm)
            t1 = time.perf_counter()
            _, ue, nsv, rcs, sts = h.solve_ensemble(kparams(2), U0, T=Tm)
            w = time.perf_counter() - t1
            res[f"{N}_species_K16_lockstep"] = {"wall_s": w, "solves_per_s": 16 / w, "members_ok": int((rcs == 0).sum()), "members": 16,
                                                 "steps_per_member": float(np.mean([q["n_steps"] for q in sts])),
                                                 "factorisations_per_member": float(np.mean([q["n_factor"] for q in sts])),
                                                 "form": "lockstep rounds of batched launches, one host thread per member's controller, dense Schur inverses of "
                                                         "members that factorise together as one batched chain (DESIGN 3.5)"}
            # ... few members of the same network: K kin_solve calls on K host threads, one solve-only copy of the handle each
            for K8 in (4, 8):
                U8 = np.zeros((K8, N)); U8[:, 0] = 1.0
                T8 = 1000.0 + 10.0 * np.arange(K8)
                h.solve_ensemble(kparams(2), U8, T=T8)
                t1 = time.perf_counter()
                _, ue, nsv, rcs, sts = h.solve_ensemble(kparams(2), U8, T=T8)
                w = time.perf_counter() - t1
                res[f"{N}_species_K{K8}_threads"] = {"wall_s": w, "solves_per_s": K8 / w, "members_ok": int((rcs == 0).sum()), "members": K8,
                                                    "steps_per_member": float(np.mean([q["n_steps"] for q in sts])),
                                                    "form": "K <= 12 members of a network beyond the resident kernel: K kin_solve calls on K host threads"}
            # ... and a mid-size network (beyond the resident kernel's LDS budget as well): 3 000 species, 64 members
            net3, Ea3, A3 = synthetic_crn(3000, 15000)
            h3 = capi.HipNetwork.from_flat(net3)
            try:
                h3.set_arrhenius(Ea3, A3, k_max=1e12)
                U3 = np.zeros((64, 3000)); U3[:, 0] = 1.0
                T3 = 1000.0 + 5.0 * np.arange(64)
                h3.solve_ensemble(kparams(2), U3, T=T3)
                t1 = time.perf_counter()
                _, ue, nsv, rcs, sts = h3.solve_ensemble(kparams(2), U3, T=T3)
                w = time.perf_counter() - t1
                h3.rates_at(1000.0)
                h3.solve(kparams(2), U3[0])
                t1 = time.perf_counter()
                h3.solve(kparams(2), U3[0])
                w1 = time.perf_counter() - t1
                res["3000_species_K64_lockstep"] = {"wall_s": w, "solves_per_s": 64 / w, "members_ok": int((rcs == 0).sum()), "members": 64,
                                                    "steps_per_member": float(np.mean([q["n_steps"] for q in sts])),
                                                    "one_member_alone_kin_solve_s": w1, "over_one_member_at_a_time": (64 / w) * w1}
            finally:
                h3.close()
            return dict(res, workload="kin_solve_ensemble: K members of one network, first 2 default chunks each, host buffers in and out; "
                                      "300 / 1000 species: ONE launch of kin::resident_bdf_kernel (one workgroup per member; the members' temperatures "
                                      "span 900-1300 K, a launch takes as long as its slowest member once K <= the 256 compute units)")
        sn = out.setdefault("solve_network", {})
        guarded(sn, "crossover", crossover)
        guarded(sn, "ensemble_one_launch", ensemble_launch)

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
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    h.close()
    if rank == 0:
        assert out["n_gpus"] == world
        # The solve half of the metric ("wall-clock per solve_network") where the driver keeps it: the CPU-vs-GPU pair of the
        # headline solve inside `cpu_baseline` (the detail stays under `solve_network`), and `cpu_baseline` as the LAST key of the line
        sn = out.get("solve_network")
        if isinstance(sn, dict) and "gpu_wall_s" in sn:
            stt = sn.get("stats", {})
            brief_sn = {"workload": f"C3 StaticODESolve 1000 K, {args.solve_chunks} chunks of 1 ms, abstol 1e-10, reltol 1e-8 (kin_solve, warm handle)",
                        "gpu_wall_s": sn["gpu_wall_s"], "gpu_cold_wall_s": sn.get("cold_wall_s"), "retcode": sn.get("retcode"),
                        "steps": stt.get("n_steps"), "factorisations": stt.get("n_factor"), "corrector_failures": stt.get("n_newton_fail"),
                        "same_chunks": sn.get("same_chunks"), "gpu_wall_same_chunks_s": sn.get("gpu_wall_same_chunks_s"),
                        "cpu_wall_same_chunks_s": sn.get("cpu_wall_same_chunks_s"), "speedup_same_chunks_1core": sn.get("speedup_same_chunks_1core"),
                        "cpu_kind": "port, 1 core (oracle/cpu_bdf.cpp)"}
            vt = sn.get("configs", {}).get("C3_100_chunks_vs_truth", {})
            if isinstance(vt, dict) and isinstance(vt.get("chunkwise"), dict):
                brief_sn["vs_truth_c3_long_units"] = vt["chunkwise"].get("vs_truth_in_tolerance_units")
            c4 = sn.get("configs", {}).get("C4_prefix", {})
            if isinstance(c4, dict) and "wall_s" in c4:
                brief_sn["C4_prefix_20_chunks_wall_s"] = c4["wall_s"]
            if "cpu_baseline" in out:
                out["cpu_baseline"]["solve_network"] = brief_sn
            else:                              # (N > 1 or --no-cpu: no CPU baseline is taken; the summary still ends the line)
                out["solve_network_summary"] = brief_sn
        for key in ("solve_network_summary", "roofline", "cpu_baseline"):
            if key in out:
                out[key] = out.pop(key)      # (dicts keep insertion order: these end the line, inside the tail the driver keeps)
        print(json.dumps(out), flush=True)      # the LAST line of rank 0's stdout (backends may print banners before it)


if __name__ == "__main__":
    main()
