"""One-process-per-GPU layer: torch.distributed over RCCL ("nccl" backend on ROCm) on device tensors, gloo in the
CPU / single-card rehearsals.

A single solve_network trajectory is a sequential chain in time (each chunk starts from the previous chunk's final
state, reference src/solving/methods.jl:819), so time chunks do not shard. What does:

  (1) the rows of the discrete rate table (calculate_discrete_rates, solve_utils.jl:91-109): ranks generate disjoint
      slices of time stops on their devices (`rate_table_sharded`); an all-gather of device buffers follows only when the
      caller wants the whole sol_k;
  (2) independent replicas / ensemble members (`solve_ensemble`): one kin_solve per rank, no data-path collective; the
      per-species maxima identify_next_seeds reads (explore_utils.jl:344-351) are all-gathered from device buffers
      (`gather_solution_max`); ensembles of ONE network shard by members (`solve_ensemble_sharded`): one
      kin_solve_ensemble call per rank, then one row per member all-gathered;
  (3) the reactions of ONE trajectory's right-hand side (`rhs_reaction_blocks`): rank g evaluates its block of reactions
      and an all-reduce of N doubles sums the partial du. Latency bound at these sizes (80 kB at 10k species) - built so
      that the cost can be measured (`time_rhs_reaction_blocks`, reported by bench.py at N > 1), not as the recommended path.

Collectives take device tensors when the backend is RCCL and stage through the host when it is gloo (gloo cannot
all-gather device tensors), so the same code runs 2 ranks on one card in the tests.
"""
from __future__ import annotations

import os
import time

import numpy as np


def rank_world(dist=None):
    if dist is not None and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n, rank, world):
    """Contiguous, balanced partition of range(n): the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _active(dist):
    return dist is not None and dist.is_initialized() and dist.get_world_size() > 1


def _on_device(dist):
    """True when collectives can take device tensors directly (RCCL)."""
    return dist.get_backend() == "nccl"


def max_over_ranks(x, dist=None, device="cpu"):
    """MAX all-reduce of a python float (the bench contract's max-over-ranks step time)."""
    if not _active(dist):
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_rows(t, dist):
    """All-gather of a [rows_r][C] tensor whose row count differs by at most one between ranks (shard_range slices):
    returns the concatenation in rank order on every rank, on the tensor's device. Rows are padded to the largest
    slice for the collective (RCCL's all-gather wants equal sizes)."""
    import torch
    if not _active(dist):
        return t
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device if _on_device(dist) else "cpu")
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    src = t if _on_device(dist) else t.cpu()
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
    pad[:t.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).to(t.device)


def rate_table_sharded(h, T_stops, dist=None, device=None, gather=True):
    """(1) Each rank generates the rate-table rows of ITS slice of time stops on its device (kin_rate_table_dev) -
    5.6 GB at C4 is 0.7 GB per rank on 8 GPUs. gather=True all-gathers the slices (device buffers over RCCL) and
    returns the full [S][R] tensor on every rank; gather=False returns (lo, hi, slice)."""
    import torch
    rank, world = rank_world(dist) if _active(dist) else (0, 1)
    T_stops = np.asarray(T_stops, dtype=np.float64)
    lo, hi = shard_range(len(T_stops), rank, world)
    mine = torch.empty((hi - lo, h.nr), dtype=torch.float64, device=device)
    if hi > lo:
        h.rate_table_dev(T_stops[lo:hi], mine.data_ptr())
    if not gather:
        return lo, hi, mine
    return all_gather_rows(mine, dist)


def gather_solution_max(h, dist=None, device=None):
    """(2) max_t u_i(t) of this rank's stored solution, reduced on the device, all-gathered over the ranks from device
    buffers: [world][N] numpy on every rank."""
    import torch
    mine = torch.empty(h.n, dtype=torch.float64, device=device)
    h.solution_max_dev(mine.data_ptr())
    if not _active(dist):
        return mine.cpu().numpy()[None, :]
    world = dist.get_world_size()
    if _on_device(dist):
        full = torch.empty((world, h.n), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(full, mine)
        return full.cpu().numpy()
    parts = [torch.empty(h.n, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, mine.cpu())
    return torch.stack(parts).numpy()


def solve_ensemble(members, solve_one, dist=None):
    """(2) Ensemble of independent solves (replicas): member i goes to rank i % world; every rank returns the list of
    results in member order. `solve_one(member)` runs one solve on this rank's GPU and returns a float64 vector of a
    fixed length (e.g. max_t u_i(t), kin_solution_max); results travel as tensors, not pickles."""
    import torch
    rank, world = rank_world(dist) if _active(dist) else (0, 1)
    mine = [(i, np.asarray(solve_one(m), dtype=np.float64)) for i, m in enumerate(members) if i % world == rank]
    if world == 1:
        return [r for _, r in mine]
    dev = torch.device("cuda", torch.cuda.current_device()) if _on_device(dist) else torch.device("cpu")
    width = torch.tensor([mine[0][1].size if mine else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    w = int(width.item())
    rows = torch.tensor(np.stack([r for _, r in mine]) if mine else np.zeros((0, w)), dtype=torch.float64, device=dev)
    allrows = all_gather_rows(rows, dist).cpu().numpy()
    # rank r holds members r, r + world, ...: undo the round-robin
    counts = [len(range(r, len(members), world)) for r in range(world)]
    offs = np.concatenate([[0], np.cumsum(counts)])
    out = [None] * len(members)
    for r in range(world):
        for j, i in enumerate(range(r, len(members), world)):
            out[i] = allrows[offs[r] + j]
    return out


def solve_ensemble_sharded(h, pars, u0s, T=None, k=None, dist=None, device=None, reduce="max", solve_fn=None):
    """(2b) An ensemble of K trajectories of ONE network over the ranks: rank r takes the contiguous block
    shard_range(K, r, world) of members and integrates it with ONE kin_solve_ensemble call on its GPU (resident kernel or
    lockstep rounds, csrc/resident.hip / ensemble.cpp) - no data-path collective. What travels afterwards is one row per
    member, all-gathered in member order: reduce="max" -> max_t u_i(t) (what identify_next_seeds reads,
    explore_utils.jl:344-351), "final" -> the last saved state. Returns (rows [K][N], retcodes [K]) on every rank.
    `solve_fn(h, pars, u0s, T, k)` replaces h.solve_ensemble in tests without a device."""
    import torch
    rank, world = rank_world(dist) if _active(dist) else (0, 1)
    u0s = np.ascontiguousarray(u0s, dtype=np.float64)
    K, N = u0s.shape
    lo, hi = shard_range(K, rank, world)
    if hi > lo:
        fn = solve_fn or (lambda h_, p_, u_, T_, k_: h_.solve_ensemble(p_, u_, T=T_, k=k_))
        t, u, ns, rcs, _ = fn(h, pars, u0s[lo:hi], None if T is None else np.asarray(T, dtype=np.float64)[lo:hi],
                              None if k is None else np.ascontiguousarray(k, dtype=np.float64)[lo:hi])
        rows = np.stack([(u[i, :max(int(ns[i]), 1)].max(axis=0) if reduce == "max" else u[i, max(int(ns[i]), 1) - 1]) for i in range(hi - lo)])
        rcs = np.asarray(rcs, dtype=np.float64)
    else:
        rows, rcs = np.zeros((0, N)), np.zeros(0)
    if not _active(dist):
        return rows, rcs.astype(np.int32)
    dev = (device or torch.device("cuda", torch.cuda.current_device())) if _on_device(dist) else torch.device("cpu")
    both = torch.tensor(np.concatenate([rows, rcs[:, None]], axis=1), dtype=torch.float64, device=dev)
    full = all_gather_rows(both, dist).cpu().numpy()          # contiguous blocks in rank order = member order
    return full[:, :N], full[:, N].astype(np.int32)


def _nonnull_stream():
    """Pointer of torch's current stream, which is made a non-null one first: kernels enqueued through the C ABI on this
    stream are ordered with torch's collectives (a null pointer would select the handle's own stream instead)."""
    import torch
    if torch.cuda.current_stream().cuda_stream == 0:
        torch.cuda.set_stream(torch.cuda.Stream())
    return torch.cuda.current_stream().cuda_stream


def rhs_reaction_blocks(h, d_u, d_du, dist=None, stream=0):
    """(3) One right-hand side of ONE trajectory with the reactions split over the ranks: this rank's block into d_du
    (a torch device tensor of N doubles), then a SUM all-reduce. Every rank ends with the full du."""
    import torch
    stream = stream or _nonnull_stream()
    rank, world = rank_world(dist) if _active(dist) else (0, 1)
    lo, hi = shard_range(h.nr, rank, world)
    h.rhs_block_dev(lo, hi, d_u.data_ptr(), d_du.data_ptr(), stream)
    if _active(dist):
        if _on_device(dist):
            dist.all_reduce(d_du, op=dist.ReduceOp.SUM)
        else:
            torch.cuda.synchronize()
            t = d_du.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            d_du.copy_(t)
    return d_du


def time_rhs_reaction_blocks(h, d_u, dist=None, reps=200):
    """Measured cost of (3): microseconds per reaction-block RHS + all-reduce, per single-rank full RHS on the same
    buffers, and per bare all-reduce of N doubles (max over ranks)."""
    import torch
    d_du = torch.empty_like(d_u)
    stream = _nonnull_stream()

    def run(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        if _active(dist):
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    dev = d_u.device if (not _active(dist) or _on_device(dist)) else "cpu"
    split = max_over_ranks(run(lambda: rhs_reaction_blocks(h, d_u, d_du, dist, stream)), dist, dev)
    full = max_over_ranks(run(lambda: h.rhs_block_dev(0, h.nr, d_u.data_ptr(), d_du.data_ptr(), stream)), dist, dev)
    bare = None
    if _active(dist) and _on_device(dist):
        bare = max_over_ranks(run(lambda: dist.all_reduce(d_du, op=dist.ReduceOp.SUM)), dist, dev)
    return {"split_rhs_plus_allreduce_us": split, "single_rank_full_rhs_us": full, "bare_allreduce_us": bare,
            "bytes_all_reduced": int(d_u.numel()) * 8, "ranks": rank_world(dist)[1] if _active(dist) else 1}
