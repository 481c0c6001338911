"""One-process-per-GPU helpers (torch.distributed over RCCL on GPUs, gloo in the CPU tests).

A single solve_network trajectory is a sequential chain in time (each chunk starts from the
previous chunk's final state, reference src/solving/methods.jl:819), so time chunks do not shard
(SURVEY.md 8(e)). What does: independent replicas (ensemble members: different u0 / conditions)
and the rows of the discrete rate table (one row per time stop, solve_utils.jl:91-109). Both
partitions need no data-path collective; results are gathered once at the end.
"""
from __future__ import annotations

import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n, rank, world):
    """Contiguous, balanced partition of range(n): the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(x, dist=None, device="cpu"):
    """MAX all-reduce of a python float (the bench contract's max-over-ranks step time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rate_table_sharded(make_rows, T_stops, dist=None):
    """Each rank generates the rows of its slice of time stops (make_rows(T_slice) -> [s][R]);
    rank 0 receives the full table in stop order (only needed when the caller wants sol_k)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
    lo, hi = shard_range(len(T_stops), rank, world)
    mine = make_rows(np.asarray(T_stops)[lo:hi])
    if world == 1:
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, (lo, np.asarray(mine)))
    parts.sort(key=lambda p: p[0])
    return np.concatenate([p[1] for p in parts], axis=0)


def solve_ensemble(members, solve_one, dist=None):
    """Ensemble of independent solves (replicas): member i goes to rank i % world; every rank
    returns the full list of results in member order. `solve_one(member)` runs one solve_network
    on this rank's GPU and returns a picklable result (e.g. max_t u_i(t) for identify_next_seeds)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None and dist.is_initialized() else (0, 1)
    mine = [(i, solve_one(m)) for i, m in enumerate(members) if i % world == rank]
    if world == 1:
        return [r for _, r in mine]
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    flat = sorted((x for p in parts for x in p), key=lambda x: x[0])
    return [r for _, r in flat]
