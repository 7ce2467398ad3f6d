"""Multi-GPU driver: one process per GPU, the EPS index space sharded in contiguous slices, and the
incumbent objective bound as the only payload exchanged during the search.

The reference is single-GPU (device 0 hard-coded: gpu_dive_and_solve.hpp:537,636, barebones:532); its
only inter-workgroup state is `next_subproblem`, `appx_best_bound` and a stop flag
(barebones_dive_and_solve.hpp:409-453).  Across GPUs the work counter becomes a static slice per rank
(`tb_eps_slice`) and the bound becomes an `all_reduce(MIN)` of one int32 over RCCL (backend "nccl" on
ROCm; "gloo" in the CPU tests).  A 4-byte message is latency bound: link bandwidth is irrelevant.
"""
from __future__ import annotations

import time

PINF = 2**31 - 1


def exchange_until_done(session, dist=None, tensor_device="cpu", period_s: float = 0.0005, max_seconds: float | None = None):
    """Drive one started session to completion.

    `session` needs poll() -> (local_best, done), push_bound(b) and stop().  With a process group,
    every rank calls this collectively: each round all-reduces (min) the pair (best bound, done flag),
    so all ranks leave the loop in the same round, and every rank imports the global incumbent.
    Returns (global_best, rounds).
    """
    world = dist.get_world_size() if dist is not None else 1
    buf = None
    if world > 1:
        import torch
        buf = torch.empty(2, dtype=torch.int32, device=tensor_device)
    gbest, rounds, t0 = PINF, 0, time.perf_counter()
    while True:
        best, done = session.poll()
        rounds += 1
        if world > 1:
            buf[0] = int(best)
            buf[1] = 1 if done else 0
            dist.all_reduce(buf, op=dist.ReduceOp.MIN)
            rbest, all_done = int(buf[0].item()), int(buf[1].item())
        else:
            rbest, all_done = int(best), int(bool(done))
        if rbest < gbest:
            gbest = rbest
            session.push_bound(gbest)
        if all_done:
            return gbest, rounds
        if max_seconds is not None and time.perf_counter() - t0 > max_seconds:
            session.stop()
        time.sleep(period_s)


def reduce_results(has_solution: bool, best_bound: int, stats: dict, dist=None, tensor_device="cpu"):
    """End of search: min of bounds (ties -> lowest rank = lowest subproblem slice), sum of counters.
    Returns (winner_rank or -1, global_bound, summed_stats)."""
    keys = ["nodes", "fails", "solutions", "fixpoint_iterations", "num_deductions", "eps_solved_subproblems",
            "eps_skipped_subproblems", "num_blocks_done", "store_writes"]
    if dist is None or dist.get_world_size() == 1:
        return (0 if has_solution else -1), (best_bound if has_solution else PINF), {k: stats.get(k, 0) for k in keys}
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    # pack (bound, rank) so that MIN picks the best bound and, on ties, the lowest rank
    key = ((int(best_bound) + 2**31) * world + rank) if has_solution else (2**62)
    k = torch.tensor([key], dtype=torch.int64, device=tensor_device)
    dist.all_reduce(k, op=dist.ReduceOp.MIN)
    kmin = int(k.item())
    winner, gbound = (-1, PINF) if kmin == 2**62 else (kmin % world, kmin // world - 2**31)
    s = torch.tensor([int(stats.get(x, 0)) for x in keys], dtype=torch.int64, device=tensor_device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return winner, gbound, dict(zip(keys, (int(v) for v in s.tolist())))
