"""Multi-GPU driver: one process per GPU of one node.

The reference is single-GPU (device 0 hard-coded: gpu_dive_and_solve.hpp:537,636, barebones:532); its only
inter-workgroup state is `next_subproblem`, `appx_best_bound` and a stop flag (barebones_dive_and_solve.hpp:409-453).
Across GPUs:

* the 2^d subproblems are dealt block-cyclically (`tb_eps_global_index`) -- static, balanced, no communication;
* every session owns one cell in fine-grained device memory (queue word + imported bound).  `link_group` exchanges the
  cells' IPC handles through torch.distributed (an all_gather of 64 bytes per rank: RCCL with backend "nccl", gloo in
  the CPU tests) and maps them, after which the KERNELS exchange the incumbent (one int32 atomicMin per peer) and steal
  work from each other directly over xGMI.  During the search the processes do not talk to each other at all;
* when the cells cannot be mapped (no IPC / no peer path) the processes fall back to relaying the bound through the
  host: `exchange_until_done`, an all_reduce(MIN) of one int32 per round.

torch.distributed is plumbing here: rendezvous, the handle exchange, barriers, and the end-of-search reductions.
"""
from __future__ import annotations

import time

PINF = 2**31 - 1


def agree_on_plan(session, dist=None, tensor_device="cpu") -> dict:
    """All ranks must have planned the same 2^d (and chunking): the shares only tile the index space then."""
    plan = session.plan()
    if dist is None or dist.get_world_size() == 1:
        return plan
    import torch
    mine = torch.tensor([plan["subproblems_power"], -plan["subproblems_power"], plan["eps_chunk_log2"], -plan["eps_chunk_log2"]],
                        dtype=torch.int64, device=tensor_device)
    dist.all_reduce(mine, op=dist.ReduceOp.MIN)
    lo, hi, klo, khi = int(mine[0]), -int(mine[1]), int(mine[2]), -int(mine[3])
    if lo != hi or klo != khi:
        raise RuntimeError(f"ranks planned different subproblem counts (2^{lo} .. 2^{hi}, chunk 2^{klo} .. 2^{khi}): "
                           f"pass the same subproblems_power / eps_chunk_log2 to every rank")
    return plan


def link_group(session, dist=None, tensor_device="cpu") -> bool:
    """Map every other rank's cell into this session (collective).  Returns True when all ranks are fully linked --
    the kernels then exchange bounds and work among themselves -- False when the host relay must be used."""
    if dist is None or dist.get_world_size() == 1:
        return True
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = 1
    try:
        handle = session.export_peer()
    except Exception:
        handle, ok = bytes(64), 0
    mine = torch.tensor(list(handle) + [ok], dtype=torch.uint8, device=tensor_device)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    for r, t in enumerate(everyone):
        if r == rank:
            continue
        raw = bytes(t.cpu().tolist())
        if not raw[64]:
            ok = 0
            continue
        try:
            session.import_peer(r, raw[:64])
        except Exception:
            ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=tensor_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    linked = bool(int(flag.item()))
    if not linked:
        # partly linked: every rank drops what it imported, so that no rank counts the group's node budget in a cell the others do not add to
        session.unlink_peers()
    return linked


def run_linked(session, dist=None, period_s: float = 0.0002, max_seconds: float | None = None):
    """One search of a linked group (collective): arm, synchronise, start, wait for the own kernel, finish, synchronise.
    A peer must not touch a cell that is being reset, hence the two barriers.  Returns session.finish()."""
    session.arm()
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()
    t_released = time.time()  # (one host: the ranks share this clock)
    session.start()
    t_started = time.time()
    t0 = time.perf_counter()
    while True:
        _, done = session.poll()
        if done:
            break
        if max_seconds is not None and time.perf_counter() - t0 > max_seconds:
            session.stop()
        time.sleep(period_s)
    out = session.finish()
    # when this rank's kernel was launched (host clock) and how long the launch call took after the barrier released it: the ranks of a step
    # start within the spread of these times, and a fixed node budget for the group is consumed by whoever runs
    out[2]["host_start_time"] = t_started
    out[2]["host_launch_latency_s"] = t_started - t_released
    if dist is not None and dist.get_world_size() > 1:
        dist.barrier()
    return out


_relay_group = None


def relay_group(dist):
    """A gloo group over all ranks for everything the processes exchange WHILE a search kernel runs (collective, created once).
    The persistent kernel fills every CU: a collective of the default RCCL group would queue its own kernel
    behind it and deliver the bound when the search is over.  gloo moves a few bytes over CPU tensors and touches no GPU."""
    global _relay_group
    if dist is None or dist.get_world_size() == 1:
        return None
    if _relay_group is None:
        _relay_group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _relay_group


def exchange_until_done(session, dist=None, tensor_device="cpu", period_s: float = 0.0005, max_seconds: float | None = None, target: int | None = None,
                        trace: dict | None = None):
    """Host relay (fallback when the cells are not linked): drive one started session to completion.

    `session` needs poll() -> (local_best, done), push_bound(b) and stop().  With a process group,
    every rank calls this collectively: each round all-reduces (min) the pair (best bound, done flag),
    so all ranks leave the loop in the same round, and every rank imports the global incumbent.
    The all-reduce runs on CPU tensors over the gloo side group (relay_group), never on the GPU the search kernel occupies;
    `tensor_device` is kept for callers of earlier rounds and ignored.
    `target`: stop every rank as soon as the group's incumbent is <= target (bench.py --mode solve: time to a target objective).
    `trace` (optional dict) receives host timestamps (time.perf_counter): `t_target` -- the round in which the group's incumbent was first <= target (the same
    round on every rank, so the time to target is defined for any world size), `t_stop` -- when this rank asked its kernel to stop, `t_own_done` -- when this
    rank's own kernel was first seen finished.
    Returns (global_best, rounds).
    """
    world = dist.get_world_size() if dist is not None else 1
    buf, group = None, None
    if world > 1:
        import torch
        group = relay_group(dist)
        buf = torch.empty(2, dtype=torch.int32, device="cpu")
    gbest, rounds, t0 = PINF, 0, time.perf_counter()
    stopped = False
    if trace is not None:
        trace.update({"t_target": None, "t_stop": None, "t_own_done": None})
    while True:
        best, done = session.poll()
        rounds += 1
        if trace is not None and done and trace["t_own_done"] is None:
            trace["t_own_done"] = time.perf_counter()
        if world > 1:
            buf[0] = int(best)
            buf[1] = 1 if done else 0
            dist.all_reduce(buf, op=dist.ReduceOp.MIN, group=group)
            rbest, all_done = int(buf[0].item()), int(buf[1].item())
        else:
            rbest, all_done = int(best), int(bool(done))
        if rbest < gbest:
            gbest = rbest
            session.push_bound(gbest)
        if trace is not None and target is not None and gbest <= target and trace["t_target"] is None:
            trace["t_target"] = time.perf_counter()
        if all_done:
            return gbest, rounds
        if not stopped and ((max_seconds is not None and time.perf_counter() - t0 > max_seconds) or (target is not None and gbest <= target)):
            session.stop()
            stopped = True
            if trace is not None:
                trace["t_stop"] = time.perf_counter()
        time.sleep(period_s)


def reduce_results(has_solution: bool, best_bound: int, stats: dict, dist=None, tensor_device="cpu"):
    """End of search (reduce_blocks across ranks, barebones_dive_and_solve.hpp:1033-1067): the winner has the best bound; ties go to
    the lowest SUBPROBLEM INDEX (`stats["best_subproblem"]`), then to the lowest rank -- with block-cyclic shares and work
    stealing the lowest rank does not hold the lowest indices, and the canonical pass (`use_fixed_bound`) is only
    deterministic if the lowest subproblem wins.  Counters are summed.
    Returns (winner_rank or -1, global_bound, summed_stats)."""
    keys = ["nodes", "fails", "solutions", "fixpoint_iterations", "num_deductions", "eps_solved_subproblems",
            "eps_skipped_subproblems", "num_blocks_done", "store_writes", "eps_stolen_subproblems"]
    if dist is None or dist.get_world_size() == 1:
        return (0 if has_solution else -1), (best_bound if has_solution else PINF), {k: stats.get(k, 0) for k in keys}
    import torch
    world = dist.get_world_size()
    sub = int(stats.get("best_subproblem", -1))
    mine = torch.tensor([1 if has_solution else 0, int(best_bound), sub if sub >= 0 else 2**62], dtype=torch.int64, device=tensor_device)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    rows = [tuple(int(x) for x in t.tolist()) for t in everyone]
    cands = [(b, sp, r) for r, (h, b, sp) in enumerate(rows) if h]
    winner, gbound = (-1, PINF) if not cands else (min(cands)[2], min(cands)[0])
    s = torch.tensor([int(stats.get(x, 0)) for x in keys], dtype=torch.int64, device=tensor_device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return winner, gbound, dict(zip(keys, (int(v) for v in s.tolist())))


def gather_rank_rows(row: dict, dist=None, tensor_device="cpu") -> list:
    """Per-rank balance figures (kernel time, waiting time, stolen work ...) gathered on every rank, in rank order."""
    keys = sorted(row)
    if dist is None or dist.get_world_size() == 1:
        return [dict(row, rank=0)]
    import torch
    mine = torch.tensor([float(row[k]) for k in keys], dtype=torch.float64, device=tensor_device)
    everyone = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(everyone, mine)
    return [dict(zip(keys, t.tolist()), rank=r) for r, t in enumerate(everyone)]
