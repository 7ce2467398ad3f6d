#!/usr/bin/env python3
"""Benchmark of the dive-and-solve hot path (BASELINE.json metric: propagations/sec + nodes/sec on
wordpress7_500.fzn at 1/2/4/8 MI355X).

A "step" is one launch of the persistent search kernel(s) over the EPS index space with a fixed node budget for the
WHOLE search (all workgroups of all GPUs together, `stop_after_n_nodes_total`), inputs already resident in HBM
(tb_session_create uploads them before the timed region).  The budget does not depend on the number of GPUs, so
`--gpus N` measures strong scaling: the same amount of search in less time.  (`--scaling weak` multiplies it by N.)

`python bench.py --gpus N` started on its own (no WORLD_SIZE / RANK in the environment) launches its N ranks itself: the parent
never touches a GPU, starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, relays
its output and exits with its code.  Started by torch.distributed.run (the driver's way) it is one of the ranks.  A rank whose
world size differs from --gpus, or that has no device of its own (without --share-device), exits non-zero: the line never
says `n_gpus` != --gpus.

N > 1: one process per GPU (torch.distributed; backend "nccl" is RCCL).  The 2^d subproblems are dealt block-cyclically,
every rank's session is linked to the others' (IPC handles exchanged with an all_gather), and during a step the KERNELS
exchange the incumbent bound and steal work from each other over xGMI; the processes only meet at the barriers around a
step.  `--exchange host` uses the fallback instead (all_reduce MIN of one int32 relayed by the hosts, static shares).

Default fixpoint is the engine's event-driven one (same tree as WAC1, fewest propagator evaluations per node); the
reference-compatible WAC1 sweep is timed beside it (`--side-steps`).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# instance, node budget of one step for the event / sweeping fixpoints.  The budget is the same at every N (strong scaling), so
# it is sized for the LARGEST run: at 8 GPUs a step of the headline workload still lasts >= 0.3 s (process-launch skew after the
# arm -> barrier -> start handshake is well under that), which makes it 3-4 s on one GPU.
WORKLOADS = {
    "wordpress7_500": ("example_wordpress7_500.fzn", 144_000_000, 6_000_000),
    "accap_a3": ("accap_a3.fzn", 24_000_000, 24_000_000),
    "trains15": ("trains15.fzn", 24_000_000, 8_000_000),
    "synthetic": ("synthetic 100k x 500k (seed 42)", 48_000, 24_000),
}
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBPS = 34500.0      # MI355X_MICROARCH.md: aggregate L2 bandwidth (8 XCDs)
LDS_BYTES_PER_CLK_CU = 256  # MI355X_MICROARCH.md: ds_read_b64 / b128
RECORD_BYTES = 16           # one bytecode per propagation ...
DOMAIN_BYTES = 24           # ... and three 8-byte domains; + 8 B per narrowed bound written
FP_CODE = {"ac1": 0, "wac1": 1, "event": 2}


def pinned_core() -> int | None:
    """Pin this process to one core for the CPU leg (1 core, pinned); returns the core or None."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        core = cores[len(cores) // 2]
        os.sched_setaffinity(0, {core})
        return core
    except Exception:
        return None


def cpu_baseline(tcn, subproblems_power: int, seconds: float, leaf_rule: str = "barebones") -> dict:
    """The oracle (CPU restatement of cpu_solving.hpp) on a bounded sample of the SAME node population the GPU step
    explores: the same 2^d EPS decomposition walked in index order (dive nodes + solve nodes of the first subproblems),
    1 core, pinned, built -O3 -march=native on this machine.  The plain DFS (the reference's own `-arch cpu` order) is a
    second row."""
    from oracle import pyoracle
    before = None
    try:
        before = os.sched_getaffinity(0)
    except Exception:
        pass
    native = pyoracle.build_native(tempfile.gettempdir())
    if native:
        pyoracle.use_library(native)
    core = pinned_core()
    try:
        # The CPU leg walks the SAME tree as the GPU rows: the leaf rule of the GPU rows (`--leaf-rule`; r05 always ran cpu_solving.hpp:36's rule here, under which an
        # all-entailed box with open variables is an inner node -- a different tree wherever such boxes occur; on a bounded sample of the headline instance none does
        # and the measured rates were identical, but "like for like" should not rest on that).  The rule is recorded in the row.
        lra = int(leaf_rule == "gpu")
        _, _, st = pyoracle.solve(tcn, subproblems_power=subproblems_power, timeout_ms=int(seconds * 1000), leaf_requires_assignment=lra)
        _, _, dfs = pyoracle.solve(tcn, subproblems_power=0, timeout_ms=int(seconds * 300), leaf_requires_assignment=lra)
    finally:
        if before is not None:
            try:
                os.sched_setaffinity(0, before)
            except Exception:
                pass
    secs, dsecs = max(st["solve_seconds"], 1e-9), max(dfs["solve_seconds"], 1e-9)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": st["num_deductions"] / secs, "unit": "propagations/s", "cores": 1, "kind": "port",
            "nodes_per_sec": st["nodes"] / secs,
            "sample": f"first {secs:.1f} s of the sequential walk over the same 2^{subproblems_power} EPS subproblems the GPU step explores "
                      f"(dive + solve nodes, AC1 Gauss-Seidel, leaf rule of -arch {leaf_rule}, the same as the GPU rows): {st['nodes']} nodes, {st['num_deductions']} propagations, "
                      f"{st['eps_solved_subproblems'] + st['eps_skipped_subproblems']} subproblems done",
            "leaf_rule": leaf_rule,
            "build": "gcc -O3 -march=native (built on this host)" if native else "gcc -O3 (prebuilt, portable)",
            "pinned_core": core, "host_cpus": os.cpu_count(), "cpu_model": model,
            "dfs_sample": {"value": dfs["num_deductions"] / dsecs, "nodes_per_sec": dfs["nodes"] / dsecs,
                           "sample": f"first {dsecs:.1f} s of the plain DFS branch-and-bound (no EPS; deep nodes only): {dfs['nodes']} nodes"}}


def reference_invocation(seconds: float) -> dict:
    """The reference's own example command (README.md:25: `turbo -s -v -i -t 20000 benchmarks/example_wordpress7_500.fzn`) through
    this repo's `turbo` executable, with a shorter -t: what a user of the reference sees for the headline instance."""
    import subprocess
    exe = os.path.join(ROOT, "turbo_amd", "bin", "turbo")
    cmd = [exe, "-s", "-v", "-i", "-t", str(int(seconds * 1000)), os.path.join("benchmarks", "example_wordpress7_500.fzn")]
    rec = {"command": "turbo_amd/bin/turbo " + " ".join(cmd[1:]), "reference_command": "turbo -s -v -i -t 20000 benchmarks/example_wordpress7_500.fzn (README.md:25)"}
    if not os.path.exists(exe):
        return dict(rec, error="turbo_amd/bin/turbo is not built (make cli)")
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=seconds + 120)
    except Exception as e:  # noqa: BLE001
        return dict(rec, error=repr(e))
    stats = {}
    for line in p.stdout.splitlines():
        if line.startswith("%%%mzn-stat: ") and "=" in line:
            k, v = line[len("%%%mzn-stat: "):].split("=", 1)
            stats[k] = v.strip().strip('"')
    num = lambda k: (float(stats[k]) if k in stats else None)  # noqa: E731
    rec.update({"returncode": p.returncode, "wall_s": time.perf_counter() - t0,
                "best_objective": int(stats["objective"]) if "objective" in stats else None,
                "solutions_printed": sum(1 for l in p.stdout.splitlines() if l == "----------"),
                "exhaustive": any(l == "==========" for l in p.stdout.splitlines()),
                "nodes": int(stats["nodes"]) if "nodes" in stats else None,
                "failures": int(stats["failures"]) if "failures" in stats else None,
                "peak_depth": int(stats["peakDepth"]) if "peakDepth" in stats else None,
                "num_deductions": int(stats["num_deductions"]) if "num_deductions" in stats else None,
                "solve_time_s": num("solveTime"), "init_time_s": num("initTime"), "kernel_time_s": num("kernel_time"),
                "nodes_per_second": num("nodes_per_second"), "propagations_per_second": num("propagations_per_second"),
                "fixpoint": stats.get("fixpoint"), "num_blocks": stats.get("num_blocks"), "memory_configuration": stats.get("memory_configuration"),
                "eps_num_subproblems": stats.get("eps_num_subproblems"), "eps_solved_subproblems": stats.get("eps_solved_subproblems")})
    if p.returncode != 0:
        rec["stderr_tail"] = p.stderr[-500:]
    return rec


def profile_figures(workload: str, fixpoint: str) -> dict | None:
    """Counter-derived figures of the same command, collected by scripts/profile_round.sh in separate rocprofv3 --pmc
    passes and committed under profiles/ (they are NOT measured in this run: the source file is named)."""
    for name in ("r06_counters.json", "r05_counters.json", "r04_counters.json", "r03_counters.json", "r02_counters.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            rec = json.load(open(path)).get(f"{workload}/{fixpoint}")
        except Exception:
            rec = None
        if rec:
            return dict(rec, source=f"profiles/{name}")
    return None


def roofline_record(info: dict, mem_kind: int, props: float, writes: float, kernel_s: float) -> dict:
    """algorithmic bytes per propagation = 16 B record + 3 x 8 B domains (+ 8 B per narrowed bound).  `levels` prices the algorithmic bytes each level of
    the memory system would have to serve against THAT level's peak (so no fraction can exceed 1); `bound` names the level that actually serves the domains:
    "lds" when the store is LDS resident (records then come from the L2), and for a store in global memory "hbm" until attach_counters() has looked at the counters of
    the same command -- an L2 hit rate above one half and fabric reads well under the 40 algorithmic bytes per propagation mean the L2 serves the kernel, and `bound` /
    `frac` / `peak` are switched to that level (the team rows read "0.68 of HBM" while HBM carried 0.11 of its peak).
    `props`, `writes`: per launch; `kernel_s`: average launch duration."""
    lds_peak = info["compute_units"] * LDS_BYTES_PER_CLK_CU * info["clock_khz"] * 1e3 / 1e9
    dt = max(kernel_s, 1e-12)
    dom_gbps = (props * DOMAIN_BYTES + writes * 8) / dt / 1e9
    rec_gbps = props * RECORD_BYTES / dt / 1e9
    alg = dom_gbps + rec_gbps
    if mem_kind != 0:
        levels = {"lds": {"what": "3 x 8 B domains per propagation + 8 B per narrowed bound", "achieved": dom_gbps, "peak": lds_peak, "frac": dom_gbps / lds_peak},
                  "l2": {"what": "16 B record per propagation", "achieved": rec_gbps, "peak": L2_PEAK_GBPS, "frac": rec_gbps / L2_PEAK_GBPS},
                  "hbm": {"what": "nothing in steady state (records and snapshot slabs are L2 / Infinity Cache resident); measured bytes: hbm_counters", "achieved": None, "peak": HBM_PEAK_GBPS, "frac": None}}
        roof = {"bound": "lds", "achieved": dom_gbps, "peak": lds_peak, "unit": "GB/s", "frac": dom_gbps / lds_peak, "levels": levels,
                "records_from_l2": {"achieved": rec_gbps, "peak": L2_PEAK_GBPS, "unit": "GB/s", "frac": rec_gbps / L2_PEAK_GBPS}}
    else:
        levels = {"lds": {"what": "the hot tier's share of the gathers (event kernel of a store in global memory), not counted separately", "achieved": None, "peak": lds_peak, "frac": None},
                  "l2": {"what": "all 40 B per propagation (+ writes) if the L2 serves them", "achieved": alg, "peak": L2_PEAK_GBPS, "frac": alg / L2_PEAK_GBPS},
                  "hbm": {"what": "all 40 B per propagation (+ writes) if every access went to memory", "achieved": alg, "peak": HBM_PEAK_GBPS, "frac": alg / HBM_PEAK_GBPS}}
        roof = {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBPS, "levels": levels,
                "bound_chosen_by": "residence (no counters of this command found under profiles/)"}
    roof.update({"traffic": None, "kernel": "tb::solve_kernel", "avg_launch_ms": kernel_s * 1000.0,
                 "algorithmic_bytes_per_launch": props * (RECORD_BYTES + DOMAIN_BYTES) + writes * 8})
    return roof


def attach_counters(roof: dict, workload: str, fixpoint: str, launch_s: float) -> None:
    """`traffic` = memory-side bytes per launch from the committed rocprofv3 --pmc passes of the same command (profiles/, NOT measured in this run:
    the source file is named), scaled to this run's launch duration when the profiled launch was sized differently."""
    prof = profile_figures(workload, fixpoint)
    if not prof:
        return
    roof["traffic_source"] = prof.get("source")
    if prof.get("hbm_bytes_per_launch") and prof.get("launch_ms"):
        roof["traffic"] = prof["hbm_bytes_per_launch"] * (launch_s * 1000.0 / prof["launch_ms"])
        roof["traffic_note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B of the counter passes named in traffic_source, per launch, scaled by this run's launch duration over "
                                "the profiled one; MALL hits included (the guide's correction for 16-byte-per-lane streams)")
    roof["hbm_counters"] = {k: prof[k] for k in ("hbm_bytes_per_launch", "hbm_gbps", "hbm_frac_of_peak", "launch_ms", "tcc_hit_rate", "fabric_read_bytes_per_propagation") if k in prof}
    if "hbm" in roof.get("levels", {}) and prof.get("hbm_gbps") is not None:
        roof["levels"]["hbm"]["measured_gbps"] = prof["hbm_gbps"]
        roof["levels"]["hbm"]["measured_frac"] = prof["hbm_gbps"] / HBM_PEAK_GBPS
    if roof["bound"] == "hbm":
        # a store in global memory: which level serves the gathers is a measured fact, not a property of the plan
        hit, fab = prof.get("tcc_hit_rate"), prof.get("fabric_read_bytes_per_propagation")
        if hit is not None:
            served_by_l2 = hit >= 0.5 and (fab is None or fab < 0.5 * (RECORD_BYTES + DOMAIN_BYTES))
            lvl = "l2" if served_by_l2 else "hbm"
            roof.update({"bound": lvl, "peak": roof["levels"][lvl]["peak"], "frac": roof["levels"][lvl]["frac"],
                         "bound_chosen_by": f"counters of the same command ({prof.get('source')}): L2 hit rate {hit:.2f}" + (f", {fab:.1f} B of fabric reads per propagation" if fab is not None else "")
                                            + (" -- the L2 serves the 40 algorithmic bytes" if served_by_l2 else " -- most gathers go to memory")})
    roof["issue"] = {k: prof[k] for k in ("valu_busy", "salu_busy", "lds_busy", "wait_any_share", "wait_inst_any_share",
                                          "valu_per_64_propagations", "salu_per_64_propagations", "valu_per_node", "salu_per_node", "icache_hit_rate") if k in prof}
    roof["issue"]["note"] = "binding resource; rocprofv3 --pmc SQ_* passes of this command, not measured in this run"


# --mode solve: (fixed bound of the proof run, target of the time-to-target run, 2^d) per workload, calibrated on one MI355X so that a run lasts seconds
# (scripts/r05_solve_probe.py).  The proof run is a satisfaction search under the CONSTANT constraint objective <= B (tb_config.use_fixed_bound:
# no incumbent is exchanged, the tree does not depend on timing): with B below the optimum every subproblem is refuted -- a fixed amount of work whatever the number of
# GPUs, which is what a strong-scaling curve needs.  -1: the reference's 2^d rule.
# Calibration (gpurun_out/r05_solve_ladder.log, one MI355X): wordpress7_500 -- `objective <= 500` is refuted in 0.86 s (4.7e7 nodes, 2^21 subproblems: 2 060 486 solved +
# 36 666 skipped), <= 1000 is not within 17 s; branch and bound reaches 14 000 after 9 s (4.9e8 nodes).  trains15 -- <= 0 is refuted in 0.54 s (every one of the 2^20 subproblems dies
# in its dive: queue and dive throughput), <= 10 leaves a dozen hard subproblems after 17 s; 75 is reached after 0.9 s.  accap_a3 at 2^16 -- <= 55 is refuted in 1.2 s, 135 reached in 0.06 s.
# 2^d is FIXED per workload (what the reference's rule gives one MI355X): the rule 2^d >= 300 x workgroups x GPUs would double the subproblems -- and the dive nodes of a proof --
# with every doubling of the GPUs, and a strong-scaling curve needs the same work at every N.
SOLVE_DEFAULTS = {
    "wordpress7_500": (500, 14000, 21),
    "accap_a3": (55, 135, 16),
    "trains15": (0, 75, 20),
    "synthetic": (None, None, 17),
}


def share_of_device(args, tcn, base: dict, world: int, capi) -> int:
    """--share-device (the one-GPU stand-in for an N-GPU node): N full-grid persistent kernels of N processes are NOT co-resident on one GPU -- the second one's
    workgroups only start when the first one's leave, and `tb_session_start` of the second rank (a clock kernel on its stream) waits behind the first rank's search --
    so every rank takes 1/N of the workgroups the engine would launch: the ranks then run side by side like N smaller GPUs.  Returns the or_nodes to use (0: as planned)."""
    if not args.share_device or world <= 1 or base.get("or_nodes", 0) != 0:
        return base.get("or_nodes", 0)
    probe = capi.Session(tcn, capi.make_config(**base))
    blocks = probe.plan()["num_blocks"]
    probe.close()
    return max(1, blocks // world)


def make_search_runner(args, tcn, rank, world, local_rank, dist, tdev, capi, tdist, torch, power: int):
    """Whole searches instead of node budgets (`--mode solve`, and the `sharded_search` record of the default line): returns run(cfg_extra, stop_at) -> record.
    Every rank calls run() collectively."""
    base = dict(fixpoint=FP_CODE[args.fixpoint], or_nodes=args.or_nodes, threads_per_block=args.threads, timeout_ms=int(args.solve_timeout * 1000), device=local_rank,
                rank=rank, world_size=world, debug=args.debug_bits, subproblems_power=power, leaf_requires_assignment=int(args.leaf_rule == "gpu"))
    base["or_nodes"] = share_of_device(args, tcn, base, world, capi)

    def run(cfg_extra: dict, stop_at: int | None):
        sess = capi.Session(tcn, capi.make_config(**base, **cfg_extra))
        tdist.agree_on_plan(sess, dist, tdev)
        linked = world > 1 and args.exchange == "peer" and tdist.link_group(sess, dist, tdev)
        plan = sess.plan()
        sess.arm()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.start()
        t_started = time.perf_counter() - t0
        first_hit, t_stop_seen = None, None
        guard = None
        if world > 1 and args.check_relay:
            # (test aid) between start and finish nothing the ranks exchange may touch the GPU: every collective must run on CPU tensors over the gloo side group --
            # a collective of the RCCL group would queue a kernel behind the persistent search kernel and deliver the bound when the search is over
            guard = {"calls": 0, "orig": dist.all_reduce}
            relay = tdist.relay_group(dist)

            def checked_all_reduce(tensor, *a, **kw):
                assert tensor.device.type == "cpu", f"a {tensor.device} tensor was all-reduced while the search kernel runs"
                assert kw.get("group") is relay, "a collective outside the gloo side group while the search kernel runs"
                guard["calls"] += 1
                return guard["orig"](tensor, *a, **kw)
            dist.all_reduce = checked_all_reduce
        if world > 1:
            # collective loop over the gloo side group: agrees on the group's incumbent, relays it when the cells are not linked, ends every rank together
            trace = {}
            gbest, _ = tdist.exchange_until_done(sess, dist, period_s=0.0005, max_seconds=args.solve_timeout, target=stop_at, trace=trace)
            if trace.get("t_target") is not None:
                first_hit = trace["t_target"] - t0   # the round in which the GROUP's incumbent was first <= target (the same round on every rank)
            t_own_done = (trace["t_own_done"] - t0) if trace.get("t_own_done") is not None else None
        else:
            gbest = 2**31 - 1
            t_own_done = None
            while True:
                best, done = sess.poll()
                gbest = min(gbest, best)
                if stop_at is not None and gbest <= stop_at and first_hit is None:
                    first_hit = time.perf_counter() - t0
                    sess.stop()
                if done:
                    t_own_done = time.perf_counter() - t0
                    break
                time.sleep(0.0005)
        t_loop = time.perf_counter() - t0
        if guard is not None:
            dist.all_reduce = guard["orig"]
        has, best, st = sess.finish()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
            t = torch.tensor([wall], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        row = {"kernel_ms": st["kernel_ns"] * 1e-6, "nodes": st["nodes"], "eps_solved": st["eps_solved_subproblems"], "eps_skipped": st["eps_skipped_subproblems"],
               "stolen_subproblems": st["eps_stolen_subproblems"], "wait_share": st["wait_time_ns"] / max(1, st["cumulative_time_block_ns"]),
               "first_block_idle_ms": st["min_block_ns"] * 1e-6, "last_block_ms": st["max_block_ns"] * 1e-6, "blocks_done": st["num_blocks_done"],
               "exhaustive": st["exhaustive"], "has_solution": int(has), "best_bound": st["best_bound"] if has else 2**31 - 1, "propagations": st["num_deductions"],
               # per-rank timeline (host clock, seconds after the group's barrier): launch call returned / this rank's kernel seen finished / the agreement loop left
               "t_start_returned_s": t_started, "t_own_kernel_done_s": t_own_done if t_own_done is not None else -1.0, "t_loop_left_s": t_loop}
        per_rank = tdist.gather_rank_rows(row, dist if world > 1 else None, tdev)
        sess.close()
        tot = {k: sum(r[k] for r in per_rank) for k in ("nodes", "eps_solved", "eps_skipped", "stolen_subproblems", "propagations")}
        rec = {"seconds": wall, "time_to_target_s": first_hit, "linked": bool(linked) if world > 1 else None, "subproblems_power": plan["subproblems_power"],
               "workgroups_per_gpu": plan["num_blocks"], "threads": plan["threads_per_block"],
               "exhaustive": int(all(r["exhaustive"] for r in per_rank)), "has_solution": int(any(r["has_solution"] for r in per_rank)),
               "best_objective_bound": int(min(r["best_bound"] for r in per_rank)), **tot,
               "every_subproblem_accounted_once": int(tot["eps_solved"] + tot["eps_skipped"]) == (1 << plan["subproblems_power"]),
               "nodes_per_sec": tot["nodes"] / max(wall, 1e-9), "per_rank": per_rank}
        if guard is not None:
            rec["relay_rounds_checked_cpu_only"] = guard["calls"]
        return rec

    return run


PROOF_WHAT = ("canonical pass under the constant constraint objective <= B: no incumbent is exchanged, the tree is the same at every N; "
              "exhaustive and no solution = B is proved infeasible (every subproblem solved or skipped exactly once)")


def solve_mode(args, tcn, fzn, rank, world, local_rank, dist, tdev, capi, tdist, torch) -> int:
    """Whole searches instead of node budgets: what `reduce_blocks` (barebones:1033-1067) and `first_block_idle_time` (barebones:887-894) are about.
    Every rank runs this collectively; rank 0 prints one JSON line."""
    d_bound, d_target, d_power = SOLVE_DEFAULTS[args.workload]
    bound = args.fixed_bound if args.fixed_bound is not None else d_bound
    target = args.target if args.target is not None else d_target
    power = args.subproblems_power if args.subproblems_power >= 0 else d_power
    run = make_search_runner(args, tcn, rank, world, local_rank, dist, tdev, capi, tdist, torch, power)

    out = {"mode": "solve", "n_gpus": world, "workload": fzn, "fixpoint": args.fixpoint, "leaf_rule": args.leaf_rule, "scaling": "strong", "data": "reference instance file (no randomness)"}
    if bound is not None:
        rec = run(dict(use_fixed_bound=1, fixed_bound=int(bound)), None)
        rec["fixed_bound"] = int(bound)
        rec["what"] = PROOF_WHAT
        out["proof"] = rec
    if target is not None or bound is None:
        rec = run(dict(), target)
        rec["target"] = target
        rec["what"] = ("branch and bound until the group's incumbent is <= target (None: until optimality is proved); the incumbent travels between the GPUs "
                       "(peer cells over xGMI when linked, else the gloo relay), so the tree depends on when it arrives")
        out["to_target"] = rec
    head = out.get("proof") or out["to_target"]
    out.update({"metric": "seconds of a whole sharded search (proof under a fixed bound; time to a target objective) on " + fzn, "value": head["seconds"], "unit": "s",
                "higher_is_better": False, "vs_baseline": None, "dtype": "int32"})
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` on its own: run the N ranks as FRESH child processes (one per GPU, torch.distributed.run on
    127.0.0.1) and relay what they print.  This process never imports torch and never touches a GPU; it exits with the
    launcher's code, so a rank that fails (e.g. no device for it) makes the whole run fail."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    assert child.stdout is not None
    for line in child.stdout:  # rank 0's JSON line (and nothing else) comes through stdout; stderr is inherited
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wordpress7_500", choices=sorted(WORKLOADS))
    ap.add_argument("--fixpoint", default="event", choices=["ac1", "wac1", "event"])
    ap.add_argument("--nodes-total", type=int, default=0, help="node budget of one step, all GPUs together (0 = workload default)")
    ap.add_argument("--cutnodes", type=int, default=0, help="additionally cap every workgroup at this many nodes (the reference's -cutnodes)")
    ap.add_argument("--or-nodes", type=int, default=0, help="workgroups per GPU (0 = fill the GPU)")
    ap.add_argument("--threads", type=int, default=0, help="threads per workgroup (0 = the engine's choice)")
    ap.add_argument("--subproblems-power", type=int, default=-1, help="2^d EPS subproblems (-1 = the reference's rule: 2^d >= 300 x workgroups x GPUs)")
    ap.add_argument("--side-steps", type=int, default=2, help="steps of each of the other fixpoints (wac1, ac1, event) timed beside the headline (0 = skip)")
    ap.add_argument("--other-steps", type=int, default=2,
                    help="steps of each of the OTHER BASELINE configurations (accap_a3, trains15, synthetic 100k x 500k) timed after the headline, N = 1 only (0 = skip)")
    ap.add_argument("--reference-seconds", type=float, default=8.0,
                    help="-t of the reference's README invocation run through the turbo CLI for the `reference_invocation` record (0 = skip)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the node budget of a step does not depend on the number of GPUs; weak: it is multiplied by it")
    ap.add_argument("--exchange", default="peer", choices=["peer", "host"],
                    help="peer: the kernels exchange bound and work through their cells over xGMI; host: all_reduce(MIN) relay, static shares")
    ap.add_argument("--no-simplify", action="store_true", help="skip the network simplifier (the reference's -disable_simplify)")
    ap.add_argument("--leaf-rule", default="barebones", choices=["barebones", "gpu"],
                    help="barebones (the reference's default -arch of a GPU build): all propagators entailed = solution (barebones:988-993); gpu: ... and every variable "
                         "assigned (gpu_dive_and_solve.hpp:333-338, cpu_solving.hpp:36)")
    ap.add_argument("--mode", default="throughput", choices=["throughput", "solve"],
                    help="throughput (default, the contract of this file): steps with a fixed node budget.  solve: wall time of whole searches that exercise the SHARDED "
                         "search -- queue, subtree skips, work stealing, incumbent exchange: (i) proof of `objective <= --fixed-bound` (canonical pass: fixed total work, "
                         "strong scaling) and (ii) branch and bound until the group's incumbent reaches --target")
    ap.add_argument("--fixed-bound", type=int, default=None, help="--mode solve: the bound B of the proof run (default: per workload, SOLVE_DEFAULTS)")
    ap.add_argument("--target", type=int, default=None, help="--mode solve: the target objective of the time-to-target run (default: per workload)")
    ap.add_argument("--solve-timeout", type=float, default=120.0, help="--mode solve: give up after this many seconds per run")
    ap.add_argument("--fail-import-rank", type=int, default=-1, help="test aid: this rank refuses to map its peers' cells (TB_FAIL_IMPORT): the group must fall back to the host relay as a whole")
    ap.add_argument("--check-relay", action="store_true", help="test aid (--mode solve, N > 1): assert that every collective between start and finish runs on CPU tensors over the gloo side group")
    ap.add_argument("--sharded-search", type=int, default=1, help="append the `sharded_search` record (proof under a fixed bound: fixed total work) to the default line (0 = skip)")
    ap.add_argument("--sharded-reps", type=int, default=2, help="runs of the sharded search (the fastest is reported, all are listed)")
    ap.add_argument("--debug-bits", type=lambda v: int(v, 0), default=0, help="tuning knobs of tb_config.reserved[0] (experiments only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"], help="default nccl (= RCCL); gloo with --share-device (RCCL refuses two ranks on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="testing aid: every rank uses cuda:0")
    args = ap.parse_args()
    if args.dist_backend is None:
        args.dist_backend = "gloo" if args.share_device else "nccl"

    # The cells the GPUs exchange through travel between processes as dmabuf IPC handles: the legacy IPC mode of the HSA runtime
    # is not supported by the host driver (hipIpcGetMemHandle fails with "invalid argument").  Set before HIP initialises, in
    # the launcher and in every rank.
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return launch_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.fail_import_rank == rank:
        os.environ["TB_FAIL_IMPORT"] = "1"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` or under "
              f"torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        return 2
    import torch
    from turbo_amd import capi, frontend
    from turbo_amd import distributed as tdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and the engine has no CPU fallback")
    if args.share_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} needs device {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible "
              f"(--share-device puts every rank on cuda:0, a testing aid)", file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    dist = None
    tdev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)  # "nccl" is RCCL over xGMI
        dist.barrier()  # communicator set-up happens here, outside every timed region
        tdist.relay_group(dist)  # gloo side group (collective): what the ranks exchange while a kernel runs never queues an RCCL kernel behind it

    def load_workload(name: str):
        file_name = WORKLOADS[name][0]
        if name == "synthetic":
            from turbo_amd.synth import make_synthetic
            return make_synthetic(100_000, 500_000, seed=42)
        if args.no_simplify:
            return frontend.load_fzn(os.path.join(ROOT, "benchmarks", file_name))
        # the reference's default pipeline: root fixpoint (tb_propagate on this GPU) + network simplifier, outside the timed region
        from turbo_amd import preprocess
        return preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", file_name), device=local_rank)[1]

    fzn, budget_event, budget_sweep = WORKLOADS[args.workload]
    tcn = main_tcn = load_workload(args.workload)

    def make_session(fixpoint: str, budget: int, tcn=None):
        """Session + whether its cell is linked to every other rank's."""
        tcn = tcn if tcn is not None else main_tcn
        linked = world > 1 and args.exchange == "peer"
        per_rank_budget = budget if (world == 1 or linked) else max(1, budget // world)  # unlinked ranks count on their own
        kw = dict(fixpoint=FP_CODE[fixpoint], stop_after_n_nodes_total=per_rank_budget, stop_after_n_nodes=args.cutnodes,
                  or_nodes=args.or_nodes, threads_per_block=args.threads, timeout_ms=600000, device=local_rank, rank=rank, world_size=world, debug=args.debug_bits,
                  subproblems_power=args.subproblems_power, leaf_requires_assignment=int(args.leaf_rule == "gpu"))
        kw["or_nodes"] = share_of_device(args, tcn, kw, world, capi)  # (--share-device: 1/N of the workgroups each, so that the N kernels are co-resident)
        cfg = capi.make_config(**kw)
        sess = capi.Session(tcn, cfg)  # inputs uploaded to HBM here, outside the timed region
        tdist.agree_on_plan(sess, dist, tdev)
        if linked:
            linked = tdist.link_group(sess, dist, tdev)
            if not linked and rank == 0:
                print("bench.py: the sessions' cells could not be mapped across processes; using the host relay", file=sys.stderr)
        if world > 1 and not linked and per_rank_budget == budget:  # fell back after planning for a shared counter
            sess.close()
            cfg.stop_after_n_nodes_total = max(1, budget // world)
            sess = capi.Session(tcn, cfg)
        return sess, linked

    def one_step(sess, linked: bool) -> dict:
        if world == 1 or linked:
            _, _, st = tdist.run_linked(sess, dist if world > 1 else None)
            return st
        sess.arm()
        dist.barrier()
        sess.start()
        tdist.exchange_until_done(sess, dist, tensor_device=tdev, period_s=0.0002)
        _, _, st = sess.finish()
        dist.barrier()
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(sess, linked, steps, warmup):
        for _ in range(warmup):
            one_step(sess, linked)
        sync()
        t0 = time.perf_counter()
        keys = ("nodes", "num_deductions", "store_writes", "kernel_ns", "fixpoint_iterations", "wait_time_ns", "eps_stolen_subproblems",
                "num_blocks_done", "cumulative_time_block_ns", "eps_solved_subproblems", "eps_skipped_subproblems", "active_lane_evaluations")
        tot = {k: 0 for k in keys}
        tot["start_times"] = []
        last = None
        for _ in range(steps):
            last = one_step(sess, linked)
            for k in keys:
                tot[k] += last[k]
            tot["start_times"].append(last.get("host_start_time", 0.0))
        sync()
        elapsed = time.perf_counter() - t0
        g = dict(tot)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            agg = torch.tensor([tot["nodes"], tot["num_deductions"], tot["store_writes"], tot["active_lane_evaluations"], tot["eps_solved_subproblems"],
                                tot["eps_skipped_subproblems"], tot["eps_stolen_subproblems"], tot["kernel_ns"]], dtype=torch.int64, device=tdev)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            (g["nodes"], g["num_deductions"], g["store_writes"], g["active_lane_evaluations"], g["eps_solved_subproblems"], g["eps_skipped_subproblems"],
             g["eps_stolen_subproblems"], g["kernel_ns"]) = (int(x) for x in agg.tolist())
            # start skew of a step: spread of the ranks' kernel-launch times (host clock), worst step
            st = torch.tensor(tot["start_times"] or [0.0], dtype=torch.float64, device=tdev)
            hi, lo = st.clone(), st.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            g["start_skew_ms_max"] = float((hi - lo).max().item()) * 1000.0
        return elapsed, tot, g, last

    if args.mode == "solve":
        rc = solve_mode(args, tcn, fzn, rank, world, local_rank, dist, tdev, capi, tdist, torch)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return rc

    scale = world if args.scaling == "weak" else 1
    budget = (args.nodes_total or (budget_event if args.fixpoint == "event" else budget_sweep)) * scale
    session, linked = make_session(args.fixpoint, budget)
    plan = session.plan()
    elapsed, tot, glob, last = timed(session, linked, args.steps, args.warmup)
    steps = max(args.steps, 1)

    # per-rank balance: kernel time, share of workgroup-time spent without a subproblem, stolen work
    blocks = max(1, last["num_blocks"])
    row = {"kernel_ms": tot["kernel_ns"] * 1e-6 / steps, "nodes": tot["nodes"] / steps, "propagations": tot["num_deductions"] / steps,
           "wait_share": tot["wait_time_ns"] / max(1, tot["cumulative_time_block_ns"]),
           "stolen_subproblems": tot["eps_stolen_subproblems"] / steps, "blocks_done": tot["num_blocks_done"] / steps,
           "eps_solved": tot["eps_solved_subproblems"] / steps, "eps_skipped": tot["eps_skipped_subproblems"] / steps,
           "launch_latency_ms": last.get("host_launch_latency_s", 0.0) * 1000.0,
           "first_block_idle_ms": last["min_block_ns"] * 1e-6, "last_block_ms": last["max_block_ns"] * 1e-6, "workgroups": blocks}
    per_rank = tdist.gather_rank_rows(row, dist if world > 1 else None, tdev)

    # the reference's own fixpoints on the same workload, beside the headline: `-fp wac1` (its GPU default) and `-fp ac1` (what its
    # CPU path runs: the like-for-like row for the AC1 CPU baseline); `event` when the headline is one of those
    sides = []
    if args.side_steps > 0:
        notes = {"wac1": "the reference's own GPU default (-fp wac1): every propagator is evaluated in every sweep, a wave iterates its 64 to a local fixpoint",
                 "ac1": "the reference's -fp ac1 (block-synchronous sweeps; the fixpoint loop of its CPU path): like-for-like with the AC1 CPU baseline",
                 "event": "the engine's event-driven fixpoint: propagators re-evaluated only when one of their variables was narrowed"}
        for other in ("wac1", "ac1", "event"):
            if other == args.fixpoint:
                continue
            b2 = args.nodes_total or (budget_event if other == "event" else (budget_sweep if other == "wac1" else max(1, budget_sweep // 6)))
            b2 *= scale
            sess2, linked2 = make_session(other, b2)
            e2, _, g2, l2 = timed(sess2, linked2, args.side_steps, 1)
            sides.append({"fixpoint": other, "steps": args.side_steps, "nodes_total": b2, "nodes_per_sec": g2["nodes"] / e2,
                          "propagations_per_sec": g2["num_deductions"] / e2, "ms_per_step": e2 * 1000.0 / args.side_steps,
                          "workgroups_per_gpu": l2["num_blocks"], "threads": l2["threads_per_block"],
                          "memory": capi.MEM_KINDS[l2["mem_kind"]], "note": notes[other] + "; same search tree"})
            sess2.close()

    # the other BASELINE.json configurations (N = 1): accap_a3 (configs[2]), trains15 (configs[3]'s instance) and the synthetic 100k x 500k network
    # (configs[4], the store in global memory: the HBM-bound roofline point), a few short steps each with its own roofline record
    others = []
    if world == 1 and args.other_steps > 0:
        session.close()  # one persistent search at a time
        info0 = capi.device_info(local_rank)
        for name in ("accap_a3", "trains15", "synthetic"):
            if name == args.workload:
                continue
            t_load = time.perf_counter()
            tcn2 = load_workload(name)
            t_load = time.perf_counter() - t_load
            for fp in (("event", "wac1", "ac1") if name == "synthetic" else ("event",)):
                b2 = WORKLOADS[name][1] if fp == "event" else WORKLOADS[name][2]
                sess2, _ = make_session(fp, b2, tcn2)
                pl2 = sess2.plan()
                e2, t2, g2, l2 = timed(sess2, False, args.other_steps, 1)
                ks2 = t2["kernel_ns"] * 1e-9 / args.other_steps
                roof2 = roofline_record(info0, l2["mem_kind"], t2["num_deductions"] / args.other_steps, t2["store_writes"] / args.other_steps, ks2)
                attach_counters(roof2, name, fp, ks2)
                others.append({"workload": f"{WORKLOADS[name][0]}{'' if args.no_simplify or name == 'synthetic' else ' (simplified network)'}: {tcn2.n_vars} interval variables x {tcn2.n_props} ternary propagators",
                               "fixpoint": fp, "steps": args.other_steps, "nodes_total": b2, "nodes_per_sec": g2["nodes"] / e2, "propagations_per_sec": g2["num_deductions"] / e2,
                               "active_lane_evaluations_per_sec": g2["active_lane_evaluations"] / e2, "ms_per_step": e2 * 1000.0 / args.other_steps,
                               "workgroups": l2["num_blocks"], "threads": l2["threads_per_block"], "memory": capi.MEM_KINDS[l2["mem_kind"]], "lds_bytes_per_workgroup": l2["shared_bytes"],
                               "subproblems_power": pl2["subproblems_power"], "kernel_opt": pl2["kernel_opt"], "roofline": roof2, "load_and_simplify_s": t_load})
                sess2.close()

    # The sharded search itself, in the DEFAULT command: the node-budget steps above measure N kernels' node rate -- on the headline instance no
    # subproblem completes inside a step, so queue, subtree skips, stealing never enter `value`.  This record is a WHOLE search of fixed total work: the proof that
    # `objective <= B` is infeasible (B below the optimum, a constant constraint: no incumbent travels, the tree is the same at every N), every one of the 2^d
    # subproblems solved or skipped exactly once whichever GPU takes it.  Its `seconds` at N = 1, 2, 4, 8 IS the strong-scaling curve of the sharded search.
    sharded = None
    if args.sharded_search and SOLVE_DEFAULTS.get(args.workload, (None,))[0] is not None and args.fixpoint == "event":
        session.close()  # one persistent search at a time
        d_bound, _, d_power = SOLVE_DEFAULTS[args.workload]
        runner = make_search_runner(args, main_tcn, rank, world, local_rank, dist, tdev, capi, tdist, torch, d_power)
        runner(dict(use_fixed_bound=1, fixed_bound=int(d_bound)), None)  # warm-up (code object, allocations)
        reps = [runner(dict(use_fixed_bound=1, fixed_bound=int(d_bound)), None) for _ in range(max(1, args.sharded_reps))]
        sharded = min(reps, key=lambda r: r["seconds"])
        sharded.update({"fixed_bound": int(d_bound), "what": PROOF_WHAT, "seconds_all_runs": [r["seconds"] for r in reps], "scaling": "strong",
                        "metric": "seconds of the whole sharded search that refutes objective <= B (fixed total work at every N; lower is better)"})

    if rank == 0:
        kernel_s = tot["kernel_ns"] * 1e-9 / steps                 # average launch duration of solve_kernel (HIP events on its stream, rank 0)
        props = tot["num_deductions"] / steps                     # propagations of one launch, rank 0
        writes = tot["store_writes"] / steps
        info = capi.device_info(local_rank)
        roof = roofline_record(info, last["mem_kind"], props, writes, kernel_s)
        roof["note"] = ("integer propagation: no MFMA.  achieved = algorithmic bytes of the level named in `bound` / average launch duration "
                        "(HIP events around the launch, this run).  The kernel is instruction-issue bound, not memory bound: see `issue`.")
        attach_counters(roof, args.workload, args.fixpoint, kernel_s)
        out = {
            "metric": "propagations/sec (+ nodes/sec) on wordpress7_500.fzn" if args.workload == "wordpress7_500" else f"propagations/sec (+ nodes/sec) on {fzn}",
            "value": glob["num_deductions"] / elapsed, "unit": "propagations/s",
            "nodes_per_sec": glob["nodes"] / elapsed,
            # `value` counts wave iterations x 64 as the reference does (barebones:958-960); this is the same count without the idle lanes of partly
            # filled slices (class padding, the network's last slice)
            "active_lane_evaluations_per_sec": glob["active_lane_evaluations"] / elapsed,
            "evaluations_per_node": {"counted_x64": glob["num_deductions"] / max(1, glob["nodes"]), "active_lanes": glob["active_lane_evaluations"] / max(1, glob["nodes"])},
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1000.0 / steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32",
            "data": "synthetic (seed 42)" if args.workload == "synthetic" else "reference instance file (no randomness)",
            "config": {"workload": f"{fzn}{'' if args.no_simplify or args.workload == 'synthetic' else ' (simplified network)'}: {tcn.n_vars} interval variables x {tcn.n_props} ternary propagators, "
                                   f"{last['num_blocks']} workgroups x {last['threads_per_block']} threads per GPU, "
                                   f"{capi.MEM_KINDS[last['mem_kind']]} ({last['shared_bytes']} B LDS per workgroup), "
                                   f"2^{plan['subproblems_power']} subproblems, one step = {budget} nodes of the search by all GPUs together"
                                   f"{'' if not args.cutnodes else f', at most {args.cutnodes} per workgroup'}, fixpoint={args.fixpoint}, leaf rule of -arch {args.leaf_rule}",
                       "leaf_rule": args.leaf_rule,
                       "parallelism": f"eps_block_cyclic{world}" + ("" if world == 1 else ("+xgmi_steal" if linked else "+host_relay"))},
            "roofline": roof,
        }
        if world > 1:
            ks = [r["kernel_ms"] for r in per_rank]
            out["multi_gpu"] = {"exchange": "peer cells over xGMI" if linked else "host relay",
                                "dist": {"backend": dist.get_backend(), "world_size": dist.get_world_size()}, "per_rank": per_rank,
                                "kernel_ms_max_over_mean": max(ks) / max(1e-9, sum(ks) / len(ks)),
                                "eps_solved_per_step": glob["eps_solved_subproblems"] / steps, "eps_skipped_per_step": glob["eps_skipped_subproblems"] / steps,
                                "stolen_per_step": glob["eps_stolen_subproblems"] / steps, "start_skew_ms_max": glob.get("start_skew_ms_max"),
                                # step time is the slowest rank's (what `value` divides by); the rate the kernels sustained while they ran:
                                "nodes_per_sec_sum_over_kernel_time": glob["nodes"] / max(1e-9, glob["kernel_ns"] * 1e-9 / world),
                                "note": "fixed node budget for all GPUs together (every device counts its own nodes, its poller folds them into rank 0's cell once per poll period); wait_share = workgroup-time without a subproblem"}
        else:
            out["balance"] = per_rank[0]
        for side in sides:
            out[f"{side['fixpoint']}_mode"] = side
        if others:
            out["other_workloads"] = others
        if sharded is not None:
            out["sharded_search"] = sharded
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tcn, plan["subproblems_power"], args.cpu_seconds, args.leaf_rule)
            out["speedup_vs_cpu_baseline"] = {"propagations": out["value"] / max(out["cpu_baseline"]["value"], 1e-9),
                                              "nodes": out["nodes_per_sec"] / max(out["cpu_baseline"]["nodes_per_sec"], 1e-9)}
            ac1 = next((sd for sd in sides if sd["fixpoint"] == "ac1"), None)
            if ac1 is not None:  # the same fixpoint loop on both sides (`-fp ac1` on the GPU, AC1 on the CPU)
                out["speedup_vs_cpu_baseline"]["ac1_like_for_like"] = {"propagations": ac1["propagations_per_sec"] / max(out["cpu_baseline"]["value"], 1e-9),
                                                                       "nodes": ac1["nodes_per_sec"] / max(out["cpu_baseline"]["nodes_per_sec"], 1e-9),
                                                                       "note": f"same fixpoint loop (AC1) and same leaf rule (-arch {args.leaf_rule}) on both sides; the GPU walks its "
                                                                               "subproblems in parallel, the CPU in index order"}
        if world == 1 and args.workload == "wordpress7_500" and args.reference_seconds > 0:
            session.close()  # the CLI run gets the whole GPU
            out["reference_invocation"] = reference_invocation(args.reference_seconds)
        print(json.dumps(out), flush=True)
    session.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
