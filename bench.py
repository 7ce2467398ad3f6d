#!/usr/bin/env python3
"""Benchmark of the dive-and-solve hot path (BASELINE.json metric: propagations/sec + nodes/sec on
wordpress7_500.fzn at 1/2/4/8 MI355X).

A "step" is one launch of the persistent search kernel over the whole EPS index space with a fixed
node budget per workgroup (`-cutnodes`, the reference's own fixed-work switch, config.cpp:155), inputs
already resident in HBM (tb_session_create uploads them before the timed region).
N > 1: one process per GPU (torch.distributed / RCCL); the 2^d subproblems are sharded in contiguous
slices, the only payload exchanged during a step is the incumbent objective bound (all_reduce MIN of
one int32).  By default the node budget of a step is fixed and divided among the GPUs ("strong" scaling, the
north star's target); `--scaling weak` gives every GPU the full per-workgroup budget.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    "wordpress7_500": ("example_wordpress7_500.fzn", 3000),
    "accap_a3": ("accap_a3.fzn", 4000),
    "trains15": ("trains15.fzn", 2000),
    "synthetic": ("synthetic 100k x 500k (seed 42)", 40),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PROPAGATION = 40  # 16 B bytecode + 3 x 8 B domains; + 8 B per narrowed bound written


def cpu_baseline(tcn, seconds: float) -> dict:
    """The oracle (CPU restatement of cpu_solving.hpp) on a bounded sample of the same workload, 1 core."""
    from oracle import pyoracle
    has, best, st = pyoracle.solve(tcn, subproblems_power=0, timeout_ms=int(seconds * 1000))
    secs = max(st["solve_seconds"], 1e-9)
    return {"value": st["num_deductions"] / secs, "unit": "propagations/s", "cores": 1, "kind": "port",
            "nodes_per_sec": st["nodes"] / secs,
            "sample": f"first {secs:.1f} s of the sequential DFS branch-and-bound (AC1 Gauss-Seidel) on the same instance: "
                      f"{st['nodes']} nodes, {st['num_deductions']} propagations",
            "host_cpus": os.cpu_count()}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="wordpress7_500", choices=sorted(WORKLOADS))
    ap.add_argument("--cutnodes", type=int, default=0, help="node budget per workgroup and step (0 = workload default)")
    ap.add_argument("--fixpoint", default="wac1", choices=["ac1", "wac1", "event"])
    ap.add_argument("--event-steps", type=int, default=2, help="extra steps in the event-driven fixpoint mode, reported beside the headline (0 = skip)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the node budget of one step is fixed (cutnodes x workgroups of one GPU) and divided among the GPUs; "
                         "weak: every GPU gets the full per-workgroup budget")
    ap.add_argument("--no-simplify", action="store_true", help="skip the network simplifier (the reference's -disable_simplify)")
    ap.add_argument("--debug-bits", type=lambda v: int(v, 0), default=0, help="tuning knobs of tb_config.reserved[0] (experiments only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="testing aid: gloo lets two ranks share one GPU")
    ap.add_argument("--share-device", action="store_true", help="testing aid: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    import torch
    from turbo_amd import capi, frontend

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and the engine has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    tdev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)  # "nccl" is RCCL over xGMI

    fzn, default_cut = WORKLOADS[args.workload]
    cut_total = args.cutnodes or default_cut      # per workgroup at N = 1
    cut = cut_total if args.scaling == "weak" else max(1, cut_total // max(world, 1))
    if args.workload == "synthetic":
        from turbo_amd.synth import make_synthetic
        tcn = make_synthetic(100_000, 500_000, seed=42)
    elif args.no_simplify:
        tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", fzn))
    else:
        # the reference's default pipeline: root fixpoint (tb_propagate on this GPU) + network simplifier, outside the timed region
        from turbo_amd import preprocess
        _model, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", fzn), device=local_rank)
    fp_code = {"ac1": 0, "wac1": 1, "event": 2}
    cfg = capi.make_config(fixpoint=fp_code[args.fixpoint], stop_after_n_nodes=cut, timeout_ms=600000,
                           device=local_rank, rank=rank, world_size=world, debug=args.debug_bits)
    session = capi.Session(tcn, cfg)  # inputs uploaded to HBM here, outside the timed region

    from turbo_amd.distributed import exchange_until_done

    def one_step(sess=None) -> dict:
        sess = sess or session
        sess.start()
        # incumbent exchange (all_reduce MIN of one int32 over RCCL) until every rank's kernel is done
        exchange_until_done(sess, dist if world > 1 else None, tensor_device=tdev, period_s=0.0002)
        _, _, st = sess.finish()
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    tot = {"nodes": 0, "num_deductions": 0, "store_writes": 0, "kernel_ns": 0, "fixpoint_iterations": 0}
    last = None
    for _ in range(args.steps):
        last = one_step()
        for k in tot:
            tot[k] += last[k]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([tot["nodes"], tot["num_deductions"], tot["store_writes"]], dtype=torch.int64, device=tdev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        g_nodes, g_props, g_writes = (int(x) for x in agg.tolist())
    else:
        g_nodes, g_props, g_writes = tot["nodes"], tot["num_deductions"], tot["store_writes"]

    # the same workload in the engine's event-driven fixpoint (same tree, fewer propagator evaluations per node)
    event = None
    if args.event_steps > 0 and args.fixpoint != "event":
        cfg_e = capi.make_config(fixpoint=2, stop_after_n_nodes=cut * 2, timeout_ms=600000, device=local_rank, rank=rank, world_size=world)
        sess_e = capi.Session(tcn, cfg_e)
        one_step(sess_e)
        sync()
        te = time.perf_counter()
        acc = {"nodes": 0, "num_deductions": 0}
        for _ in range(args.event_steps):
            st_e = one_step(sess_e)
            for k in acc:
                acc[k] += st_e[k]
        sync()
        te = time.perf_counter() - te
        if world > 1:
            t = torch.tensor([te], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            te = float(t.item())
            agg = torch.tensor([acc["nodes"], acc["num_deductions"]], dtype=torch.int64, device=tdev)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            acc["nodes"], acc["num_deductions"] = (int(x) for x in agg.tolist())
        event = {"fixpoint": "event", "steps": args.event_steps, "cutnodes": cut * 2, "nodes_per_sec": acc["nodes"] / te,
                 "propagations_per_sec": acc["num_deductions"] / te,
                 "note": "same search tree, propagators re-evaluated only when one of their variables was narrowed"}
        sess_e.close()

    if rank == 0:
        steps = max(args.steps, 1)
        kernel_s = tot["kernel_ns"] * 1e-9 / steps                 # average launch duration of solve_kernel (HIP events, rank 0)
        alg_bytes = (tot["num_deductions"] * BYTES_PER_PROPAGATION + tot["store_writes"] * 8) / steps  # per launch, rank 0
        achieved = alg_bytes / max(kernel_s, 1e-12) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                rec = json.load(open(tfile))
                if (rec.get("workload") == args.workload and rec.get("cutnodes") == cut and rec.get("fixpoint") == args.fixpoint
                        and bool(rec.get("simplified", False)) == (not args.no_simplify and args.workload != "synthetic")):
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        issue = None  # what really bounds the kernel: SQ instruction-issue counters of the same command (profiles/)
        sfile = os.path.join(ROOT, "profiles", "r01_sq_summary.json")
        if os.path.exists(sfile) and args.workload == "wordpress7_500":
            try:
                k = json.load(open(sfile))["kernels"].get(args.fixpoint)
                if k:
                    issue = {key: k[key] for key in ("valu_busy", "salu_busy", "lds_busy", "valu_insts_per_64_propagations")}
                    issue["source"] = "profiles/r01_sq_summary.json (rocprofv3 --pmc SQ_* passes of this command)"
            except Exception:
                issue = None
        out = {
            "metric": "propagations/sec (+ nodes/sec) on wordpress7_500.fzn" if args.workload == "wordpress7_500" else f"propagations/sec (+ nodes/sec) on {fzn}",
            "value": g_props / elapsed, "unit": "propagations/s",
            "nodes_per_sec": g_nodes / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1000.0 / steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32",
            "data": "synthetic (seed 42)" if args.workload == "synthetic" else "reference instance file (no randomness)",
            "config": {"workload": f"{fzn}{'' if args.no_simplify or args.workload == 'synthetic' else ' (simplified network)'}: {tcn.n_vars} interval variables x {tcn.n_props} ternary propagators, "
                                   f"{last['num_blocks']} workgroups x {last['threads_per_block']} threads per GPU, "
                                   f"{capi.MEM_KINDS[last['mem_kind']]} ({last['shared_bytes']} B LDS per workgroup), "
                                   f"2^{last['subproblems_power']} subproblems, cutnodes={cut} per workgroup and step"
                                   f"{'' if world == 1 else f' ({cut_total} at 1 GPU: fixed total node budget)' if args.scaling == 'strong' else ''}, fixpoint={args.fixpoint}",
                       "parallelism": f"eps_shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "tb::solve_kernel", "avg_launch_ms": kernel_s * 1000.0,
                         "note": "algorithmic bytes = 40 B x propagations + 8 B x narrowed bounds; the store is LDS-resident on this "
                                 "workload, so the figure prices LDS+L2 traffic against the HBM peak (see DESIGN.md)"},
        }
        if last["mem_kind"] != 0:
            # with the store in LDS the HBM fraction says little; the 24 B of domain gathers per propagation are
            # LDS traffic, priced against 256 B/clk/CU (ds_read_b64, MI355X_MICROARCH.md LDS section) x CUs x shader clock
            info = capi.device_info(local_rank)
            lds_peak = info["compute_units"] * 256.0 * info["clock_khz"] * 1e3 / 1e9
            lds_ach = (tot["num_deductions"] * 24 / steps) / max(kernel_s, 1e-12) / 1e9
            out["roofline"]["lds"] = {"achieved": lds_ach, "peak": lds_peak, "unit": "GB/s", "frac": lds_ach / lds_peak,
                                      "note": "3 x 8 B domain gathers per propagation served by LDS"}
        if issue is not None:
            out["roofline"]["instruction_issue"] = issue
        if event is not None:
            out["event_mode"] = event
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tcn, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-9)
        print(json.dumps(out), flush=True)
    session.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
