#!/usr/bin/env python3
"""Calibration of bench.py --mode solve (SOLVE_DEFAULTS): for each instance, how long does the canonical pass under `objective <= B` run on one GPU, for a ladder of B --
and is B refuted (exhaustive, no solution) or satisfied -- and what does a few seconds of branch and bound reach.
usage: python3 scripts/r05_solve_probe.py [seconds per run] > gpurun_out/r05_solve_probe.log"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
limit = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
LADDER = {"example_wordpress7_500.fzn": ([2000, 4000, 6000, 8000, 9000, 10000, 11000], -1),
          "trains15.fzn": ([40, 60, 80, 90, 100, 105, 110], -1),
          "accap_a3.fzn": ([], 12)}
for name, (ladder, power) in LADDER.items():
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    t0 = time.perf_counter()
    has, best, st = capi.solve(tcn, capi.make_config(fixpoint=2, timeout_ms=int(limit * 1000), subproblems_power=power))
    inc = int(best[tcn.obj_var]["lb"]) if has else None
    print(json.dumps({"instance": name, "run": "branch and bound", "seconds": time.perf_counter() - t0, "exhaustive": st["exhaustive"], "incumbent_internal": inc,
                      "objective": int(tcn.objective_of(best)) if has else None, "nodes": st["nodes"], "solved": st["eps_solved_subproblems"], "skipped": st["eps_skipped_subproblems"],
                      "sub_power": st["subproblems_power"]}), flush=True)
    if not ladder and inc is not None:  # no ladder given: fractions of what branch and bound reached
        ladder = sorted({int(inc * f) for f in (0.5, 0.7, 0.8, 0.9, 0.95)} | {inc - 1})
    for B in ladder:
        t0 = time.perf_counter()
        has, best, st = capi.solve(tcn, capi.make_config(fixpoint=2, timeout_ms=int(limit * 1000), use_fixed_bound=1, fixed_bound=B, subproblems_power=power))
        print(json.dumps({"instance": name, "run": f"objective <= {B}", "seconds": time.perf_counter() - t0, "kernel_s": st["kernel_ns"] * 1e-9, "exhaustive": st["exhaustive"], "has_solution": bool(has),
                          "nodes": st["nodes"], "solved": st["eps_solved_subproblems"], "skipped": st["eps_skipped_subproblems"], "sub_power": st["subproblems_power"],
                          "first_idle_s": st["min_block_ns"] * 1e-9, "last_block_s": st["max_block_ns"] * 1e-9}), flush=True)
