#!/usr/bin/env python3
"""Time to prove optimality (whole search, no node budget) per fixpoint mode on the headline instances.
usage: python scripts/proof_probe.py [--out file.json] [instances...]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--timeout", type=int, default=120000)
ap.add_argument("--modes", default="event,wac1")
ap.add_argument("instances", nargs="*", default=["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"])
a = ap.parse_args()
rows = []
for name in a.instances:
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    for mode in a.modes.split(","):
        fp = {"ac1": 0, "wac1": 1, "event": 2}[mode]
        for rep in range(2):
            t0 = time.perf_counter()
            has, best, st = capi.solve(tcn, capi.make_config(fixpoint=fp, timeout_ms=a.timeout))
            wall = time.perf_counter() - t0
            secs = st["kernel_ns"] * 1e-9
            row = {"instance": name, "mode": mode, "rep": rep, "exhaustive": st["exhaustive"], "objective": int(tcn.objective_of(best)) if has else None,
                   "kernel_s": secs, "wall_s": wall, "nodes": st["nodes"], "props": st["num_deductions"], "blocks": st["num_blocks"], "sub_power": st["subproblems_power"],
                   "nodes_per_s": st["nodes"] / secs, "props_per_s": st["num_deductions"] / secs, "blocks_done": st["num_blocks_done"],
                   "first_idle_s": st["timers_ns"][10] * 1e-9, "solved": st["eps_solved_subproblems"], "skipped": st["eps_skipped_subproblems"]}
            rows.append(row)
            print(json.dumps(row), flush=True)
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
