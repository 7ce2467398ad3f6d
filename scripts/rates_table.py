#!/usr/bin/env python3
"""Throughput of every fixpoint mode on the benchmark configurations (fixed node budget per workgroup), as a table.

usage: python scripts/rates_table.py [--out profiles/r01_rates.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
from turbo_amd.synth import make_synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
a = ap.parse_args()
rows = []
work = [("wordpress7_500", "example_wordpress7_500.fzn", 3000), ("accap_a3", "accap_a3.fzn", 4000), ("trains15", "trains15.fzn", 2000), ("synthetic 100k x 500k", None, 40)]
modes = [("wac1", dict(fixpoint=1)), ("ac1", dict(fixpoint=0)), ("wac1 + entailed removal", dict(fixpoint=1, entailed_prop_removal=1)), ("event", dict(fixpoint=2))]
for name, fzn, cut in work:
    tcn = make_synthetic(100_000, 500_000, seed=42) if fzn is None else preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", fzn))[1]
    for mode, kw in modes:
        cfg = capi.make_config(timeout_ms=120000, stop_after_n_nodes=cut, **kw)
        for _ in range(2):
            has, sol, st = capi.solve(tcn, cfg)
        secs = st["kernel_ns"] * 1e-9
        row = {"instance": name, "variables": tcn.n_vars, "propagators": tcn.n_props, "mode": mode, "workgroups": st["num_blocks"], "threads": st["threads_per_block"],
               "memory": capi.MEM_KINDS[st["mem_kind"]], "cutnodes": cut, "propagations_per_s": st["num_deductions"] / secs, "nodes_per_s": st["nodes"] / secs,
               "propagations_per_node": st["num_deductions"] / max(1, st["nodes"]), "kernel_ms": secs * 1e3}
        rows.append(row)
        print(f"{name:22s} {mode:24s} {row['workgroups']:5d}x{row['threads']:<4d} {row['memory']:12s} {row['propagations_per_s']:.3e} props/s {row['nodes_per_s']:.3e} nodes/s "
              f"{row['propagations_per_node']:.0f} props/node", flush=True)
if a.out:
    json.dump({"note": "python scripts/rates_table.py on one MI355X; simplified networks; fixed node budget per workgroup", "rows": rows}, open(a.out, "w"), indent=1)
