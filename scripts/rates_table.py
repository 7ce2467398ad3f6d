#!/usr/bin/env python3
"""Throughput of every fixpoint mode on the benchmark configurations (node budget for the whole GPU, like bench.py), as a table.

usage: python scripts/rates_table.py [--out profiles/r02_rates.json]
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
work = [("wordpress7_500", "example_wordpress7_500.fzn", 48_000_000, 3_000_000), ("accap_a3", "accap_a3.fzn", 48_000_000, 24_000_000),
        ("trains15", "trains15.fzn", 24_000_000, 8_000_000), ("synthetic 100k x 500k", None, 8000, 8000)]
modes = [("event", dict(fixpoint=2), 0), ("event, plain store (no compact layout)", dict(fixpoint=2, debug=0x80000), 0),
         ("event, compact layout with the constants kept in the slab", dict(fixpoint=2, debug=0x40000000), 0),
         ("wac1", dict(fixpoint=1), 1), ("wac1 on the compact layout (opt-in)", dict(fixpoint=1, debug=0x100000), 1), ("ac1", dict(fixpoint=0), 1),
         ("wac1 + entailed removal", dict(fixpoint=1, entailed_prop_removal=1), 1)]
for name, fzn, b_event, b_sweep in work:
    tcn = make_synthetic(100_000, 500_000, seed=42) if fzn is None else preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", fzn))[1]
    for mode, kw, sweep in modes:
        cfg = capi.make_config(timeout_ms=120000, stop_after_n_nodes_total=b_sweep if sweep else b_event, **kw)
        for _ in range(2):
            has, sol, st = capi.solve(tcn, cfg)
        secs = st["kernel_ns"] * 1e-9
        row = {"instance": name, "variables": tcn.n_vars, "propagators": tcn.n_props, "mode": mode, "workgroups": st["num_blocks"], "threads": st["threads_per_block"],
               "memory": capi.MEM_KINDS[st["mem_kind"]], "lds_bytes": st["shared_bytes"], "nodes": st["nodes"], "propagations_per_s": st["num_deductions"] / secs, "nodes_per_s": st["nodes"] / secs,
               "propagations_per_node": st["num_deductions"] / max(1, st["nodes"]), "kernel_ms": secs * 1e3}
        rows.append(row)
        print(f"{name:22s} {mode:58s} {row['workgroups']:5d}x{row['threads']:<4d} {row['memory']:12s} {row['propagations_per_s']:.3e} props/s {row['nodes_per_s']:.3e} nodes/s "
              f"{row['propagations_per_node']:.0f} props/node", flush=True)
if a.out:
    json.dump({"note": "python scripts/rates_table.py on one MI355X; simplified networks; node budget for the whole GPU (stop_after_n_nodes_total), second of two runs", "rows": rows}, open(a.out, "w"), indent=1)
