#!/usr/bin/env python3
"""Slice runs and wave iterations per node BY CLASS on the bench workload (tuning build):
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/class_iters_probe.py [instance] [nodes_total]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
names = ["HEAVY", "ADD", "MIN", "MAX", "EQ_R", "LEQ_R", "EQ_T", "EQ_F", "LEQ_T", "LEQ_F", "mixed"]
for c, cname in enumerate(names):
    row = []
    for bits in (0, 0x400000, 0x400040):
        cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=budget, timeout_ms=120000, debug=bits | ((c + 1) << 28))
        has, best, st = capi.solve(tcn, cfg)
        row.append(st["num_deductions"] / 64 / st["nodes"])
    if row[1] > 0:
        print(f"{name} {cname:6s}: {row[1]:7.1f} runs/node ({row[2]:.1f} of them narrow nothing), {row[0]:7.1f} iterations/node, {row[0] / row[1]:.2f} iterations per run", flush=True)
