#!/usr/bin/env python3
"""Slice runs, wave iterations and rounds per node on the bench workload (node budget for the whole GPU), tuning build:
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/bench_pop_probe.py [instance] [nodes_total]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
res = {}
for bits in (0, 0x400000):
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=budget, timeout_ms=120000, debug=bits)
    for _ in range(2):
        has, best, st = capi.solve(tcn, cfg)
    res[bits] = st
a, b = res[0], res[0x400000]
n = a["nodes"]
print(f"{name}: {n} nodes, {n / (a['kernel_ns'] * 1e-9):.3e} nodes/s; per node: {a['num_deductions'] / 64 / n:.1f} wave iterations, "
      f"{b['num_deductions'] / 64 / b['nodes']:.1f} slice runs, {a['fixpoint_iterations'] / n - 1:.2f} rounds, {a['store_writes'] / n:.1f} bound writes, "
      f"fails {a['fails'] / n:.2f}, solutions {a['solutions']}")
