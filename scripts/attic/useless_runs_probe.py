#!/usr/bin/env python3
"""Slice runs per node by class, and how many of them narrowed nothing (tuning build: 0x400000 counts runs, bits 28-31 select
1 + class, 0x40 keeps only the runs that narrowed nothing):
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/useless_runs_probe.py [instance]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
CLASSES = ["heavy", "add", "min", "max", "eq_r", "leq_r", "eq_t", "eq_f", "leq_t", "leq_f", "mixed"]
def runs(bits):
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=3_000_000, timeout_ms=120000, debug=bits)
    has, best, st = capi.solve(tcn, cfg)
    return st["num_deductions"] / 64.0 / max(1, st["nodes"])
print(f"{name}: all classes: {runs(0x400000):.1f} runs per node, {runs(0x400040):.1f} of them narrowed nothing")
for c, cname in enumerate(CLASSES):
    a = runs(0x400000 | ((c + 1) << 28))
    if a > 0.05:
        print(f"  {cname:6s} {a:6.1f} runs per node, {runs(0x400040 | ((c + 1) << 28)):6.1f} narrowed nothing")
