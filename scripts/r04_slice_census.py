#!/usr/bin/env python3
"""The slices that run most (tuning build): TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so TB_DUMP_SLICES=1 python3 scripts/r04_slice_census.py [instance]
Prints the per-slice run census of a 3 M-node search (engine.hip: slice_census) next to the slice table (TB_DUMP_SLICES)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "trains15.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
print("=== search", flush=True)
sys.stderr.write("=== search\n"); sys.stderr.flush()
cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=3_000_000, timeout_ms=120000, debug=0x400000, verbose=1)
has, best, st = capi.solve(tcn, cfg)
print("nodes", st["nodes"], "runs/node", st["num_deductions"] / 64 / st["nodes"])
