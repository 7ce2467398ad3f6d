#!/usr/bin/env python3
"""Cost of the phases of the event-driven fixpoint by DOUBLING them (tuning build): each knob repeats an idempotent phase, so the
search is unchanged and the slowdown is that phase's share of the time.
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/ablate_event.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", "example_wordpress7_500.fzn"))
base = None
for name, bits in (("baseline", 0), ("marks x2", 1), ("snapshot copy x2", 4), ("one more iteration per run", 8), ("baseline again", 0)):
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=6_000_000, timeout_ms=120000, debug=bits)
    best = 0
    for _ in range(3):
        has, sol, st = capi.solve(tcn, cfg)
        best = max(best, st["nodes"] / (st["kernel_ns"] * 1e-9))
    base = base or best
    print(f"{name:30s} {best:.4e} nodes/s  ({(base / best - 1) * 100:+.1f} % time)", flush=True)
