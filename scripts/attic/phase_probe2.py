#!/usr/bin/env python3
"""Where a workgroup's time goes per node (in-kernel wall-clock timers of thread 0; tuning build for the fine ones):
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/phase_probe2.py [instance] [fixpoint]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
fp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
T = capi.TIMERS
for bits in (0, 0x10000):
    cfg = capi.make_config(fixpoint=fp, stop_after_n_nodes=4000, timeout_ms=120000, debug=bits)
    for _ in range(2):
        has, best, st = capi.solve(tcn, cfg)
    n, B = st["nodes"], st["num_blocks"]
    secs = st["kernel_ns"] * 1e-9
    per = {T[i]: st["timers_ns"][i] / n / 1e3 for i in range(len(T))}
    per.update({"prof_" + k.lower(): v / n / 1e3 for k, v in zip(capi.PROF, st["prof_ns"])})  # the engine's own phases (tuning build)
    print(f"{name} fp={fp} bits={bits:#x}: {n/secs:.3e} nodes/s, {B} workgroups, {secs*1e6*B/n:.1f} us per node per workgroup, rounds/node {st['fixpoint_iterations']/n:.1f}")
    print("   us per node: " + ", ".join(f"{k}={v:.1f}" for k, v in per.items() if k not in ("OVERALL", "LATEST_BEST_OBJ_FOUND", "FIRST_BLOCK_IDLE")))
