#!/usr/bin/env python3
"""Wave 0's time inside the rounds of the event fixpoint, split into record fetch / slice body / successor marks / barrier
(tuning build, knob 0x10000; printed by the engine on stderr):
TURBO_HIP_LIB=turbo_amd/lib/libturbo_hip_tuning.so python scripts/event_profile_gpu.py [instance]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
for bits in (0, 0x10000):
    cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=6_000_000, timeout_ms=120000, debug=bits, verbose=1 if bits else 0)
    for _ in range(2):
        has, best, st = capi.solve(tcn, cfg)
    print(f"{name} bits={bits:#x}: {st['nodes'] / (st['kernel_ns'] * 1e-9):.4e} nodes/s, {st['num_blocks']} workgroups", flush=True)
