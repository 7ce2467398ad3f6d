#!/usr/bin/env python3
"""One search launch of a bench workload for a PC-sampling run (scripts/r05_pc_sampling.sh): wordpress7_500 simplified, event fixpoint, a node budget.
    python3 scripts/pcs_worker.py [workload] [nodes] [fixpoint]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "example_wordpress7_500.fzn"
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
fp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if name == "synthetic":
    from turbo_amd.synth import make_synthetic
    tcn = make_synthetic(100_000, 500_000, seed=42)
else:
    tcn = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))[1]
t0 = time.time()
has, best, st = capi.solve(tcn, capi.make_config(fixpoint=fp, stop_after_n_nodes_total=nodes, timeout_ms=60000))
dt = time.time() - t0
print(f"{name}: {st['nodes']} nodes in {dt:.2f} s ({st['nodes'] / dt:.3e} nodes/s, host clock), kernel {st.get('kernel_ms', 0):.1f} ms", flush=True)
