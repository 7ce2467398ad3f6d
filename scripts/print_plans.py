#!/usr/bin/env python3
"""The launch plan of bench.py's sessions for the three FlatZinc workloads (what tests/test_gpu_fullgrid_paths.py asserts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
for name in ("example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"):
    _, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
    s = capi.Session(tcn, capi.make_config(fixpoint=2, stop_after_n_nodes_total=1000, timeout_ms=60000))
    print(name, s.plan(), flush=True)
    s.close()
