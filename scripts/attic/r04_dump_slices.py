#!/usr/bin/env python3
"""Slice table of an instance as the event kernels see it (TB_DUMP_SLICES): python3 scripts/r04_dump_slices.py [instance]"""
import os, sys
os.environ["TB_DUMP_SLICES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from turbo_amd import capi, preprocess
name = sys.argv[1] if len(sys.argv) > 1 else "trains15.fzn"
_, tcn, _ = preprocess.load_fzn_simplified(os.path.join(ROOT, "benchmarks", name))
cfg = capi.make_config(fixpoint=2, stop_after_n_nodes_total=100000, timeout_ms=60000)
has, best, st = capi.solve(tcn, cfg)
print("nodes", st["nodes"])
