#!/usr/bin/env python3
"""Debug aid: repeat one configuration of one instance and count the runs that were not exhaustive.  usage: dbg_flaky.py rel n [debug bits]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from turbo_amd import capi, frontend
rel, n = sys.argv[1], int(sys.argv[2])
bits = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x100000
tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", rel))
bad = []
for i in range(n):
    has, best, st = capi.solve(tcn, capi.make_config(timeout_ms=60000, fixpoint=2, debug=bits))
    if not st["exhaustive"] or st["why_not_exhaustive"]:
        bad.append((i, st["why_not_exhaustive"], st["debug_slice"], st["nodes"]))
print(os.environ.get("TURBO_HIP_LIB", "production"), "TB_NO_LEAN" in os.environ, hex(bits), "bad runs:", len(bad), "of", n, bad[:5], flush=True)
