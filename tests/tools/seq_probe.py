import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from turbo_amd import frontend, capi
from oracle import pyoracle
for rel in ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "test_data/sudoku_opt_p0.fzn"]:
    tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", rel))
    for d in (0, 4, 12, 19):
        ho, bo, so = pyoracle.solve(tcn, subproblems_power=d)
        hg, bg, sg = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=d, timeout_ms=120000, fixpoint=2))
        keys = ("nodes", "fails", "solutions", "depth_max", "eps_skipped_subproblems", "exhaustive")
        ok = all(sg[k] == so[k] for k in keys) and np.array_equal(bg, bo)
        print(rel, d, "OK" if ok else f"MISMATCH gpu={[sg[k] for k in keys]} orc={[so[k] for k in keys]} why={sg['why_not_exhaustive']}", flush=True)
