"""One-off: many random models, GPU tree vs oracle tree (see tests/test_gpu_parity.py::test_random_models_tree_identical)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_models import random_model
from oracle import pyoracle
from turbo_amd import capi, frontend
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    tcn = frontend.Model.from_string(random_model(seed)).tcn()
    has_o, best_o, st_o = pyoracle.solve(tcn, subproblems_power=0)
    exp, efailed, eent, _, _ = pyoracle.propagate(tcn.store, tcn.props)
    for fp, dbg in ((1, 0), (2, 0), (2, 0x100000), (0, 0)):
        got, failed, ent, _, _, _ = capi.propagate(tcn.props, tcn.store[None, :], capi.make_config(fixpoint=fp, debug=dbg))
        ok = bool(failed[0]) == efailed and (efailed or (bool(ent[0]) == eent and np.array_equal(got[0], exp)))
        has_g, best_g, st_g = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=0, timeout_ms=60000, fixpoint=fp, debug=dbg))
        ok = ok and has_g == has_o and all(st_g[k] == st_o[k] for k in ("nodes", "fails", "solutions", "depth_max", "exhaustive"))
        ok = ok and (not has_o or np.array_equal(best_g, best_o))
        # whole chip, parallel: same status and optimum
        has_p, best_p, st_p = capi.solve(tcn, capi.make_config(timeout_ms=60000, fixpoint=fp, debug=dbg))
        ok = ok and has_p == has_o and (tcn.goal == 0 or not has_o or tcn.objective_of(best_p) == tcn.objective_of(best_o))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "fp", fp, hex(dbg), flush=True)
print("seeds", lo, hi, "mismatches", bad)
