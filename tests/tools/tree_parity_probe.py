import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from turbo_amd import frontend, capi
from oracle import pyoracle
for rel in ["test_data/pat2.fzn", "test_data/sudoku_opt_p0.fzn", "test_data/pennies5.fzn", "test_data/pat9.fzn", "test_data/bug4.fzn"]:
    tcn = frontend.load_fzn("benchmarks/" + rel)
    ho, bo, so = pyoracle.solve(tcn, subproblems_power=4)
    for T in (256, 512, 1024):
        for gm in (0, 1):
            hg, bg, sg = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=4, timeout_ms=60000, fixpoint=2, threads_per_block=T, only_global_memory=gm))
            ok = all(sg[k] == so[k] for k in ("nodes", "fails", "solutions", "depth_max")) and np.array_equal(bg, bo)
            print(rel, T, gm, "OK" if ok else f"MISMATCH gpu={[sg[k] for k in ('nodes','fails','solutions')]} orc={[so[k] for k in ('nodes','fails','solutions')]}")
# wordpress: compare node counts for a fixed single-block cutnodes run across configs (same tree => same stats)
tcn = frontend.load_fzn("benchmarks/example_wordpress7_500.fzn")
ref = None
for T in (256, 512, 1024):
    for gm in (0, 1):
        for fp in (1, 2):
            h, b, s = capi.solve(tcn, capi.make_config(or_nodes=1, subproblems_power=6, stop_after_n_nodes=300, timeout_ms=60000, fixpoint=fp, threads_per_block=T, only_global_memory=gm))
            key = (s["nodes"], s["fails"], s["solutions"], s["depth_max"], s["best_bound"])
            ref = ref or key
            print("wordpress", T, gm, fp, key, "OK" if key == ref else "MISMATCH", "sweeps/node", s["fixpoint_iterations"] / s["nodes"])
