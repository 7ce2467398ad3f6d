"""Stress: many concurrent copies of the same search nodes at full occupancy must all match the oracle."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from turbo_amd import frontend, capi
from oracle import pyoracle
from test_gpu_parity import random_nodes
for rel in sys.argv[1:] or ["test_data/sudoku_opt4.fzn", "test_data/pat2.fzn", "accap_a3.fzn"]:
    tcn = frontend.load_fzn(os.path.join(ROOT, "benchmarks", rel))
    nodes = random_nodes(tcn, 32, seed=3)
    exp = [pyoracle.propagate(nodes[i], tcn.props) for i in range(nodes.shape[0])]
    reps = 256
    stores = np.tile(nodes, (reps, 1))
    for fp, dbg in ((2, 0), (2, 0x100000), (1, 0)):  # event, event with the compact store layout, wac1
        got, failed, ent, iters, ded, _ = capi.propagate(tcn.props, stores, capi.make_config(fixpoint=fp, debug=dbg))
        bad = 0
        for j in range(stores.shape[0]):
            e = exp[j % nodes.shape[0]]
            if bool(failed[j]) != e[1]: bad += 1; continue
            if not e[1] and (bool(ent[j]) != e[2] or not np.array_equal(got[j], e[0])): bad += 1
        print(rel, "fp", fp, "debug", hex(dbg), "stores", stores.shape[0], "mismatches", bad, flush=True)
