#!/usr/bin/env python3
"""Long stress of the wake-up filters of the event fixpoint (engine.hip: Chains, conditional wake-up) on the element-model fuzz family, beyond the seeds of the test suite:
  1. tree identity of one workgroup against the oracle (2^0 and 2^4 subproblems, 1500 nodes, last store) in three kernel configurations per seed;
  2. full occupancy: 32 random search nodes of every 8th model, 128 copies each, propagated concurrently by the batch kernel -- all copies must reach the oracle's fixpoint.
usage (GPU box): python3 tests/tools/stress_element.py [first_seed] [n_seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import pyoracle
from turbo_amd import capi, frontend
from fuzz_models import element_model
from test_gpu_parity import random_nodes
first = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
COMPACT, C16, C8 = 0x100000, 0x10100000, 0x30100000
MODES = {"event_compact": dict(debug=COMPACT), "event_compact16": dict(debug=C16), "event_compact_4waves": dict(debug=COMPACT, threads_per_block=256),
         "event_compact8": dict(debug=C8), "event_compact8_4waves": dict(debug=C8, threads_per_block=256)}
bad = trees = 0
for seed in range(first, first + n):
    tcn = frontend.Model.from_string(element_model(seed)).tcn()
    for power in (0, 4):
        has_o, best_o, st_o, trace, last_o = pyoracle.solve_traced(tcn, 1500, power)
        for mode, kw in MODES.items():
            kw = dict(kw); dbg = kw.pop("debug")
            s = capi.Session(tcn, capi.make_config(or_nodes=1, subproblems_power=power, stop_after_n_nodes=1500, timeout_ms=120000, fixpoint=2, debug=dbg | 0x800000, **kw))
            s.start()
            while not s.poll()[1]:
                pass
            has_g, best_g, st_g = s.finish()
            last_g = s.debug_last_store(0)
            s.close()
            trees += 1
            ok = has_g == has_o and all(st_g[k] == st_o[k] for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems"))
            ok = ok and (not has_o or np.array_equal(best_g, best_o)) and (not st_o["nodes"] or trace[-1] or np.array_equal(last_g, last_o))
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed} power {power} mode {mode}: gpu nodes {st_g['nodes']} oracle {st_o['nodes']}", flush=True)
    if seed % 8 == 0:
        nodes = random_nodes(tcn, 32, seed=seed, max_decisions=6)
        exp = [pyoracle.propagate(nodes[i], tcn.props) for i in range(nodes.shape[0])]
        stores = np.tile(nodes, (128, 1))
        for dbg in (COMPACT, C16, C8):
            got, failed, ent, _, _, _ = capi.propagate(tcn.props, stores, capi.make_config(fixpoint=2, debug=dbg, timeout_ms=60000))
            miss = 0
            for j in range(stores.shape[0]):
                e = exp[j % nodes.shape[0]]
                if bool(failed[j]) != e[1] or (not e[1] and (bool(ent[j]) != e[2] or not np.array_equal(got[j], e[0]))):
                    miss += 1
            if miss:
                bad += miss
                print(f"MISMATCH seed {seed} batch debug {dbg:#x}: {miss} of {stores.shape[0]} stores", flush=True)
    if (seed - first) % 20 == 19:
        print(f"... seed {seed}: {trees} trees, {bad} mismatches so far", flush=True)
print(f"stress_element: seeds {first}..{first + n - 1}: {trees} trees walked, mismatches {bad}")
sys.exit(1 if bad else 0)
