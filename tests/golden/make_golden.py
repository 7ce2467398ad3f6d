#!/usr/bin/env python3
"""Generate the golden fixtures of this directory with the CPU oracle (run from the repository root, no GPU needed).

The reference cannot be built here (its lala dependencies are absent), so the only vectors it pins are the objectives of
benchmarks/test_list.csv; everything finer-grained is recorded from the oracle once it reproduces those objectives:

  golden.json          per instance: lowered sizes, SHA-256 of the root fixpoint, sequential search tree statistics
                       (nodes, fails, solutions, depth), objective and SHA-256 of the DFS-first optimal solution,
                       the same for the sequential EPS walk with 2^6 subproblems
  nodes_<name>.npz     for the small fixtures: a batch of search-node stores (inputs), their fixpoints, failed and
                       all-entailed flags -- the vectors the node-level GPU parity tests replay without the oracle

Both the oracle (tests/test_golden.py, CPU) and the engine (GPU) are checked against these files.
"""
import hashlib
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import BENCH, SLOW_FOR_ORACLE, known_answers  # noqa: E402
from oracle import pyoracle  # noqa: E402
from test_gpu_parity import random_nodes  # noqa: E402
from turbo_amd import frontend  # noqa: E402

SMALL = ["test_data/sudoku_opt2.fzn", "test_data/sudoku_opt3.fzn", "test_data/sudoku_opt3b.fzn", "test_data/sudoku_opt4b.fzn",
         "test_data/bug1.fzn", "test_data/bug3.fzn", "test_data/bug5.fzn", "test_data/reified_in.fzn",
         "test_data/minimize_unconstrained.fzn", "test_data/maximize_unconstrained2.fzn", "test_data/pat2.fzn", "test_data/pat7.fzn"]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main() -> None:
    out = {}
    for rel, expected in known_answers():
        tcn = frontend.load_fzn(os.path.join(BENCH, rel))
        root, failed, entailed, _, _ = pyoracle.propagate(tcn.store, tcn.props)
        rec = {"expected_objective": expected, "n_vars": tcn.n_vars, "n_props": tcn.n_props, "n_strategies": tcn.n_strats,
               "root_failed": bool(failed), "root_all_entailed": bool(entailed), "root_fixpoint_sha256": sha(root)}
        if rel not in SLOW_FOR_ORACLE:
            for key, power in (("tree", 0), ("eps6", 6)):
                has, best, st = pyoracle.solve(tcn, subproblems_power=power)
                assert has and st["exhaustive"] and tcn.objective_of(best) == expected, rel
                rec[key] = {k: int(st[k]) for k in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems")}
                rec[key]["best_store_sha256"] = sha(best)
        out[rel] = rec
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for rel in SMALL:
        tcn = frontend.load_fzn(os.path.join(BENCH, rel))
        stores = random_nodes(tcn, 24, seed=zlib.crc32(rel.encode()) % 1000)
        fix, failed, ent = [], [], []
        for i in range(stores.shape[0]):
            o, f_, e_, _, _ = pyoracle.propagate(stores[i], tcn.props)
            fix.append(o); failed.append(f_); ent.append(e_)
        name = os.path.basename(rel).replace(".fzn", "")
        np.savez_compressed(os.path.join(HERE, f"nodes_{name}.npz"), stores=stores, fixpoints=np.stack(fix),
                            failed=np.array(failed, dtype=np.int8), all_entailed=np.array(ent, dtype=np.int8), props=tcn.props)
    print("wrote", len(out), "instances,", len(SMALL), "node batches")


if __name__ == "__main__":
    main()
