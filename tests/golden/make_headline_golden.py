#!/usr/bin/env python3
"""Golden vectors for the SEARCH of the headline instances (BASELINE.json configs: wordpress7_500, accap_a3, trains15),
raw and simplified networks, written by the CPU oracle (run from the repository root, no GPU needed, a few minutes).

The small instances of test_list.csv are compared tree-for-tree in full; these three cannot be searched to the end by the
sequential oracle, so the comparison is on a prefix of the tree: one workgroup, 2^0 and 2^6 subproblems, the first K nodes.
Per case: node / fail / solution / depth / subproblem counters after K nodes, SHA-256 of the best store found so far and
SHA-256 of the store the search stopped on.  K is moved down by at most 63 so that the last node is not a failed one
(the contents of a failed store depend on the evaluation order, everything else does not).

  headline_trees.json   read by tests/test_headline_trees.py: the oracle (CPU, the K <= 500 cases) and the engine (GPU,
                        all cases, without running the oracle) are held to it.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle  # noqa: E402
from turbo_amd import frontend, preprocess  # noqa: E402

INSTANCES = ["example_wordpress7_500.fzn", "accap_a3.fzn", "trains15.fzn"]
BUDGETS = [500, 2000]
POWERS = [0, 6]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def oracle_propagate(store, props):
    out, failed, _, _, _ = pyoracle.propagate(store, props)
    return out, failed


def networks(name):
    path = os.path.join(ROOT, "benchmarks", name)
    yield "raw", frontend.load_fzn(path)
    _, tcn, _ = preprocess.load_fzn_simplified(path, propagate=oracle_propagate)  # same simplifier, root fixpoints by the oracle
    yield "simplified", tcn


def record(tcn, power, budget):
    _, _, st, trace, _ = pyoracle.solve_traced(tcn, budget, power)
    k = int(st["nodes"])
    while k > max(1, int(st["nodes"]) - 63) and trace[k - 1]:
        k -= 1
    has, best, st, trace, last = pyoracle.solve_traced(tcn, k, power)
    assert int(st["nodes"]) == k
    rec = {key: int(st[key]) for key in ("nodes", "fails", "solutions", "depth_max", "eps_solved_subproblems", "eps_skipped_subproblems", "best_bound")}
    rec["cutnodes"] = k
    rec["best_store_sha256"] = sha(best) if has else None
    rec["last_node_failed"] = bool(trace[k - 1])
    rec["last_store_sha256"] = None if trace[k - 1] else sha(last)
    return rec


def main() -> None:
    out = {}
    for name in INSTANCES:
        for kind, tcn in networks(name):
            net = {"n_vars": int(tcn.n_vars), "n_props": int(tcn.n_props), "network_sha256": sha(np.ascontiguousarray(tcn.props)) + sha(np.ascontiguousarray(tcn.store)), "cases": {}}
            for power in POWERS:
                for budget in BUDGETS:
                    rec = record(tcn, power, budget)
                    net["cases"][f"sub{power}_cut{budget}"] = dict(rec, subproblems_power=power)
                    print(name, kind, power, budget, rec, flush=True)
            out[f"{name}/{kind}"] = net
    with open(os.path.join(HERE, "headline_trees.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
